import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms")}, d["cpu_baseline"] and d["cpu_baseline"]["value"])
print(d["roofline"]); print("icp iters", d["config"]["icp_iters_mean"], "n_model", d["config"]["n_model"], "n_visible", d["config"]["n_visible"])
for k, v in sorted(d["per_kernel"].items(), key=lambda kv: -kv[1]["total_ms_per_frame"]):
    print("%-20s %7.1f us/frame  %5.1f launches  %7.2f us avg  %s" % (k, 1000 * v["total_ms_per_frame"], v["launches_per_frame"], v["avg_us"], "%.0f GB/s" % v["achieved_GBs"] if "achieved_GBs" in v else ""))
