"""Dump the per-kernel statistics of a rocprofv3 --kernel-trace --stats results.db as a text table
(profiles/*.txt are produced with this from gpurun_out/)."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % note)
        f.write("# %-90s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            f.write("%-92s %8d %14.3f %12.3f %7.2f\n" % (name[:92], calls, tot / 1000.0 if tot > 1e7 else tot, avg / 1000.0 if tot > 1e7 else avg, pct))
    print(open(out_path).read())


def launch_mix(db_path, out_json, note="", pixels=640 * 480):
    """The dominant kernel as the PROFILE sees it (round 6): the relabelling launches of the trace grouped by frames per launch
    (grid_z), and the roofline fraction over the trace's own launch mix -- algorithmic bytes 14 (RGB-D) / 9 (RGB) x pixels x frames
    of every launch over the sum of their durations -- beside the per-kernel-NAME shares.  bench.py's hipEvent figure is quoted at
    full batches; this one includes the short leading batches of a sequence (`roofline.frac_from_rocprof`)."""
    import json
    import os
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, end - start, grid_z from kernels").fetchall()
    tot = float(sum(d for _, d, _ in rows)) or 1.0
    by_name, mix = {}, {}
    for name, d, gz in rows:
        e = by_name.setdefault(name.split("(")[0].replace("void ssf::", ""), [0, 0.0])
        e[0] += 1; e[1] += d
        for tag, key, bpp in (("k_update_pass<true", "update_pass_rgbd", 14.0), ("k_update_pass<false", "update_pass_rgb", 9.0)):
            if tag in name:
                m = mix.setdefault(key, dict(bytes_per_pixel=bpp, by_frames_per_launch={}))
                g = m["by_frames_per_launch"].setdefault(str(int(gz)), dict(launches=0, total_us=0.0))
                g["launches"] += 1; g["total_us"] += d / 1000.0
    for key, m in mix.items():
        n = sum(g["launches"] for g in m["by_frames_per_launch"].values())
        us = sum(g["total_us"] for g in m["by_frames_per_launch"].values())
        fr = sum(int(z) * g["launches"] for z, g in m["by_frames_per_launch"].items())
        for g in m["by_frames_per_launch"].values():
            g["avg_us"] = g["total_us"] / g["launches"]
        m.update(launches=n, avg_us=us / max(n, 1), mean_frames_per_launch=fr / max(n, 1),
                 achieved_GBs=m["bytes_per_pixel"] * pixels * fr / (us * 1e-6) / 1e9 if us else None)
        m["frac_of_8TBs"] = m["achieved_GBs"] / 8000.0 if us else None
    top = sorted(by_name.items(), key=lambda kv: -kv[1][1])[:12]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump(dict(note=note, source_sha=bench.kernel_source_sha(), pixels=pixels, gpu_time_us=tot / 1000.0,
                   top_kernels_by_name=[dict(kernel=k, launches=v[0], total_us=v[1] / 1000.0, avg_us=v[1] / 1000.0 / v[0], share=v[1] / tot) for k, v in top],
                   relabelling_launch_mix=mix), open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    if len(sys.argv) > 4:
        launch_mix(sys.argv[1], sys.argv[4], sys.argv[3], int(sys.argv[5]) if len(sys.argv) > 5 else 640 * 480)
