#!/usr/bin/env python
"""Vector-issue utilisation per kernel (round 6): how close a "compute-bound" kernel is to the part's VALU issue rate.
  python tools/pmc_issue.py <pmc_sq2_dir> <trace_results.db> <out.json> ["note"]
SQ_INSTS_VALU (wave-level vector instructions per launch, from the rocprofv3 --pmc pass that holds it) over the kernel's mean
duration in the SEPARATE --kernel-trace run (counter collection perturbs durations), against the part's issue peak: 256 CUs x 4
SIMDs, one wave64 vector instruction per SIMD every 4 cycles at 2.4 GHz = 614 G wave-instructions/s = 39.3 T lane-operations/s
(MI355X_MICROARCH.md).  A packed-fp32 instruction counts once (it is one issue slot)."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_counters      # noqa: E402

PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4.0


def main(sq2_dir, trace_db, out, note=""):
    vals = defaultdict(lambda: defaultdict(list))
    for name, counter, value, grid in pmc_counters.rows_of(sq2_dir, with_grid=True):
        nm = name.split("(")[0].replace("void ", "").replace("ssf::", "")
        vals[nm][counter].append((grid, value))
    dur = defaultdict(list)
    cur = sqlite3.connect(trace_db).cursor()
    for name, d, gx, gy, gz in cur.execute("select name, end - start, grid_x, grid_y, grid_z from kernels"):
        dur[name.split("(")[0].replace("void ", "").replace("ssf::", "")].append((gx * gy * gz, d / 1000.0))
    res = {}
    for nm, cs in vals.items():
        if "SQ_INSTS_VALU" not in cs or nm not in dur:
            continue
        gmax = max(g for g, _ in cs["SQ_INSTS_VALU"])           # (full-size launches only: the batch ramp launches smaller ones)
        valu = [v for g, v in cs["SQ_INSTS_VALU"] if g == gmax]
        dmax = max(g for g, _ in dur[nm])
        us = [d for g, d in dur[nm] if g == dmax]
        v, t = sum(valu) / len(valu), sum(us) / len(us)
        ent = dict(launches_counted=len(valu), valu_wave_insts_per_launch=v, avg_us_in_trace=t,
                   valu_issue_frac=v / (t * 1e-6) / PEAK_WAVE_INSTS)
        for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in cs:
                w = [x for g, x in cs[c] if g == gmax]
                ent[c + "_per_launch"] = sum(w) / len(w)
        res[nm] = ent
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump(dict(note=note, source_sha=bench.kernel_source_sha(), peak_wave_insts_per_s=PEAK_WAVE_INSTS, kernels=res), open(out, "w"), indent=1)
    for nm, e in sorted(res.items(), key=lambda kv: -kv[1]["valu_issue_frac"]):
        print("%-40s %12.0f VALU wave-insts / launch  %9.2f us  -> %.3f of the vector issue peak" % (nm[:40], e["valu_wave_insts_per_launch"], e["avg_us_in_trace"], e["valu_issue_frac"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
