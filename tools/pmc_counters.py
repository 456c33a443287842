#!/usr/bin/env python
"""Mean value per launch of every counter in a rocprofv3 --pmc CSV directory, per kernel (short names).
  python tools/pmc_counters.py <dir> [kernel substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows_of(d, with_grid=False):
    """(kernel name, counter name, value[, grid size]) of every dispatch: rocprofv3 CSV output or its rocpd sqlite database"""
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                row = (r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]))
                yield row + (int(r.get("Grid_Size", 0) or 0),) if with_grid else row
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        for name, counter, value, grid in cur.execute("select kernel_name, counter_name, value, grid_size from counters_collection"):
            yield (name, counter, float(value), int(grid)) if with_grid else (name, counter, float(value))


def main(d, filt):
    """extract kernels are launched over 2, 4 and 8 frames (the batch ramp): only the full-size launches of a kernel are averaged"""
    raw = defaultdict(lambda: defaultdict(list))
    for name, counter, value, grid in rows_of(d, with_grid=True):
        nm = name.split("(")[0].replace("void ", "").replace("ssf::", "")
        if filt and not any(x in nm for x in filt):
            continue
        raw[nm][counter].append((grid, value))
    batched = ("k_update_pass", "k_ingest", "k_init_", "k_eval_samples", "k_plane_filter", "k_render_moments", "k_finalize_surfels", "k_bilateral")
    acc = defaultdict(lambda: defaultdict(list))
    for nm in raw:
        for c, lst in raw[nm].items():
            gmax = max(g for g, _ in lst)
            acc[nm][c] = [v for g, v in lst if g == gmax or not nm.startswith(batched)]
    for nm in sorted(acc):
        n = max(len(v) for v in acc[nm].values())
        print("%s  (%d launches)" % (nm, n))
        for c in sorted(acc[nm]):
            v = acc[nm][c]
            print("    %-28s %14.1f" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
