#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pmc_rNN.json.
  python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<note>"
Counters are KB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the
bytes read (calibrated on this build's own k_scatter, see the note in the output)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NAMES = [("k_update_pass<true, 1", "update_pass_rgbd"), ("k_update_pass<false, 1", "update_pass_rgb"),
         ("k_update_pass<true, 2", "update_pass_rgbd_wide"), ("k_update_pass<false, 2", "update_pass_rgb_wide"), ("k_move_rows", "reorder_move"), ("k_scatter", "reorder_scatter"),
         ("k_classify", "classify"), ("k_icp", "icp_accumulate"), ("k_match", "match"), ("k_render_moments", "render_moments"),
         ("k_ingest", "ingest"), ("k_eval_samples", "eval_samples"), ("k_init_disp", "init_disp"), ("k_init_samples", "init_samples"),
         ("k_plane_filter", "plane_filter"), ("k_finalize_surfels", "finalize_surfels"), ("k_update_insert", "update_insert"),
         ("k_scan_blocks", "scan_blocks")]


def short(name):
    for pat, nm in NAMES:
        if pat in name:
            return nm
    return None


def collect(d, counter):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pmc_counters
    raw = defaultdict(list)
    for name, cname, value, grid in pmc_counters.rows_of(d, with_grid=True):
        if cname != counter:
            continue
        nm = short(name)
        if nm:
            raw[nm].append((grid, value))
    # extract kernels are launched over 2, 4 and 8 frames (the batch ramp): per-launch figures are those of the full batch
    batched = ("update_pass", "ingest", "init_", "eval_samples", "plane_filter", "render_moments", "finalize_surfels")
    acc = defaultdict(list)
    for nm, lst in raw.items():
        gmax = max(g for g, _ in lst)
        acc[nm] = [v for g, v in lst if g == gmax or not nm.startswith(batched)]
    return acc


def main(fetch_dir, write_dir, out, note):
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for nm in sorted(set(fe) | set(wr)):
        f = sum(fe[nm]) / max(len(fe[nm]), 1)
        w = sum(wr[nm]) / max(len(wr[nm]), 1)
        kernels[nm] = dict(FETCH_SIZE_KB_mean=f, WRITE_SIZE_KB_mean=w, launches=len(fe[nm]),
                           hbm_bytes_per_launch=1024.0 * (2.0 * f + w))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    eb = int(os.environ.get("PMC_EXTRACT_BATCH", "8"))
    json.dump(dict(note=note, source_sha=bench.kernel_source_sha(), extract_batch=eb, kernels=kernels), open(out, "w"), indent=1)
    for nm, k in kernels.items():
        print("%-20s %6d launches  fetch %10.1f KB  write %10.1f KB  -> %8.2f MB / launch" %
              (nm, k["launches"], k["FETCH_SIZE_KB_mean"], k["WRITE_SIZE_KB_mean"], k["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
