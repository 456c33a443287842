#!/usr/bin/env python
"""Throughput of the extract stage alone (ssf_dbg_extract_only: frames go through the batch contexts and are
retired unread).  python tools/extract_only_probe.py [depth batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from supersurfel_fusion_amd import binding  # noqa: E402

dev = torch.device("cuda", 0)
lib = binding.load_lab()          # (the probe entry points live in the lab build: -DSSF_EXPERIMENTS)
fn = lib.lib.ssf_dbg_extract_only
fn.restype = C.c_double
fn.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int]
frames = bench.render_frames(16)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
pr = (C.c_void_p * 16)(*[t.data_ptr() for t in d_rgb])
pd = (C.c_void_p * 16)(*[t.data_ptr() for t in d_depth])
cases = ((0, 1), (0, 4), (1, 4), (2, 4), (2, 8), (3, 4)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),)
for depth, batch in cases:
    f = binding.Fusion(lib, bench.make_cfg(lib, 100000, pipeline_depth=depth, extract_batch=batch))
    print("depth %d batch %d: extract only %.1f us/frame" % (depth, batch, fn(f.h, pr, pd, 16, 640)))
    del f
