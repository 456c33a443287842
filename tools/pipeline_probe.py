#!/usr/bin/env python
"""Host-time split of the pipelined loop: how long ssf_submit_frame and ssf_process_submitted
occupy the calling thread (bench workload).  python tools/pipeline_probe.py [depth]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from supersurfel_fusion_amd import binding, synthetic  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
lib = binding.load_lab()          # (the probe entry points live in the lab build: -DSSF_EXPERIMENTS)
nf = 100
frames = bench.render_frames(nf)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=depth, extract_batch=batch))
f.set_model(model, nvis, 30)
ts, tp = [], []
nsub = 0
t00 = None
for i in range(nf):
    if i == 20:
        torch.cuda.synchronize(); t00 = time.perf_counter(); ts, tp = [], []
        lib.lib.ssf_dbg_host_times(f.h, (__import__("ctypes").c_double * 8)())
    while nsub < nf and f.can_submit():
        t0 = time.perf_counter()
        f.submit_frame(d_rgb[nsub].data_ptr(), d_depth[nsub].data_ptr(), on_device=True)
        ts.append(time.perf_counter() - t0); nsub += 1
    t0 = time.perf_counter()
    f.process_submitted()
    tp.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = time.perf_counter() - t00
import ctypes as C
ht = (C.c_double * 8)()
lib.lib.ssf_dbg_host_times(f.h, ht)
print("host split over %d frames: submit %.1f us, icp loop %.1f us (first iteration incl. wait for extract %.1f us), "
      "match+fuse %.1f us; extract already done at activation: %d%%" %
      (ht[3], ht[0] / ht[3], ht[1] / ht[3], ht[5] / ht[3], ht[2] / ht[3], 100 * ht[4] / ht[3]))
print("depth %d batch %d: %.1f us/frame total; submit host %.1f us (n=%d), process host %.1f us" %
      (depth, batch, 1e6 * tot / (nf - 20), 1e6 * np.mean(ts), len(ts), 1e6 * np.mean(tp)))

# burst: how concurrent are the extract chains on the GPU?  Submit n extracts back to back, wait for all.
for nb in ():
    fb = binding.Fusion(lib, bench.make_cfg(lib, 100000, pipeline_depth=nb - 1))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for j in range(nb):
            fb.submit_frame(d_rgb[j].data_ptr(), d_depth[j].data_ptr(), on_device=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for j in range(nb):
            fb.process_submitted()
    print("burst of %d extracts: host submit %.0f us, all done after %.0f us (%.0f us per frame)" %
          (nb, 1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t2 - t0) / nb))
    fb.close()
