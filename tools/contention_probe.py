"""What slows the track chain when host frames are uploaded beside it?  The bench sequence with frames RESIDENT in HBM (no upload by the
library) while a background thread of this script does one of: nothing | host memcpys of 2.1 MB (no HIP call) | 2.1 MB page-locked ->
device copies on a stream of its own | 64-byte copies at the same call rate.  python tools/contention_probe.py"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product()
nf = 24 + 1200
frames = bench.render_frames(64)
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
order = [(i % 126) if (i % 126) < 64 else 126 - (i % 126) for i in range(nf)]
dev = torch.device("cuda", 0)
keep = [(torch.from_numpy(f[0]).to(dev), torch.from_numpy(f[1]).to(dev)) for f in frames]
NB = 2150400
src_np = [np.random.default_rng(i).integers(0, 255, NB, dtype=np.uint8) for i in range(64)]
pin = [torch.empty(NB, dtype=torch.uint8).pin_memory() for _ in range(8)]
dst = [torch.empty(NB, dtype=torch.uint8, device=dev) for _ in range(8)]
side = torch.cuda.Stream(dev)
stop = False
calls = [0]
def bg(mode, rate_hz):
    k = 0
    period = 1.0 / rate_hz
    nxt = time.perf_counter()
    while not stop:
        if mode == "memcpy": np.copyto(pin[k % 8].numpy(), src_np[k % 64])
        elif mode == "h2d_2MB":
            with torch.cuda.stream(side): dst[k % 8].copy_(pin[k % 8], non_blocking=True)
        elif mode == "h2d_64B":
            with torch.cuda.stream(side): dst[k % 8][:64].copy_(pin[k % 8][:64], non_blocking=True)
        k += 1; calls[0] = k
        if k % 64 == 0 and mode.startswith("h2d"): side.synchronize()
        nxt += period
        d = nxt - time.perf_counter()
        if d > 0: time.sleep(d)
        else: nxt = time.perf_counter()
for mode, rate in (("none", 1), ("memcpy", 10000), ("h2d_2MB", 10000), ("h2d_64B", 20000), ("h2d_2MB", 5000), ("none", 1)):
    f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=2, extract_batch=8))
    f.set_model(model, nvis, 30)
    pr = [keep[k][0].data_ptr() for k in order]; pd = [keep[k][1].data_ptr() for k in order]
    f.process_prepared(f.prepare_sequence(pr[:24], pd[:24]), on_device=True)
    prep = f.prepare_sequence(pr[24:], pd[24:])
    stop = False; calls[0] = 0
    th = threading.Thread(target=bg, args=(mode, rate)) if mode != "none" else None
    if th: th.start(); time.sleep(0.05)
    c0 = calls[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.process_prepared(prep, on_device=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = calls[0]
    stop = True
    if th: th.join()
    print("background %-8s (asked %5d calls/s, made %6.0f/s): %.1f us/frame, %.0f frames/s" % (mode, rate, (c1 - c0) / dt, 1e6 * dt / (nf - 24), (nf - 24) / dt))
    f.close()
