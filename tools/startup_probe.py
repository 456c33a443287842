"""Where the time of a SHORT sequence goes (the driver runs bench.py --steps 20): completion time of every frame of a
20-frame ssf_process_sequence on an idle GPU, next to the steady-state frame time of a long one.
  python tools/startup_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product(); dev = torch.device("cuda", 0)
frames = bench.render_frames(64)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]; d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=2, extract_batch=8))
f.set_model(model, nvis, 30)
idx = lambda j: j % 60
seq = lambda n, o=0: f.prepare_sequence([d_rgb[idx(o + j)].data_ptr() for j in range(n)], [d_depth[idx(o + j)].data_ptr() for j in range(n)])
f.process_prepared(seq(40)); f.process_prepared(seq(20, 40))            # graphs of every batch size built
lib.lib.ssf_sequence_times.argtypes = [C.c_void_p, C.c_void_p]
for rep in range(3):
    prep = seq(20, 60 + 20 * rep)
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.process_prepared(prep); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t = np.zeros(64); lib.lib.ssf_sequence_times(f.h, t.ctypes.data_as(C.c_void_p))
    print("20 frames: %.0f us in all (%.0f frames/s); frame k done at [us]: %s" % (1e6 * dt, 20 / dt, " ".join("%d" % v for v in t[:20])))
    print("   per-frame increments: %s" % " ".join("%d" % v for v in np.diff(np.concatenate([[0], t[:20]]))))
prep = seq(600, 120)
torch.cuda.synchronize(); t0 = time.perf_counter(); f.process_prepared(prep); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("600 frames: %.1f us per frame" % (1e6 * dt / 600))
