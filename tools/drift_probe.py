"""Frame time over a long pipelined run, in chunks of 120 frames (does the rate drift as the map evolves?)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product()
dev = torch.device("cuda", 0)
frames = bench.render_frames(64)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=2, extract_batch=4))
f.set_model(model, nvis, 30)
k = 0
for chunk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    order = [((k + i) % 126) if ((k + i) % 126) < 64 else 126 - ((k + i) % 126) for i in range(120)]
    k += 120
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = f.process_sequence([d_rgb[j].data_ptr() for j in order], [d_depth[j].data_ptr() for j in order])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = f.counts()
    print("frames %5d: %.1f us/frame  n_model %d n_visible %d recentres %d icp iters %.2f removed %d inserted %d" % (
        k, 1e6 * dt / 120, c["n_model"], c["n_visible"], f.debug_recentre_count(), sum(r["icp_iters"] for r in res) / 120.0,
        sum(r["n_removed"] for r in res), sum(r["n_inserted"] for r in res)))
