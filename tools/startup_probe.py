import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product(); dev = torch.device("cuda", 0)
frames = bench.render_frames(64)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]; d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=2, extract_batch=8))
f.set_model(model, nvis, 30)
prep = f.prepare_sequence([d_rgb[j].data_ptr() for j in range(40)], [d_depth[j].data_ptr() for j in range(40)]); f.process_prepared(prep)
for K in (20, 20, 50):
    prep = f.prepare_sequence([d_rgb[j % 60].data_ptr() for j in range(K)], [d_depth[j % 60].data_ptr() for j in range(K)])
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.process_prepared(prep); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("native K=%d: call %.0f us, +sync %.0f us" % (K, 1e6*(t1-t0), 1e6*(t2-t1)))
# manual loop with timestamps
K = 20
torch.cuda.synchronize(); t0 = time.perf_counter(); ts = []; nsub = 0
for k in range(K):
    while nsub < K and f.can_submit():
        f.submit_frame(d_rgb[nsub % 60].data_ptr(), d_depth[nsub % 60].data_ptr(), on_device=True); nsub += 1
    f.process_submitted(); ts.append(1e6 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("manual (full batches of 8): frame completion times us:", [int(t) for t in ts], "end %.0f" % (1e6 * (time.perf_counter() - t0)))
