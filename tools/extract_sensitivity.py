import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product(); dev = torch.device("cuda", 0)
frames = bench.render_frames(64)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]; d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
for seg in (10, 6, 2):
    cfg = bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=2, extract_batch=4); cfg.seg_iter = seg
    f = binding.Fusion(lib, cfg); f.set_model(model, nvis, 30)
    order = [(i % 126) if (i % 126) < 64 else 126 - (i % 126) for i in range(1248)]
    prep = f.prepare_sequence([d_rgb[j].data_ptr() for j in order[:48]], [d_depth[j].data_ptr() for j in order[:48]]); f.process_prepared(prep)
    prep = f.prepare_sequence([d_rgb[j].data_ptr() for j in order[48:]], [d_depth[j].data_ptr() for j in order[48:]])
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.process_prepared(prep); torch.cuda.synchronize()
    print("seg_iter %d: %.1f us/frame" % (seg, 1e6 * (time.perf_counter() - t0) / 1200)); f.close()
