"""Stress of tests/test_baseline_configs_gpu.py::test_config2 (1 M rows, host frames, pipelined 2 x 8): N fresh handles."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product()
W, H, nf, n_seed = 640, 480, 10, 1000000
model, nvis = synthetic.seed_model_cam0(n_seed, W, H, stamp=30)
base = [util.frame(k, W, H) for k in range(6)]
order = [(i % 10) for i in range(nf)]
frames = [(np.ascontiguousarray(base[j if j < 6 else 10 - j][0], np.uint8), np.ascontiguousarray(base[j if j < 6 else 10 - j][1], np.float32)) for j in order]
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    fh = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=n_seed + 65536, pipeline_depth=2, extract_batch=8))
    fh.set_model(model, nvis, 30)
    t0 = time.time()
    try:
        got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    except Exception as e:
        print("rep", rep, "FAILED after %.1f s:" % (time.time() - t0), e, flush=True)
        break
    sig = [(g["n_model"], g["n_visible"], g["icp_iters"], g["pose"].tobytes()) for g in got]
    if ref is None:
        ref = sig
    print("rep", rep, "ok %.2f s" % (time.time() - t0), "same" if sig == ref else "DIFFERENT", flush=True)
    fh.close()
