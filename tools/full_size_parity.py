"""One-off check at BASELINE size (not in the suite: ~1 minute of oracle time): the bench workload (640x480, map seeded
with 1 M supersurfels) through the pipelined HIP path against the oracle, frame results and the whole final map bit
for bit.  python tools/full_size_parity.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, util
from conftest import ORACLE_LIB
from supersurfel_fusion_amd import binding, synthetic
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if len(sys.argv) > 4 and sys.argv[4] == "config3":            # BASELINE config 3: 1280x960, all seeded supersurfels in view, forced iterations
    bench.W, bench.H = 1280, 960
olib = binding.Library(ORACLE_LIB); plib = binding.load_product()
frames = bench.render_frames(32)
order = [(i % 62) if (i % 62) < 32 else 62 - (i % 62) for i in range(nf)]
seq = [(np.ascontiguousarray(frames[k][0]), np.ascontiguousarray(frames[k][1])) for k in order]
c3 = bench.W == 1280
model, nvis = (synthetic.seed_model_cam0_visible if c3 else synthetic.seed_model_cam0)(bench.N_MODEL, bench.W, bench.H, stamp=30)
fo = binding.Fusion(olib, bench.make_cfg(olib, bench.N_MODEL + 65536, force_icp=c3))
fh = binding.Fusion(plib, bench.make_cfg(plib, bench.N_MODEL + 65536, force_icp=c3, pipeline_depth=depth, extract_batch=batch))
fo.set_model(model, nvis, 30); fh.set_model(model, nvis, 30)
t0 = time.time()
want = [fo.process_frame(r, d) for r, d in seq]
t1 = time.time()
got = fh.process_sequence([r.ctypes.data for r, _ in seq], [d.ctypes.data for _, d in seq], on_device=False)
for i, (a, b) in enumerate(zip(want, got)):
    for key in util.RESULT_KEYS:
        assert a[key] == b[key], (i, key, a[key], b[key])
    util.assert_same_bits(a["pose"], b["pose"], "pose of frame %d" % i)
util.compare_state(fo, fh)
print("ok (%dx%d, depth %d, batch %d): %d frames, n_model %d n_visible %d, oracle %.0f s" % (bench.W, bench.H, depth, batch, nf, want[-1]["n_model"], want[-1]["n_visible"], t1 - t0))
