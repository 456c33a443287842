"""Long bit-exactness soaks (not part of the test suite: the oracle needs minutes): pipelined / batched runs of 300-1500
frames at 320x240 and 640x480 against the oracle, with culling pressure and store upkeep.  python tools/long_soak.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import util
from supersurfel_fusion_amd import binding
from conftest import ORACLE_LIB
olib = binding.Library(ORACLE_LIB); plib = binding.load_product()
for (W, H, nf, kw, depth, batch) in [(320, 240, 1500, dict(nb_supersurfels_max=9000, delta_t=15), 2, 4),
                                      (320, 240, 1200, dict(nb_supersurfels_max=2500, delta_t=6, conf_thresh=1e9), 2, 3),
                                      (640, 480, 300, dict(nb_supersurfels_max=60000), 1, 8)]:
    fo = binding.Fusion(olib, util.make_cfg(olib, W, H, **kw))
    fh = binding.Fusion(plib, util.make_cfg(plib, W, H, pipeline_depth=depth, extract_batch=batch, **kw))
    base = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(0, 61, 2)]
    seq = (base + base[-2:0:-1]) * (nf // 58 + 1)
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in seq[:nf]]
    t0 = time.time()
    want = [fo.process_frame(r, d) for r, d in frames]
    t1 = time.time()
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for i, (a, b) in enumerate(zip(want, got)):
        for key in util.RESULT_KEYS:
            assert a[key] == b[key], (i, key, a[key], b[key])
        util.assert_same_bits(a["pose"], b["pose"], "pose of frame %d" % i)
    util.compare_state(fo, fh)
    print("ok %dx%d %d frames: n_model %d, recentres %d, oracle %.0f s, hip %.1f s" % (W, H, nf, want[-1]["n_model"], fh.debug_recentre_count(), t1 - t0, time.time() - t1), flush=True)
