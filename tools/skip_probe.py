"""How many tiles of a relabelling pass prove themselves clean (k_update_pass, clean-tile skipping)?  Lab build: per-pass counters.
python tools/skip_probe.py [W H]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding
lib = binding.load_lab()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=50000))
ntiles = ((W + 30 + 31) // 32) * ((H + 31) // 32)
for k in range(3):
    f.process_frame(*util.frame(k, W, H))
    out = np.zeros(64, np.uint32)
    lib.lib.ssf_dbg_pass_skips.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.lib.ssf_dbg_pass_skips(f.h, out.ctypes.data_as(C.c_void_p)) == 0
    print("frame %d: %d tiles per pass; clean tiles in pass 0..39 (counters wrap at 32: passes 32..39 share 0..7):" % (k, ntiles))
    print("   ", out[1:33].tolist())
