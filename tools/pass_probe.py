"""Ablation timing of the relabelling pass kernel (back-to-back launches, hipEvents)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding
lib = binding.load_product()
lib.lib.ssf_dbg_time_pass.restype = C.c_double
lib.lib.ssf_dbg_time_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=50000))
rgb, depth = util.frame(0, W, H)
f.stage_extract(rgb, depth)
names = {0: "full", 2: "tile+copy only", 18: "tile load only", 34: "copy-out only", 50: "prologue loads only", 7+48: "nothing"}
for rgbd in (0, 1):
    print("rgbd" if rgbd else "rgb ", {names[d]: "%.1f us" % lib.lib.ssf_dbg_time_pass(f.h, 200, rgbd, d) for d in (0, 2, 18, 34, 50, 55)})
