"""Ablation timing of the relabelling pass kernel (back-to-back launches, hipEvents), for 1..8
frames per launch.  python tools/pass_probe.py [W H]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding
lib = binding.load_lab()          # (the probe entry points live in the lab build: -DSSF_EXPERIMENTS)
lib.lib.ssf_dbg_time_pass.restype = C.c_double
lib.lib.ssf_dbg_time_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=50000, extract_batch=8))
for k in range(8):
    f.submit_frame(*util.frame(k, W, H))
f.process_submitted()
names = {0: "full", 1: "no window rows", 8: "no replay", 4: "nothing eligible", 2: "staging only", 3: "tile load only",
         35: "prologue loads only"}
for nb in (1, 2, 4, 8):
    for rgbd in (0, 1):
        print("nb=%d %s" % (nb, "rgbd" if rgbd else "rgb "),
              {names[d]: "%.1f" % lib.lib.ssf_dbg_time_pass(f.h, 200, rgbd, d, nb) for d in names})
