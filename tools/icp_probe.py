"""Ablation timing of the ICP accumulate kernel (back-to-back launches, hipEvents)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding, synthetic
# (the measurement arms and their switches live in the lab build of the sources)
lib = binding.load_lab()          # (ssf_dbg_time_icp and the probe bits: lab build)
lib.lib.ssf_dbg_time_icp.restype = C.c_double
lib.lib.ssf_dbg_time_icp.argtypes = [C.c_void_p, C.c_int, C.c_int]
W, H = (1280, 960) if "--config3" in sys.argv else (640, 480)
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=1100000))
model, nvis = (synthetic.seed_model_cam0_visible if "--config3" in sys.argv else synthetic.seed_model_cam0)(1000000, W, H)
f.set_model(model, nvis, 30)
rgb, depth = util.frame(0, W, H)
f.stage_extract(rgb, depth); f.icp_begin()
names = {0: "full", 4: "no tail", 6: "no accumulation, no tail", 7: "loads only", 12: "no tail, frame supersurfel independent of the pixel word (2 trips)",
         20: "no tail, lanes gather adjacent pixels", 28: "no tail, both", 14: "no accumulation, no tail, 2 trips", 22: "no accumulation, no tail, adjacent pixels"}
print("per_lane", os.environ.get("SSF_ICP_PER_LANE"), "wred", os.environ.get("SSF_ICP_WRED"), "nvis", nvis)
for d in (0, 4, 6, 7, 12, 20, 28, 14, 22):
    if os.environ.get("SSF_ICP_WRED") and d not in (0, 4):
        continue
    print("   %-75s %.1f us" % (names[d], lib.lib.ssf_dbg_time_icp(f.h, 200, d)))
