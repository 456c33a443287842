"""Timing of the classify-histogram scan / counter publication kernel alone (back-to-back launches)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product()
lib.lib.ssf_dbg_time_scan.restype = C.c_double
lib.lib.ssf_dbg_time_scan.argtypes = [C.c_void_p, C.c_int, C.c_int]
f = binding.Fusion(lib, util.make_cfg(lib, 640, 480, nb_supersurfels_max=1100000))
model, nvis = synthetic.seed_model_cam0(1000000, 640, 480)
f.set_model(model, nvis, 30)
f.process_frame(*util.frame(0, 640, 480))
for mode, name in ((1, "one thread, counters only"), (0, "one thread, counters + mailbox publication"), (2, "full scan kernel")):
    print("%-45s %.1f us" % (name, lib.lib.ssf_dbg_time_scan(f.h, 200, mode)))
