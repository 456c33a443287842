"""Which arm bounds the fuse launch (k_update_insert: update | insert | classification of the visible rows | classification of the
out-of-view slots)?  Back-to-back launches on one frame with arms switched off (lab build: ssf_dbg_time_fuse)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_product() if os.environ.get("SSF_PRODUCT_VARIANT") else binding.load_lab()      # (a variant: built with -DSSF_EXPERIMENTS too)
lib.lib.ssf_dbg_time_fuse.restype = C.c_double
lib.lib.ssf_dbg_time_fuse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
W, H = 640, 480
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=1100000))
model, nvis = synthetic.seed_model_cam0(1000000, W, H)
f.set_model(model, nvis, 30)
b2 = (C.c_longlong * 2)()
for k in range(3):
    r = f.process_frame(*util.frame(k, W, H))
    lib.lib.ssf_dbg_time_fuse(f.h, 0, 0, b2)
    print("frame %d: n_visible %d, removed %d; out-of-view blocks %d, with rows that move %d" % (k, r["n_visible"], r["n_removed"], b2[0], b2[1]))
rgb, depth = util.frame(3, W, H)
f.stage_extract(rgb, depth); f.icp_begin()
again = True
while again:
    again = f.icp_update(f.icp_accumulate())
f.icp_end()
best, matched = f.match()
print("n_visible", f.counts()["n_visible"], "n_model", f.counts()["n_model"])
names = {0: "full", 1: "without the out-of-view arm", 2: "without update + insert", 3: "classification of the visible rows only",
         6: "out-of-view arm only", 7: "empty grid (launch floor)"}
print("   first launch on the frame: %.2f us;" % lib.lib.ssf_dbg_time_fuse(f.h, 1, 0, b2), "out-of-view blocks %d, with rows that move %d" % (b2[0], b2[1]))
for mode in (0, 1, 2, 3, 6, 7, 0):
    t = lib.lib.ssf_dbg_time_fuse(f.h, 200, mode, b2)
    print("   %-45s %.2f us   (out-of-view blocks %d, with rows that move %d)" % (names[mode], t, b2[0], b2[1]))

# where the time goes inside ONE launch: every workgroup's entry | end of its arm | memory acknowledged (100 MHz wall clock)
import numpy as np
lib.lib.ssf_dbg_trace_fuse.restype = C.c_int
lib.lib.ssf_dbg_trace_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
buf = np.zeros((65536, 3), np.uint64); arms = (C.c_int * 4)()
for mode in (0, 1, 6, 3):
    n = lib.lib.ssf_dbg_trace_fuse(f.h, buf.ctypes.data, 65536, arms, mode)
    if n <= 0:
        continue
    t = buf[:n].astype(np.int64); t0 = t[:, 0].min(); us = (t - t0) / 100.0
    print("one launch (%s), %d workgroups; microseconds after the first workgroup's entry:" % (names[mode], n))
    lo = 0
    for name, k in zip(("update", "insertion", "visible rows", "out-of-view span"), arms):
        a = us[lo:lo + k]; lo += k
        if len(a):
            d = a[:, 2] - a[:, 0]
            print("   %-17s %5d workgroups  entry %.2f .. %.2f   arm done median %.2f max %.2f   memory acknowledged median %.2f max %.2f   inside the workgroup: min %.2f median %.2f max %.2f"
                  % (name, len(a), a[:, 0].min(), a[:, 0].max(), np.median(a[:, 1]), a[:, 1].max(), np.median(a[:, 2]), a[:, 2].max(), d.min(), np.median(d), d.max()))
