"""Where does a tile's time go inside ONE relabelling pass launch?  Every workgroup leaves five ticks of the 100 MHz wall clock (lab
build: g_pass_trace / ssf_dbg_trace_pass): entry | everything requested up front has arrived | staged (first barrier) | decisions taken
(second barrier) | end.  Frames through the real pipeline first (the state of a finished extract), then one traced launch of pass 24.
    python tools/pass_trace.py [frames per launch ...]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from supersurfel_fusion_amd import binding
lib = binding.load_lab()
L = lib.lib
L.ssf_dbg_trace_pass.restype = C.c_int
L.ssf_dbg_trace_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
W, H = 640, 480
f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=50000, extract_batch=8))
for k in range(8):
    f.submit_frame(*util.frame(k, W, H))
f.process_submitted()
buf = np.zeros((65536, 5), np.uint64); g = (C.c_int * 3)()
for nb in [int(a) for a in sys.argv[1:]] or (1, 4, 8):
    for rgbd in (0, 1):
        n = L.ssf_dbg_trace_pass(f.h, rgbd, nb, buf.ctypes.data, 65536, g)
        if n <= 0:
            print("no trace", n); continue
        t = buf[:n].astype(np.int64); ok = t[:, 4] > 0
        t = t[ok]; us = (t - t[:, 0].min()) / 100.0
        ph = np.diff(us, axis=1)                     # loads | staging | decisions | flush
        q = lambda a: "%.2f / %.2f / %.2f" % (np.percentile(a, 10), np.median(a), np.percentile(a, 90))
        print("%s pass, %d frames per launch: %d workgroups (grid %d x %d x %d); microseconds, 10th / median / 90th percentile" % ("RGB-D" if rgbd else "RGB  ", nb, len(t), g[0], g[1], g[2]))
        print("   entry after the first workgroup's       %s   (last %.2f)" % (q(us[:, 0]), us[:, 0].max()))
        print("   entry -> requested data has arrived     %s" % q(ph[:, 0]))
        print("   -> staged (first barrier)               %s" % q(ph[:, 1]))
        print("   -> decisions taken (second barrier)     %s" % q(ph[:, 2]))
        print("   -> flushed, end                         %s" % q(ph[:, 3]))
        print("   inside the workgroup                    %s   launch: last end %.2f" % (q(us[:, 4] - us[:, 0]), us[:, 4].max()))
