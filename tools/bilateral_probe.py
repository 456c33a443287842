"""Time ssf_bilateral_filter (640x480 and 1280x960, frames in HBM) and check it against the CPU oracle bit for bit.
Variants through the environment: SSF_BILATERAL_GENERIC=1 (the any-radius kernel), SSF_BIL_WAVES=2|3|4."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from supersurfel_fusion_amd import binding, synthetic  # noqa: E402


def main():
    lib = (binding.load_lab() if (os.environ.get("SSF_BIL_WAVES") or os.environ.get("SSF_BILATERAL_GENERIC")) else binding.load_product())
    olib = binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle.so"))
    dev = torch.device("cuda", 0)
    for (W, H) in ((640, 480), (1280, 960)):
        K = synthetic.intrinsics(W, H)
        cfg = {k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}
        f = binding.Fusion(lib, lib.default_config(nb_supersurfels_max=20000, **cfg))
        fo = binding.Fusion(olib, olib.default_config(nb_supersurfels_max=20000, **cfg))
        R, t = synthetic.orbit_pose(0)
        rgb, depth, _ = synthetic.render(R, t, W, H, noise=True, holes=0.1, rng=np.random.default_rng(1000))
        depth = np.ascontiguousarray(depth, np.float32)
        want = fo.bilateral_filter(depth)
        got = f.bilateral_filter(depth)
        same = bool(np.array_equal(want.view(np.uint32), got.view(np.uint32)))
        d_in = torch.from_numpy(depth).to(dev); d_out = torch.empty_like(d_in)
        f.set_profile(1)
        for rep in range(12):
            if rep == 2:
                f.reset_kernel_times()
            f._ck(lib.lib.ssf_bilateral_filter(f.h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), 1), "ssf_bilateral_filter")
        ms, calls = f.kernel_times().get("bilateral_prefilter", (0.0, 0))
        print("%dx%d  %s  %.2f us  bit-exact vs oracle: %s" % (W, H, os.environ.get("SSF_BILATERAL_GENERIC") and "generic" or "r7 waves=" + os.environ.get("SSF_BIL_WAVES", "3"),
                                                              1000.0 * ms / max(calls, 1), same))


if __name__ == "__main__":
    main()
