"""k_deformation on its own (round 6): 1 M seeded rows, N / 50 nodes, hipEvent-timed by the library; the product or a variant build
(SSF_PRODUCT_VARIANT=<tag>: tools/build_variant.sh <tag> -DSSF_DEFORM_FORM=n).  Prints avg us per call (pack launch + apply launch)
and a checksum of the deformed map (the forms that are not ablations must agree bit for bit).
    python tools/deform_probe.py [rows] [coherent]
coherent: a row's four nodes are drawn near row / 50 (rows that are neighbours in the array share nodes) instead of uniformly."""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
from supersurfel_fusion_amd import binding, synthetic     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
coherent = len(sys.argv) > 2 and sys.argv[2] == "coherent"
lib = binding.load_product()
W, H = 640, 480
K = synthetic.intrinsics(W, H)
cfg = lib.default_config(**dict({k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}, nb_supersurfels_max=n + 4096))
f = binding.Fusion(lib, cfg)
model, nvis = synthetic.seed_model_cam0(n, W, H, stamp=30)
f.set_model(model, nvis, 30)
rng = np.random.default_rng(5)
m = max(n // 50, 4)
npos = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
ang = rng.uniform(-0.05, 0.05, (m, 3))
nrot = np.zeros((m, 9), np.float32)
for k in range(m):            # small rotations (Rodrigues, float64 -> float32)
    a = ang[k]; th = np.linalg.norm(a) + 1e-12; u = a / th
    Kx = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    nrot[k] = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).astype(np.float32).reshape(9)
ntr = rng.uniform(-1e-3, 1e-3, (m, 3)).astype(np.float32)
w4 = rng.dirichlet(np.ones(4), n).astype(np.float32)
if coherent:
    idx = np.clip((np.arange(n)[:, None] // 50) + rng.integers(-2, 3, (n, 4)), 0, m - 1).astype(np.int32)
else:
    idx = rng.integers(0, m, (n, 4)).astype(np.int32)
f.set_profile(1)
for rep in range(6):
    if rep == 1:
        f.reset_kernel_times()
    f.apply_deformation(npos, nrot, ntr, w4, idx)
ms, calls = f.kernel_times().get("apply_deformation", (0.0, 0))
mm = f.get_model()
crc = 0
for name in sorted(mm):
    crc = zlib.crc32(np.ascontiguousarray(mm[name]).view(np.uint8), crc)
us = 1000.0 * ms / max(calls, 1)
print("variant %-6s rows %d nodes %d %s: %.2f us per call (%d calls)  %.0f GB/s algorithmic (176 B/row) = %.3f of 8 TB/s   crc %08x" %
      (os.environ.get("SSF_PRODUCT_VARIANT", "product"), n, m, "coherent" if coherent else "random", us, calls, 176.0 * n / us / 1e3, 176.0 * n / us / 1e3 / 8000.0, crc))
f.close()
