import ctypes, numpy as np, os, sys, struct
import torch
print("torch", torch.__version__, "cuda avail", torch.cuda.is_available())
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe.so"))
lib.probe_launch.restype = ctypes.c_double; lib.probe_graph.restype = ctypes.c_double; lib.probe_roundtrip.restype = ctypes.c_double
if torch.cuda.is_available():
    x = torch.ones(4, device="cuda"); torch.cuda.synchronize(); print("torch tensor ok", x.sum().item())
rng = np.random.default_rng(0)
n = 1 << 20
a = (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32)
b = (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32)
a[:1000] = (rng.standard_normal(1000) * 1e-38).astype(np.float32)  # denormal products
b[:1000] = (rng.standard_normal(1000) * 1e-3).astype(np.float32)
out = np.zeros((n, 8), np.float32)
rc = lib.probe_arith(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n)
print("probe_arith rc", rc)
with np.errstate(all="ignore"):
    ref = np.zeros_like(out)
    ref[:,0] = a / b
    ref[:,1] = np.sqrt(np.abs(a))
    ref[:,2] = (a * b).astype(np.float32) + a
    ref[:,3] = np.float32(1.0) / np.sqrt(np.abs(b))
    da, db = a.astype(np.float64), b.astype(np.float64)
    ref[:,4] = (da / db).astype(np.float32)
    ref[:,5] = np.sqrt(np.abs(da) + 1e-300).astype(np.float32)
    t = da * 1073741824.0
    ok = np.abs(t) < 9e18
    ref[:,6] = np.where(ok, np.rint(np.where(ok, t, 0)), 0).astype(np.float32)
    ref[:,7] = np.rint(a * np.float32(1048576.0))
names = ["f32 div", "f32 sqrt", "mul+add nocontract", "1/sqrt", "f64 div", "f64 sqrt", "double2ll_rn", "rintf"]
for j, nm in enumerate(names):
    m = np.ones(n, bool) if j != 6 else ok
    same = (out[m, j].view(np.uint32) == ref[m, j].view(np.uint32)) | (np.isnan(out[m, j]) & np.isnan(ref[m, j]))
    print(f"{nm:22s} bit-identical {same.sum()}/{m.sum()}")
for grid in (1, 256, 1200):
    print("launch chain us/kernel grid", grid, lib.probe_launch(50, 20, grid), "graph", lib.probe_graph(50, 20, grid))
print("launch+D2H+sync roundtrip us", lib.probe_roundtrip(200))
