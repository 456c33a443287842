"""Replay rate when the caller hands over HOST images (pageable numpy arrays, or page-locked torch tensors) instead of
frames resident in HBM: ssf_process_sequence(on_device=0).  python tools/host_buffer_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from supersurfel_fusion_amd import binding, synthetic
lib = binding.load_lab()          # (the probe entry points live in the lab build: -DSSF_EXPERIMENTS)
nf = 24 + int(os.environ.get('PROBE_NF', '1200'))
frames = bench.render_frames(64)
model, nvis = synthetic.seed_model_cam0(bench.N_MODEL, bench.W, bench.H, stamp=30)
order = [(i % 126) if (i % 126) < 64 else 126 - (i % 126) for i in range(nf)]
for kind in os.environ.get("PROBE_KINDS", "device,pageable,pinned").split(","):
    if kind == "device":
        dev = torch.device("cuda", 0)
        keep = [(torch.from_numpy(f[0]).to(dev), torch.from_numpy(f[1]).to(dev)) for f in frames]
    elif kind == "pinned":
        keep = [(torch.from_numpy(f[0]).pin_memory(), torch.from_numpy(f[1]).pin_memory()) for f in frames]
    else:
        keep = [(torch.from_numpy(np.ascontiguousarray(f[0])), torch.from_numpy(np.ascontiguousarray(f[1]))) for f in frames]
    f = binding.Fusion(lib, bench.make_cfg(lib, bench.N_MODEL + 65536, pipeline_depth=int(os.environ.get("PROBE_DEPTH", "2")), extract_batch=int(os.environ.get("PROBE_BATCH", "8")),
                                           prefilter=int(os.environ.get("PROBE_PREFILTER", "0"))))
    f.set_model(model, nvis, 30)
    pr = [keep[k][0].data_ptr() for k in order]; pd = [keep[k][1].data_ptr() for k in order]
    f.process_prepared(f.prepare_sequence(pr[:24], pd[:24]), on_device=(kind == "device"))
    prep = f.prepare_sequence(pr[24:], pd[24:])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.process_prepared(prep, on_device=(kind == "device"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-9s prefilter %s frames: %.1f us/frame, %.0f frames/s" % (kind, os.environ.get("PROBE_PREFILTER", "0"), 1e6 * dt / (nf - 24), (nf - 24) / dt))
    import ctypes as C
    ht = (C.c_double * 8)()
    lib.lib.ssf_dbg_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.lib.ssf_dbg_host_times(f.h, ht)
    nfh = max(ht[3], 1.0)
    print("          host thread, per frame: submit %.1f us | ICP loop %.1f us (until its first record %.1f) | association + fuse %.1f us; extract ready at activation %.0f %%"
          % (ht[0] / nfh, ht[1] / nfh, ht[5] / nfh, ht[2] / nfh, 100.0 * ht[4] / nfh))
    if kind != "device" and hasattr(lib.lib, "ssf_upload_stats"):
        st = (C.c_double * 6)()
        lib.lib.ssf_upload_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.lib.ssf_upload_stats(f.h, st)
        nfr = max(st[1], 1.0)
        print("          upload: %d workers, %d frames; per frame and worker-thread: ring wait %.0f us, staging memcpy %.0f us, two hipMemcpyAsync %.0f us; "
              "submitting thread waited %.0f us per frame for uploads" % (st[0], st[1], st[2] / nfr, st[3] / nfr, st[4] / nfr, st[5] / nfr))
    f.close()
