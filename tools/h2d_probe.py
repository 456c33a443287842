"""Host-to-device copy rate of page-locked frames as the upload ring makes them (hipMemcpyAsync on a compute stream), by copy
size and number of streams -- what bounds a sequence of HOST frames.  python tools/h2d_probe.py"""
import time
import torch
dev = torch.device("cuda", 0)
def rate(nbytes, streams, reps=200):
    src = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    dst = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(streams)]
    st = [torch.cuda.Stream(dev) for _ in range(streams)]
    for s in range(streams):
        with torch.cuda.stream(st[s]): dst[s].copy_(src[s], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for s in range(streams):
            with torch.cuda.stream(st[s]): dst[s].copy_(src[s], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return nbytes * reps * streams / dt / 1e9, dt / (reps * streams) * 1e6
for nb in (921600, 1228800, 2150400, 8601600, 17203200, 68812800):
    print("%9d bytes: " % nb + "  ".join("%d stream(s) %5.1f GB/s (%6.1f us per copy)" % ((s,) + rate(nb, s)) for s in (1, 2, 3)))
