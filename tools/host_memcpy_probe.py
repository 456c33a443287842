"""Host memcpy rate as the upload workers see it: 2.1 MB frames out of a working set far larger than the caches into a page-locked
slot, 1 / 2 / 4 threads (numpy copies release the GIL).  Beside tools/h2d_probe.py and tools/host_buffer_probe.py it says which side
bounds a sequence of HOST frames on the box at hand.  python tools/host_memcpy_probe.py"""
import threading, time
import numpy as np
import torch
NB = 2150400
src = [np.random.default_rng(i).integers(0, 255, NB, dtype=np.uint8) for i in range(64)]
def run(nth, reps=400):
    dst = [torch.empty(NB, dtype=torch.uint8).pin_memory().numpy() for _ in range(nth)]
    def work(t):
        for r in range(t, reps, nth): np.copyto(dst[t], src[r % 64])
    th = [threading.Thread(target=work, args=(t,)) for t in range(nth)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    return NB * reps / dt / 1e9
for nth in (1, 2, 4, 8):
    print("%d thread(s): %5.1f GB/s = %6.0f frames/s of 2.1 MB" % (nth, run(nth), run(nth) * 1e9 / NB))
