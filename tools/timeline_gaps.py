"""One frame of a ONE-FRAME-IN-FLIGHT run as a timeline: every kernel of the frame with its start, duration and the idle gap in front
of it (rocprofv3 --kernel-trace results.db).  Frames are cut at the first extract kernel (k_ingest*); the median frame by length
of the second half of the run is printed, and the per-stage sums of kernel time and of gaps over all those frames.
    python tools/timeline_gaps.py <results.db>"""
import sqlite3
import sys
import statistics as st

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("ssf::", "").replace("void ", "")[:34]
cuts = [i for i, (n, s, e) in enumerate(rows) if "k_ingest" in n]
frames = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
frames = frames[len(frames) // 2:]
if not frames:
    print("no frames found; kernels:", sorted(set(short(r[0]) for r in rows))[:40]); sys.exit(0)
length = lambda f: (f[-1][2] - f[0][1]) / 1000.0
period = [(b[0][1] - a[0][1]) / 1000.0 for a, b in zip(frames[:-1], frames[1:])]
print("%d frames; frame period (first extract kernel to the next frame's) median %.1f us, min %.1f, max %.1f" % (len(frames), st.median(period), min(period), max(period)))
ksum = [sum((e - s) for n, s, e in f) / 1000.0 for f in frames]
print("kernel time per frame median %.1f us -> idle between kernels %.1f us" % (st.median(ksum), st.median(period) - st.median(ksum)))
mid = sorted(range(len(frames) - 1), key=lambda i: period[i])[len(period) // 2]
f = frames[mid]; nxt = frames[mid + 1][0][1]
print("the median frame (%d kernels):   start us | duration us | gap in front us | kernel" % len(f))
t0 = f[0][1]; prev = None
for n, s, e in f:
    print("   %8.1f %7.1f %7.1f   %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, 0.0 if prev is None else (s - prev) / 1000.0, short(n)))
    prev = e
print("   %8.1f                   (next frame's first kernel: gap %.1f)" % ((nxt - t0) / 1000.0, (nxt - prev) / 1000.0))
# gaps by the kernel they precede, summed per frame (median over frames)
by = {}
for f in frames[:-1]:
    acc = {}
    prev = None
    for n, s, e in f:
        k = short(n)
        a = acc.setdefault(k, [0.0, 0.0, 0])
        a[0] += (e - s) / 1000.0; a[2] += 1
        if prev is not None:
            a[1] += (s - prev) / 1000.0
        prev = e
    for k, a in acc.items():
        by.setdefault(k, []).append(a)
print("per frame, median over frames:   launches | kernel time us | gaps in front us")
for k, v in sorted(by.items(), key=lambda kv: -st.median([a[0] + a[1] for a in kv[1]])):
    print("   %-36s %4d %8.1f %8.1f" % (k, round(st.median([a[2] for a in v])), st.median([a[0] for a in v]), st.median([a[1] for a in v])))
