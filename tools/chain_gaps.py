"""The serial track chain of a PIPELINED run as a timeline (rocprofv3 --kernel-trace results.db): only the track stream's kernels
(k_icp, k_match, k_update_insert, k_move_rows), cut into frames at k_update_insert; per frame the kernels' durations, the idle gaps
between them (launch latency the host did not hide) and the period.  Medians over the second half of the run.
CAVEAT (gpurun r15b): under rocprofv3 every launch call is slower, so a launch made ahead reaches the queue late -- the 3.9 us gaps this
prints in front of ICP iterations 3+ and the 8.8 us in front of the first are the tracer's, not the library's: its own untraced marks
(bench.py "pipeline_fill": entry -> first record 19 us = rest of the row moves + launch boundary + kernel; 8.4 us per further iteration =
~4 us of kernel behind the word + the host's round trip, the launch boundary hidden inside it) leave no idle gap to take out.
    python tools/chain_gaps.py <results.db>"""
import sqlite3
import sys
import statistics as st

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
track = [(n, s, e) for n, s, e in rows if any(k in n for k in ("k_icp", "k_match", "k_update_insert", "k_move_rows"))]
short = lambda n: n.split("(")[0].replace("ssf::", "").replace("void ", "")[:28]
frames, cur_f = [], []
for n, s, e in track:
    cur_f.append((n, s, e))
    if "k_move_rows" in n:
        frames.append(cur_f); cur_f = []
frames = frames[len(frames) // 2:]
if len(frames) < 4:
    print("too few frames"); sys.exit(0)
period = [(b[-1][2] - a[-1][2]) / 1000.0 for a, b in zip(frames[:-1], frames[1:])]
print("%d frames; period (end of the row moves to the next end) median %.1f us (min %.1f, p90 %.1f)" % (len(frames), st.median(period), min(period), sorted(period)[int(0.9 * len(period))]))
kern = [sum(e - s for n, s, e in f) / 1000.0 for f in frames[1:]]
print("track kernels per frame median %.1f us -> the track stream idle %.1f us per frame" % (st.median(kern), st.median(period) - st.median(kern)))
# by position in the frame: k_icp #1..#n, update_insert, move_rows; gap in front of each
pos = {}
for a, f in zip(frames[:-1], frames[1:]):
    prev = a[-1][2]
    ni = 0
    for n, s, e in f:
        key = short(n)
        if "k_icp" in n:
            ni += 1; key = "k_icp #%d" % ni
        pos.setdefault(key, []).append(((e - s) / 1000.0, (s - prev) / 1000.0))
        prev = e
print("   kernel                         frames   duration us (median)   gap in front us (median / p90)")
for k, v in pos.items():
    d = [x[0] for x in v]; g = sorted(x[1] for x in v)
    print("   %-30s %5d   %8.1f   %8.1f / %.1f" % (k, len(v), st.median(d), st.median(g), g[int(0.9 * (len(g) - 1))]))
