#!/usr/bin/env python
"""Overlap analysis of a rocprofv3 --kernel-trace CSV (kernel_trace.csv): per-queue busy time,
union busy time and mean concurrency over the steady-state window.
  python tools/timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict


def main(path, skip=0.4):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
    rows = [r for r in rows if r[0] >= t_lo]
    span = rows[-1][1] - rows[0][0]
    per_q = defaultdict(int)
    per_k = defaultdict(lambda: [0, 0])
    ev = []
    for s, e, n, q, st in rows:
        per_q[(q, st)] += e - s
        k = n.split("(")[0][:60]
        per_k[k][0] += e - s; per_k[k][1] += 1
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0; conc_time = defaultdict(int); cur = 0; last = ev[0][0]
    for t, d in ev:
        if cur > 0:
            busy += t - last
        conc_time[cur] += t - last
        cur += d; last = t
    print("window %.3f ms, %d kernels; union busy %.1f%%, sum of durations / window = %.2f" %
          (span / 1e6, len(rows), 100.0 * busy / span, sum(per_q.values()) / span))
    for c in sorted(conc_time):
        print("  %d kernels in flight: %.1f%% of the window" % (c, 100.0 * conc_time[c] / span))
    for q in sorted(per_q):
        print("  queue/stream %s: busy %.1f%%" % (q, 100.0 * per_q[q] / span))
    for k, (t, n) in sorted(per_k.items(), key=lambda kv: -kv[1][0])[:12]:
        print("  %-60s %7d calls  avg %8.2f us  %5.1f%% of window" % (k, n, t / n / 1e3, 100.0 * t / span))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
