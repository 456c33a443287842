"""Summary of the memory copies in a rocprofv3 results.db (tools/upload_trace.sh): count, bytes, duration, rate; and the timeline of
one stretch of the run (copies and kernels by queue)."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if "memory_copy" in t and "rocpd" in t]
print("tables:", mc)
t = mc[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
print(cols)
rows = cur.execute("select start, end, size from %s order by start" % t).fetchall()
import collections
by = collections.defaultdict(list)
for s, e, sz in rows: by[sz].append((e - s) / 1e3)
for sz, d in sorted(by.items()):
    d.sort()
    print("%9d bytes: n %5d  median %7.1f us  p90 %7.1f us  -> %5.1f GB/s at the median" % (sz, len(d), d[len(d) // 2], d[int(len(d) * 0.9)], sz / d[len(d) // 2] / 1e3))
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
ker = cur.execute("select d.start, d.end, s.kernel_name, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
big = [r for r in rows if r[2] > 500000]
if big:
    t0 = big[len(big) // 2][0]
    ev = [(s, e, "copy %d" % sz, -1) for s, e, sz in rows if t0 <= s < t0 + 3.0e6] + [(s, e, n.split("(")[0][-28:], q) for s, e, n, q in ker if t0 <= s < t0 + 3.0e6]
    for s, e, n, q in sorted(ev)[:400]:
        print("%9.1f  q%-2d %-30s %7.1f us" % ((s - t0) / 1e3, q, n, (e - s) / 1e3))
