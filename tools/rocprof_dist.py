"""Per-kernel duration distribution (min / median / p90 / max, us) from a rocprofv3 --kernel-trace results.db."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else None
if view is None:
    print("tables:", tables); sys.exit(0)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
rows = cur.execute("select name, start, end from %s order by start" % view).fetchall()
by = {}
for name, s, e in rows:
    by.setdefault(name[:60], []).append((e - s) / 1000.0)
for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    d2 = sorted(d)
    print("%-62s n=%4d min %7.1f med %7.1f p90 %7.1f max %7.1f" % (name, len(d), d2[0], d2[len(d) // 2], d2[int(0.9 * (len(d) - 1))], d2[-1]))
if len(sys.argv) > 2:
    # sequence of one kernel's durations in launch order
    print([round((e - s) / 1000.0, 1) for n, s, e in rows if sys.argv[2] in n])
