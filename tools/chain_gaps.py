"""Track chain of a rocprofv3 --kernel-trace results.db: mean duration of each kernel of the chain and mean idle gap
between consecutive chain kernels (end of one -> start of the next), by transition, over the steady-state half."""
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
short = {"k_icp": "icp", "k_match": "match", "k_update_insert": "fuse", "k_move_rows": "move"}
chain = []
for n, s, e in rows:
    for k, v in short.items():
        if k in n:
            chain.append((v, s, e))
chain = chain[len(chain) // 2:]
dur, gap = defaultdict(list), defaultdict(list)
for i, (v, s, e) in enumerate(chain):
    dur[v].append((e - s) / 1e3)
    if i:
        gap[chain[i - 1][0] + "->" + v].append((s - chain[i - 1][2]) / 1e3)
nframes = len(dur["fuse"])
tot = 0.0
for v, d in dur.items():
    print("%-6s n=%4d  mean %6.1f us  per frame %6.1f us" % (v, len(d), sum(d) / len(d), sum(d) / nframes)); tot += sum(d) / nframes
for t, g in sorted(gap.items()):
    print("gap %-12s n=%4d  mean %6.1f us  per frame %6.1f us" % (t, len(g), sum(g) / len(g), sum(g) / nframes)); tot += sum(g) / nframes
print("frames %d, chain total per frame %.1f us (wall %.1f us)" % (nframes, tot, (chain[-1][2] - chain[0][1]) / 1e3 / nframes))
