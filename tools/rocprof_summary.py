"""Dump the per-kernel statistics of a rocprofv3 --kernel-trace --stats results.db as a text table
(profiles/*.txt are produced with this from gpurun_out/)."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % note)
        f.write("# %-90s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            f.write("%-92s %8d %14.3f %12.3f %7.2f\n" % (name[:92], calls, tot / 1000.0 if tot > 1e7 else tot, avg / 1000.0 if tot > 1e7 else avg, pct))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
