#!/usr/bin/env python
"""Mean value per launch of every counter in a rocprofv3 --pmc CSV directory, per kernel (short names).
  python tools/pmc_counters.py <dir> [kernel substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows_of(d):
    """(kernel name, counter name, value) of every dispatch: rocprofv3 CSV output or its rocpd sqlite database"""
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        for name, counter, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            yield name, counter, float(value)


def main(d, filt):
    acc = defaultdict(lambda: defaultdict(list))
    for name, counter, value in rows_of(d):
        nm = name.split("(")[0].replace("void ", "").replace("ssf::", "")
        if filt and not any(x in nm for x in filt):
            continue
        acc[nm][counter].append(value)
    for nm in sorted(acc):
        n = max(len(v) for v in acc[nm].values())
        print("%s  (%d launches)" % (nm, n))
        for c in sorted(acc[nm]):
            v = acc[nm][c]
            print("    %-28s %14.1f" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
