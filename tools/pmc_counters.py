#!/usr/bin/env python
"""Mean value per launch of every counter in a rocprofv3 --pmc CSV directory, per kernel (short names).
  python tools/pmc_counters.py <dir> [kernel substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, filt):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ssf::", "")
                if filt and not any(x in nm for x in filt):
                    continue
                acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for nm in sorted(acc):
        n = max(len(v) for v in acc[nm].values())
        print("%s  (%d launches)" % (nm, n))
        for c in sorted(acc[nm]):
            v = acc[nm][c]
            print("    %-28s %14.1f" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
