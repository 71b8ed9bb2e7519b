#!/usr/bin/env python
"""Average PMC counter values per kernel name from rocprofv3 --pmc ... --output-format csv runs.
usage: python tools/pmc_kernel_avgs.py <dir> [substring filter]"""
import collections
import csv
import glob
import os
import sys


def main(d, filt=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if filt and filt not in k:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k in sorted(acc):
        print(k[:90])
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print("    %-28s avg %14.1f   (n=%d)" % (c, s / n, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
