#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs:  python tools/pmc_summary.py DIR [kernel-substring ...]"""
import collections
import csv
import glob
import sys


def main(d, filters):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            if filters and not any(x in k for x in filters):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print("    %-40s %16.1f  (avg of %d)" % (c, acc[k][c] / calls[k][c], calls[k][c]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
