#!/usr/bin/env python3
"""Per-kernel mean of each counter in rocprofv3 --pmc --output-format csv output.

    python profiles/pmc_summary.py gpurun_out/prof_x/pmc_fetch/*counter_collection.csv
"""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: [0, 0.0])
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = (row["Kernel_Name"][:72], row["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
    print("%-72s %-14s %8s %16s" % ("kernel", "counter", "calls", "mean_per_call"))
    for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %-14s %8d %16.1f" % (k, c, n, s / n))


if __name__ == "__main__":
    main(sys.argv[1:])
