#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output per kernel.
   pmc_summary.py <dir> -> prints mean counter value per (kernel, counter) from *counter_collection.csv
                           and avg duration per kernel from *kernel_trace.csv / *kernel_stats.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("fdmi::", "")[:90]


def main(d):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (short(row["Kernel_Name"]), row["Counter_Name"])
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
        print(f"# {f}")
        for (kn, cn), (tot, n) in sorted(acc.items()):
            print(f"{cn:16s} mean={tot / n:14.1f} n={n:5d}  {kn}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {f}")
        with open(f) as fh:
            for i, row in enumerate(csv.DictReader(fh)):
                if i >= 16:
                    break
                print(f"{short(row['Name']):80s} calls={row['Calls']:>7s} avg_ns={float(row['AverageNs']):12.0f} pct={row['Percentage']}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
