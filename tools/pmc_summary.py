#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV into per-kernel averages (one row per kernel).
usage: pmc_summary.py <counter_collection.csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "")
        if flt and flt not in k:
            continue
        k = k.split("(")[0][:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id"))
print("kernel,dispatches,counter,total,per_dispatch")
for k in sorted(acc):
    n = max(len(disp[k]), 1)
    for c in sorted(acc[k]):
        print("%s,%d,%s,%.6g,%.6g" % (k, n, c, acc[k][c], acc[k][c] / n))
