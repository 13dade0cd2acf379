#!/usr/bin/env python3
"""Print a compact view of a rocprofv3 *_kernel_stats.csv: short kernel name, calls, avg/min/max us."""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = r["Name"]
    if flt and not re.search(flt, name):
        continue
    short = re.sub(r"\(.*", "", name)[:60]
    print("%-60s calls %5s avg %9.1f us  min %9.1f  max %9.1f  pct %s" % (
        short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
