#!/usr/bin/env python3
"""Per-kernel timeline of the LAST search step in a rocprofv3 kernel trace (r_kernel_trace.csv): the kernels between
the last two k_pca_gemm launches of the timed loop, in launch order, with durations and gaps."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts with the PCA GEMM of the query batch
starts = [i for i, n in enumerate(names) if "k_pca_gemm" in n]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = re.sub(r"\(.*", "", r["Kernel_Name"])[:48]
    print("%-48s start %8.1f us  dur %7.1f us  gap %6.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
    tot += e - s
print("step span %.1f us, kernel time %.1f us" % ((prev_end - t0) / 1e3, tot / 1e3))
