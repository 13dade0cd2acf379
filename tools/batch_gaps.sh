#!/bin/bash
# usage (GPU box): tools/batch_gaps.sh [config] -- kernel timeline of ONE timed batch (rocprofv3 kernel trace): durations and the idle gaps between kernels
cfg=${1:-c4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; CIS_BENCH_MIN_REPS=2 CIS_BENCH_MIN_TIMED_S=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python bench.py --config $cfg --steps 6 --warmup 2 --no-cnn --no-pcie --no-cpu-baseline --no-c4x --pipeline 1 > /dev/null 2>&1
python - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/kt/r_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_pca_gemm" in n]
from collections import Counter
lens = Counter(b - a for a, b in zip(starts[:-1], starts[1:]))
print("kernels between consecutive k_pca_gemm launches:", dict(lens), "files:", len(rows))
cands = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if any("k_adc_scan" in n for n in names[a:b]) and b - a < 60]
a, b = cands[len(cands) // 2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; tot = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
    print("%-44s start %7.1f  dur %6.1f  gap %5.1f" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    gaps += max(0, s - prev_end); prev_end = max(prev_end, e); tot += e - s
nxt = int(rows[b]["Start_Timestamp"])
print("batch: %d kernels, kernel time %.1f us, gaps inside %.1f us, until the next batch starts %.1f us" % (b - a, tot / 1e3, gaps / 1e3, (nxt - t0) / 1e3))
PY
