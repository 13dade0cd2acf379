#!/bin/bash
# usage (GPU box): tools/r06_single_timeline.sh -- kernel timeline of one single-query call (quota 10000) on the 10 M C4 index
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/r06_single_query.py 10000000 1,2,8 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python tools/r06_single_query.py 10000000 1 > /dev/null 2>&1
python - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/kt/r_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_pca_gemm" in n or "k_pca_small" in n]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; tot = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
    print("%-44s start %7.1f  dur %6.1f  gap %5.1f" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    gaps += max(0, s - prev_end); prev_end = max(prev_end, e); tot += e - s
print("call: %d kernels, kernel time %.1f us, gaps %.1f us, span %.1f us" % (b - a, tot / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
PY
