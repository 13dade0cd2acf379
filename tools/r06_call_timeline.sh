#!/bin/bash
# usage (GPU box): tools/r06_call_timeline.sh [N] -- kernel timeline (rocprofv3 kernel trace) of ONE exhaustive single-query call and ONE quota-10000
# single-query call on the N-vector C4-model index: durations and the idle gaps between kernels
N=${1:-200000000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python tools/r06_stream_lib.py $N 1 > /dev/null 2>&1
python - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/kt/r_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "k_pca_gemm" in n or "k_pca_small" in n]
def show(a, b, title):
    print("==", title)
    t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; tot = 0; gaps = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        short = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
        print("%-44s start %7.1f  dur %6.1f  gap %5.1f" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        gaps += max(0, s - prev_end); prev_end = max(prev_end, e); tot += e - s
    print("call: %d kernels, kernel time %.1f us, gaps %.1f us, span %.1f us" % (b - a, tot / 1e3, gaps / 1e3, (prev_end - t0) / 1e3))
# calls whose stream kernel ran long (exhaustive) / short (quota 10000): take the last of each kind
ex = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a < 40 and any("k_adc_stream" in n and "false" in n and int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > 200000 for i, n in enumerate(names[a:b], a))]
sh = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a < 40 and any("k_adc_stream" in n and "false" in n and int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 100000 for i, n in enumerate(names[a:b], a))]
if ex: show(*ex[-2 if len(ex) > 1 else -1], "exhaustive single query")
if sh: show(*sh[-2 if len(sh) > 1 else -1], "single query, quota 10000")
PY
