#!/bin/bash
# usage (GPU box): tools/pipeline_timeline.sh <config> [pipeline] -- ~1.3 ms of the pipelined bench in steady state, kernel by kernel with the queue
# (= lane) each ran on: what the chain of small kernels of batch b+1 does while batch b scans
cfg=${1:-c4}; P=${2:-3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pt; CIS_BENCH_MIN_REPS=3 CIS_BENCH_MIN_TIMED_S=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o r -- python bench.py --config $cfg --steps 30 --warmup 3 --no-cnn --no-pcie --no-cpu-baseline --no-c4x --pipeline $P > /dev/null 2>&1
python - <<PY
import csv, re, glob
f = glob.glob("/tmp/pt/**/r_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
scans = [i for i, r in enumerate(rows) if "k_adc_scan4" in r["Kernel_Name"] or "k_adc_scan2" in r["Kernel_Name"]]
best = None
for i in range(0, len(scans) - 29):
    span = int(rows[scans[i + 29]]["End_Timestamp"]) - int(rows[scans[i]]["Start_Timestamp"])
    if best is None or span < best[0]: best = (span, i)
a = scans[best[1] + 12]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
print("30 scans in %.1f us -> %.1f us per step" % (best[0] / 1e3, best[0] / 29e3))
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s > 1300000: break
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%8.1f .. %8.1f  (%6.1f)  lane %d  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, re.sub(r"\(.*", "", r["Kernel_Name"])[:44]))
PY
