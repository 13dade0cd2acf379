#!/bin/bash
# usage (GPU box): tools/overlap_trace.sh <config> [pipeline] -- kernel trace of the pipelined bench: how busy is the GPU, which kernels overlap
cfg=${1:-c4}; P=${2:-3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/ot; CIS_BENCH_PIPELINE=$P timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ot -o r -- python bench.py --config $cfg --steps 30 --warmup 3 --no-cnn --no-pcie --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, re, glob, collections
f = glob.glob("/tmp/ot/**/r_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: 30 steps -> the 30 consecutive k_pca_gemm launches with the smallest spacing; take the window between the 10th and 25th scan4/scan2 launch of the densest stretch
scans = [r for r in rows if "k_adc_scan4" in r["Kernel_Name"] or "k_adc_scan2" in r["Kernel_Name"]]
# find the stretch of 30 scans with the smallest span
best = None
for i in range(0, len(scans) - 29):
    span = int(scans[i + 29]["End_Timestamp"]) - int(scans[i]["Start_Timestamp"])
    if best is None or span < best[0]: best = (span, i)
i0 = best[1]
t0, t1 = int(scans[i0 + 5]["Start_Timestamp"]), int(scans[i0 + 25]["Start_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.Counter(); cnt = collections.Counter()
for r in win:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
wall = t1 - t0
print("window: 20 batches, %.1f us per batch; some kernel running %.1f %% of the time; sum of kernel durations / wall = %.2f" % (wall / 20e3, 100.0 * busy / wall, sum(tot.values()) / wall))
for n, t in tot.most_common(14):
    print("  %-40s %6.1f us per batch (avg %.1f us x %.1f)" % (n, t / 20e3, t / cnt[n] / 1e3, cnt[n] / 20.0))
PY
