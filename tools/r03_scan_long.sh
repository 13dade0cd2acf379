#!/bin/bash
# usage (GPU box, repo root): tools/r03_scan_long.sh <tag> [variants...] -- the sampled single-pass scan on long chunks (c4) against k_adc_scan2
tag=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ $# -eq 0 ]; then set -- "CIS_SCAN_LONG=0" "CIS_SCAN_LONG=1"; fi
{
for v in "$@"; do
  echo "== c4 $v"
  env $v CIS_SCAN4_DEBUG=1 timeout 300 python bench.py --config c4 --steps 12 --no-cnn --no-pcie --no-cpu-baseline 2>&1 | grep "fall-back" | sed "s/.*k_adc_scan4: //" | sort | uniq -c | sort -rn | head -3
  env $v timeout 300 python bench.py --config c4 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py
done
echo "== c2"; timeout 300 python bench.py --config c2 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py
} > gpurun_out/${tag}_scan_long.txt 2>&1
cat gpurun_out/${tag}_scan_long.txt
