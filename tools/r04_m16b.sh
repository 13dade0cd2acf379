#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "[$1] c3: "; env $1 timeout 400 python bench.py --config c3 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-230; }
{
for t in 1.1 1.25 1.4 1.6; do run "CIS_S4_M16=1 CIS_S4_SAT=$t"; CIS_S4_M16=1 CIS_S4_SAT=$t CIS_SCAN4_DEBUG=1 CIS_BENCH_PIPELINE=1 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie 2>&1 | grep "k_adc_scan4" | sort | uniq -c | sort -rn | head -2; done
run "CIS_S4_M16=0"
} 2>&1 | tee gpurun_out/r04v_m16b.txt
