#!/bin/bash
# C3 (M = 16) through k_adc_scan4 with the saturating scale: sweep of the sample target, then parity with it on
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "[$1] c3: "; env $1 timeout 400 python bench.py --config c3 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-230; }
{
run "CIS_BENCH_PIPELINE=1"
for t in 0.5 0.65 0.8 0.95; do run "CIS_BENCH_PIPELINE=1 CIS_S4_M16=1 CIS_S4_SAT=$t"; CIS_S4_M16=1 CIS_S4_SAT=$t CIS_SCAN4_DEBUG=1 CIS_BENCH_PIPELINE=1 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie 2>&1 | grep "k_adc_scan4" | sort | uniq -c | sort -rn | head -3; done
run "CIS_S4_M16=1"
CIS_S4_M16=1 timeout 900 python -m pytest tests/test_lopq_hip_parity.py tests/test_full_size_properties.py -m gpu -x -q 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r04v_m16.txt
