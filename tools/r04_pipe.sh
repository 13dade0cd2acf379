#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lopq_hip_parity.py -q -m gpu -x -k "views" 2>&1 | tail -5
run() { echo -n "[$1] $2: "; env $1 timeout 300 python bench.py --config $2 --steps 40 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-260; }
{
for rep in 1 2; do
run CIS_BENCH_PIPELINE=1 c4
run CIS_BENCH_PIPELINE=2 c4
run CIS_BENCH_PIPELINE=3 c4
run CIS_BENCH_PIPELINE=1 c2
run CIS_BENCH_PIPELINE=2 c2
run CIS_BENCH_PIPELINE=3 c2
run CIS_BENCH_PIPELINE=4 c4
run CIS_BENCH_PIPELINE=4 c2
run CIS_BENCH_PIPELINE=3 c3
done
} 2>&1 | tee gpurun_out/r04j_pipeline_ab.txt
