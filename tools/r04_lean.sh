#!/bin/bash
# lean tables (no float64 tables on the fixed-point scan's route) against the full tables: C4 / C2 / C3, serial and pipelined
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "[$1] $2: "; env $1 timeout 600 python bench.py --config $2 --steps 30 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-260; }
{
for c in c4 c2 c3; do
for e in "CIS_LEAN_TABLES=0 CIS_BENCH_PIPELINE=1" "CIS_LEAN_TABLES=1 CIS_BENCH_PIPELINE=1" "CIS_LEAN_TABLES=0" "CIS_LEAN_TABLES=1"; do run "$e" $c; done
done
timeout 1200 python -m pytest tests/test_lopq_hip_parity.py tests/test_full_size_properties.py -m gpu -x -q 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r04x_lean.txt
