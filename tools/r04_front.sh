#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "[$1] $2: "; env $1 timeout 300 python bench.py --config $2 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-260; }
{
for rep in 1 2; do
run A=1 c4
run CIS_NO_FUSED_FRONT=1 c4
run A=1 c2
run CIS_NO_FUSED_FRONT=1 c2
done
} 2>&1 | tee gpurun_out/r04i_ab.txt
timeout 1500 python -m pytest tests/test_full_size_properties.py tests/test_lopq_hip_parity.py tests/test_index_insert.py tests/test_reference_surfaces.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04i_pytest.txt
tools/batch_gaps.sh c4 > gpurun_out/r04i_batch_timeline_c4.txt 2>&1; tail -22 gpurun_out/r04i_batch_timeline_c4.txt
