#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "$1 [$2] $3: "; env $2 CIS_LIB_PATH=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/$1 timeout 300 python bench.py --config $3 --steps 30 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-200; }
{
for rep in 1 2; do
for l in libcis_hip.so libcis_gl5.so libcis_gl4.so libcis_nogl.so; do
run $l CIS_BENCH_PIPELINE=1 c4
run $l CIS_BENCH_PIPELINE=3 c4
done; done
} 2>&1 | tee gpurun_out/r04s_glists_ab.txt
timeout 1200 python -m pytest tests/test_full_size_properties.py tests/test_lopq_hip_parity.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04s_pytest.txt
