#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "$1 [$2] $3: "; env $2 CIS_LIB_PATH=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/$1 timeout 300 python bench.py --config $3 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-140; }
{
for rep in 1 2; do
run libcis_hip.so CIS_S4_BALANCE=1 c4
run libcis_hip.so CIS_S4_BALANCE=0 c4
run libcis_hip.so CIS_S4_BALANCE=1 c2
run libcis_hip.so CIS_S4_BALANCE=0 c2
done
} 2>&1 | tee gpurun_out/r04h_ab.txt
python tools/nq_sweep.py 2048,4096,6144,8192 2>&1 | grep "nq "
