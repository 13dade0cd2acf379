#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "$1 [$2] $3: "; env $2 CIS_LIB_PATH=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/$1 timeout 300 python bench.py --config $3 --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-140; }
{
for rep in 1 2; do
run libcis_hip.so CIS_S4_NWL=8 c4
run libcis_wpe8_5.so CIS_S4_NWL=8 c4
run libcis_wpe8_4.so CIS_S4_NWL=8 c4
run libcis_hip.so CIS_S4_NWL=4 c4
run libcis_hip.so CIS_S4_NWS=4 c2
run libcis_hip.so CIS_S4_NWS=8 c2
done
} 2>&1 | tee gpurun_out/r04f_ab.txt
python tools/nq_sweep.py 4096,8192 2>&1 | grep "nq "
CIS_S4_NWS=8 timeout 1200 python -m pytest tests/test_full_size_properties.py tests/test_lopq_hip_parity.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04f_pytest.txt
