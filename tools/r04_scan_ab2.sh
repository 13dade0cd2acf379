#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/r03_ab_libs.sh r04e_c4 c4 libcis_hip.so libcis_pf3.so libcis_pf4.so libcis_nodefer.so
tools/r03_ab_libs.sh r04e_c2 c2 libcis_hip.so libcis_pf3.so libcis_pf4.so libcis_nodefer.so
python tools/nq_sweep.py 4096,8192 2>&1 | grep "nq "
timeout 1200 python -m pytest tests/test_full_size_properties.py tests/test_lopq_hip_parity.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r04e_pytest.txt
