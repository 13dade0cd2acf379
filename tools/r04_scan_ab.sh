#!/bin/bash
# round 4: deferred per-query split of k_adc_scan4 against the per-row split (libcis_nodefer.so), c4 and c2, then the route-parity tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/r03_ab_libs.sh r04b_c4 c4 libcis_hip.so libcis_nodefer.so
tools/r03_ab_libs.sh r04b_c2 c2 libcis_hip.so libcis_nodefer.so
echo "== fall-backs"; for c in c4 c2; do CIS_SCAN4_DEBUG=1 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie 2>&1 | grep "k_adc_scan4" | sort | uniq -c | sort -rn | head -3; done
timeout 1200 python -m pytest tests/test_full_size_properties.py tests/test_lopq_hip_parity.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r04b_pytest.txt
