#!/bin/bash
# PCA GEMM with the loads-only fetch (k_pca_gemm_mfma_pf) against the forms before it: encode per call, C3 bench, parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== CIS_PCA_PF=0"; CIS_PCA_PF=0 python tools/enc_calls.py
echo "== CIS_PCA_PF=1"; python tools/enc_calls.py
for e in "CIS_PCA_PF=0" "CIS_PCA_PF=1"; do
  for c in c3 c2; do echo -n "[$e] $c: "; env $e timeout 600 python bench.py --config $c --steps 30 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py | cut -c1-260; done
done
timeout 1500 python -m pytest tests/test_lopq_hip_parity.py tests/test_full_size_properties.py tests/test_cnn_hip_parity.py -m gpu -x -q 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r04z_pca_pf.txt
