#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_cnn.sh <tag>  -> gpurun_out/<tag>_cnn_mfma_pmc.csv
# MFMA pipe occupancy of the CNN kernels from the PMC counters: SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CU_CYCLES
# (separate pass, kernel trace only), for tools/bench_cnn.py (DeepSentibank) and tools/bench_dlib.py
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for net in cnn dlib; do
  rm -rf /tmp/pmc_${tag}_$net
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$net -o r -- python tools/bench_$net.py > gpurun_out/${tag}_${net}_pmc.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_${tag}_$net/r_counter_collection.csv > gpurun_out/${tag}_${net}_mfma_pmc.csv
  grep -E "conv_igemm|Kernel" gpurun_out/${tag}_${net}_mfma_pmc.csv | head -40
done
