#!/bin/bash
# usage (GPU box): tools/prodv_pmc.sh <tag> -- SQ counter passes of k_tiny_select at the release operating point (V = 2048, 8192 queries)
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
pass() {
  rm -rf /tmp/pvp
  PRODV_FAST=1 timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pvp -o r -- python tools/bench_prodv.py 2048 10000000 8192 > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pvp/r_counter_collection.csv | grep -E "^kernel,|k_tiny_select|k_plan_par|k_tables_group|k_rank_sort" > gpurun_out/${tag}_$1_pmc.csv
  cat gpurun_out/${tag}_$1_pmc.csv
}
pass sqa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pass sqb "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
pass sqc "SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
