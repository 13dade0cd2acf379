cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CIS_CNN_DIRECT_CFG=${CIS_CNN_DIRECT_CFG:-0}
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/p$i
  timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/p$i -o r -- python tools/bench_dlib.py > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/p$i/r_counter_collection.csv | grep "direct"
done
