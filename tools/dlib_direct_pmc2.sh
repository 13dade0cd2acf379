#!/bin/bash
# second counter set for k_conv3x3_direct (vector-memory latency, texture-addresser load, workgroup-launch stalls); separate passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum" "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_CSN_BUSY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/q$i
  timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/q$i -o r -- python tools/bench_dlib.py > /tmp/q$i.log 2>&1
  python tools/pmc_summary.py /tmp/q$i/r_counter_collection.csv 2>/dev/null | grep "direct" || tail -3 /tmp/q$i.log
done
