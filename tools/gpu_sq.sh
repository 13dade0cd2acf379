#!/bin/bash
# usage (GPU box, repo root): tools/gpu_sq.sh <tag> <config> [env assignments...]   -> SQ counter passes of the scan kernel
tag=$1; cfg=$2; shift; shift
for e in "$@"; do export "$e"; done
A="--config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie"
tools/gpu_pmc.sh ${tag}_sqa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $A | grep adc_scan
tools/gpu_pmc.sh ${tag}_sqb "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" $A | grep adc_scan
tools/gpu_pmc.sh ${tag}_sqc "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" $A | grep adc_scan
