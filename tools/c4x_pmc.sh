#!/bin/bash
# usage (GPU box, repo root): tools/c4x_pmc.sh <tag> [N]  -- PMC passes of k_adc_stream (exhaustive queries over the c4x index)
#   -> gpurun_out/<tag>_c4x_nq{1,2}_{FETCH_SIZE,WRITE_SIZE,sq}_pmc.csv and gpurun_out/scan_traffic_c4x.json (copy both to profiles/)
tag=$1; N=${2:-200000000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nq in 1 2 4; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"; do
    name=$c; [ "${c:0:3}" = "SQ_" ] && name=sq
    rm -rf /tmp/pmc_c4x
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_c4x -o r -- python tools/stream_pmc_driver.py $N $nq 8 > gpurun_out/${tag}_c4x_nq${nq}_${name}.log 2>&1
    python tools/pmc_summary.py /tmp/pmc_c4x/r_counter_collection.csv | grep -E "^kernel|k_adc_stream" > gpurun_out/${tag}_c4x_nq${nq}_${name}_pmc.csv
  done
  # the un-instrumented kernel time of the same driver (rocprofv3 kernel trace only)
  rm -rf /tmp/pmc_c4x
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_c4x -o r -- python tools/stream_pmc_driver.py $N $nq 8 > gpurun_out/${tag}_c4x_nq${nq}_trace.log 2>&1
  grep -E "Name|k_adc_stream" /tmp/pmc_c4x/r_kernel_stats.csv > gpurun_out/${tag}_c4x_nq${nq}_kernel_stats.csv
done
python tools/c4x_traffic.py $tag $N
