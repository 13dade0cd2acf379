#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc_cnn.sh <tag>  -> gpurun_out/<tag>_{cnn,dlib}_mfma_pmc.csv, gpurun_out/<tag>_mfma_utilisation.txt
# MFMA pipe counters of the CNN forwards (their own rocprofv3 --pmc pass, kernel trace only) against the wall clock of the same command
# without the profiler: tools/mfma_pmc_summary.py explains the three figures.  dlib is profiled as ONE chain (CIS_CNN_PARTS=1): under
# counter collection the launches are serialised, the default two half-batch chains would each be measured alone at half the batch.
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${tag}_mfma_utilisation.txt; : > $out
for net in cnn dlib; do
  mac=720310816; [ $net = dlib ] && mac=270854144
  export CIS_CNN_PARTS=1
  wall=$(python tools/bench_$net.py 256 | grep batch | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  rm -rf /tmp/pmc_${tag}_$net
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$net -o r -- python tools/bench_$net.py 256 > gpurun_out/${tag}_${net}_pmc.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_${tag}_$net/r_counter_collection.csv > gpurun_out/${tag}_${net}_mfma_pmc.csv
  unset CIS_CNN_PARTS
  wall2=$(python tools/bench_$net.py 256 | grep batch | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  { echo "== tools/bench_$net.py 256, one chain (CIS_CNN_PARTS=1); 7 forwards profiled =="; python tools/mfma_pmc_summary.py gpurun_out/${tag}_${net}_mfma_pmc.csv 7 $mac 256 $wall
    echo "default configuration (dlib: two half-batch chains on two streams) wall $wall2 ms: (a) = $(python -c "print('%.3f' % (2.0*$mac*256/($wall2*1e-3)/157.3e12))")"; echo; } >> $out
done
cat $out
