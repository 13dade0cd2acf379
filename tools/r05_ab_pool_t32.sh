#!/bin/bash
# usage (GPU box, repo root): tools/r05_ab_pool_t32.sh  -- A/B of the fused first layer + max pool (dlib) and of the scan without the f32 table copy
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_cnn_hip_parity.py -m gpu -x -q 2>&1 | tail -5
for v in 0 1; do
  echo "== CIS_CNN_NO_POOL7=$v"
  for i in 1 2 3; do CIS_CNN_NO_POOL7=$v python tools/bench_dlib.py 256 | grep batch; done
done
tools/dlib_timeline.sh 256 > gpurun_out/r05k_dlib_timeline.txt 2>&1; head -12 gpurun_out/r05k_dlib_timeline.txt; tail -1 gpurun_out/r05k_dlib_timeline.txt
CIS_NO_T32=1 python -m pytest tests/test_lopq_hip_parity.py tests/test_stream_route.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do
  for cfg in c4 c2; do
    echo "== CIS_NO_T32=$v $cfg"
    CIS_NO_T32=$v python bench.py --config $cfg --no-cnn --no-cpu-baseline --no-pcie --no-c4x --detail-file gpurun_out/r05k_t32_${v}_${cfg}.json 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05k_t32_${v}_${cfg}_line.json
    python tools/bench_summary.py gpurun_out/r05k_t32_${v}_${cfg}_line.json
  done
done
