#!/bin/bash
# usage (GPU box, repo root): tools/r05_cnn_sweep.sh -- DeepSentibank: first layer from the NCHW planes, fc tile shapes / split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_cnn_hip_parity.py -m gpu -x -q 2>&1 | tail -3
run() { echo "== $*"; env "$@" tools/cnn_timeline.sh 2>&1 | grep -E "batch|conv_igemm|nchw|total" | awk '{printf "%s | ", $0} END {print ""}' | sed "s/void k_conv_igemm//g; s/ us  blocks / /g; s/  */ /g"; }
run CIS_CNN_NHWC_FIRST=1 CIS_CNN_FC_TILE=0 CIS_CNN_FC_SPLITK=4
run CIS_CNN_NHWC_FIRST=1
run CIS_X=0
run CIS_CNN_FC_TILE=3
for i in 1 2 3; do python tools/bench_cnn.py | grep batch; done
for i in 1 2 3; do python tools/bench_dlib.py | grep batch; done
