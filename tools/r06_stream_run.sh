#!/bin/bash
# usage (GPU box, repo root): tools/r06_stream_run.sh TAG -- stream-route tests, the library's route on the 200 M index (whole call + kernel launch
# time from the library's marks, then rocprofv3 kernel statistics of the same driver), the probe's best forms on the same box
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_stream_route.py -x -q -m gpu 2>&1 | tail -3
python tools/r06_stream_lib.py 200000000 ${NQS:-1,2,4,8} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_stream_lib.txt
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o r -- python tools/r06_stream_lib.py 200000000 1,2,4 > /tmp/s.log 2>&1
python tools/kstats.py /tmp/prof_s/r_kernel_stats.csv "stream|select_topl|plan|tables|slots|item_|cell_scan|cand_|front|pca|emit|copy_visited|seg_begin" | tee gpurun_out/${TAG}_stream_kernels.txt
tools/probes/stream_probe 200000000 8 2>&1 | tee gpurun_out/${TAG}_probe.txt
