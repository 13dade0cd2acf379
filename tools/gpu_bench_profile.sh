#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_bench_profile.sh <tag> [bench args]
# runs bench.py under rocprofv3 --kernel-trace --stats and leaves compact summaries in gpurun_out/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r -- python bench.py "$@" > gpurun_out/${tag}_bench.log 2>&1
cp /tmp/prof_$tag/r_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
grep '^{' gpurun_out/${tag}_bench.log | tail -1 | python tools/bench_summary.py
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv "^(void )?k_"
