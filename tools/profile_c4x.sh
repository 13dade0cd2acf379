#!/bin/bash
# usage (GPU box, repo root): tools/profile_c4x.sh <tag> [N] [nqs]  -- rocprofv3 kernel stats of tools/bench_c4x.py (the HBM-resident regime)
tag=$1; N=${2:-200000000}; nqs=${3:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r -- python tools/bench_c4x.py $N 0 $nqs > gpurun_out/${tag}_c4x_prof.log 2>&1
cp /tmp/prof_$tag/r_kernel_stats.csv gpurun_out/${tag}_c4x_kernel_stats.csv
grep "mode" gpurun_out/${tag}_c4x_prof.log
python tools/kstats.py gpurun_out/${tag}_c4x_kernel_stats.csv "stream|select_topl|emit_sorted|cand_|slots|item_|cell_scan|plan|front|tables|pca|seg_begin"
