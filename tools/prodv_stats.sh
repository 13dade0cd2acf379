#!/bin/bash
# usage (GPU box): tools/prodv_stats.sh [V=4096] -- rocprofv3 kernel statistics of tools/bench_prodv.py at a release operating point
V=${1:-4096}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_pv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pv -o r -- python tools/bench_prodv.py $V 10000000 8192 > /tmp/pv.log 2>&1
grep "quota\|flight" /tmp/pv.log | cut -c1-200
python tools/kstats.py /tmp/prof_pv/r_kernel_stats.csv | head -30
