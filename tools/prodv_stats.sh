#!/bin/bash
# usage (GPU box): tools/prodv_stats.sh [V] [nq] -- per-kernel time of the quota-10000 batches at the release operating point
V=${1:-2048}; NQ=${2:-8192}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pv; PRODV_FAST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o r -- python tools/bench_prodv.py $V 10000000 $NQ 2>&1 | grep -v "^W\|amdgpu.ids" | tail -3
python tools/kstats.py /tmp/pv/r_kernel_stats.csv | grep -v "at::\|elementwise\|Cijk\|rocprim\|cdist\|index_add\|fillBuffer" | head -40
