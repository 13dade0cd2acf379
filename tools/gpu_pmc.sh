#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> "<counters>" [bench args]  -> gpurun_out/<tag>_pmc.csv
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o r -- python bench.py "$@" > gpurun_out/${tag}_pmc_bench.log 2>&1
python tools/pmc_summary.py /tmp/pmc_$tag/r_counter_collection.csv > gpurun_out/${tag}_pmc.csv
grep -E "adc_scan" gpurun_out/${tag}_pmc.csv
