#!/bin/bash
# usage (GPU box, repo root): tools/r03_ab_libs.sh <tag> <config> lib1.so lib2.so ... -- the same bench line on several builds of the library, twice each
tag=$1; cfg=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for rep in 1 2; do for l in "$@"; do
  echo -n "$l: "; CIS_LIB_PATH=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/$l timeout 300 python bench.py --config $cfg --steps 20 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py
done; done
} > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
