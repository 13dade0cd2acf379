#!/bin/bash
# usage (GPU box, repo root): tools/r05_stream_ab.sh -- k_adc_stream variants (tools/build_variant.sh ... lopq_stream) on the 200 M index
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-hip repl2 repl2u4}; do
  lib=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/libcis_$v.so
  echo "== $v"
  rm -rf /tmp/prof_ab
  CIS_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o r -- python tools/bench_c4x.py 200000000 0 1,2 > /tmp/ab.log 2>&1
  python tools/kstats.py /tmp/prof_ab/r_kernel_stats.csv "adc_stream"
done
