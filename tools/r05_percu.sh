#!/bin/bash
# usage (GPU box, repo root): tools/r05_percu.sh -- k_adc_scan4 with fewer resident workgroups per CU (room for the other batches' kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in c4 c2; do
  for v in 0 3 2 5 4; do
    [ $cfg = c4 ] && [ $v -gt 3 ] && continue
    echo "== $cfg CIS_S4_PER_CU=$v"
    CIS_S4_PER_CU=$v python bench.py --config $cfg --no-cnn --no-cpu-baseline --no-pcie --no-c4x 2>/dev/null | grep '^{' | tail -1 > /tmp/l.json
    python tools/bench_summary.py /tmp/l.json
  done
done
