#!/bin/bash
# usage: tools/gpu_ab.sh "libA.so libB.so ..." [rounds] [bench args]   -- interleaved A/B of several builds in ONE box
LIBS=$1; R=${2:-2}; shift; shift
for i in $(seq $R); do
  for L in $LIBS; do
    out=$(CIS_LIB_PATH=$PWD/columbiaimagesearch_amd/lib/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cnn --no-pcie "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "$L $(echo $out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('q/s %.0f step %.3f scan_kernel %.3f merge %.3f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['stage_ms_per_step']['merge_ms'], d['roofline']['frac']))")"
  done
done
