#!/bin/bash
# usage: tools/gpu_ab.sh libA.so libB.so [rounds] ["geom"]   -- interleaved A/B of two builds in ONE box
A=$1; B=$2; R=${3:-3}; G=${4:-2,4,4}
for i in $(seq $R); do
  for L in $A $B; do
    out=$(CIS_LIB_PATH=$PWD/columbiaimagesearch_amd/lib/$L CIS_SCAN_GEOM=$G python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cnn 2>/dev/null | grep '^{' | tail -1)
    echo "$L $(echo $out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('q/s %.0f scan %.3f frac %.3f' % (d['value'], d['stage_ms_per_step']['scan_ms'], d['roofline']['frac']))")"
  done
done
