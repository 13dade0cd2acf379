#!/bin/bash
# usage: tools/gpu_sweep.sh "1,4,4 2,4,2 ..."  -> prints value / scan ms per geometry
for g in $1; do
  out=$(CIS_SCAN_GEOM=$g python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  echo "$g $(echo $out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('q/s %.0f ms/step %.3f scan %.3f merge %.3f frac %.3f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['scan_ms'], d['stage_ms_per_step']['merge_ms'], d['roofline']['frac']))")"
done
