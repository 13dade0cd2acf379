#!/bin/bash
# usage (GPU box): tools/r03_encode_ab.sh lib1.so lib2.so ... -- encode rate of the 10M build (c4) and the 1M x 4096-d build (c3) per library build
cd $GRAFT_REPO_ROOT
for l in "$@"; do for c in c4 c3; do
  echo -n "$l $c: "; CIS_LIB_PATH=$GRAFT_REPO_ROOT/columbiaimagesearch_amd/lib/$l timeout 300 python bench.py --config $c --steps 3 --no-cnn --no-pcie --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith(chr(123))][-1]); print('encode %.1f M vectors/s, insert %.1f ms, q/s %.0f' % (l['encode']['value']/1e6, l['build']['insert_s']*1e3, l['value']))"
done; done
