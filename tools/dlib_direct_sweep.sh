#!/bin/bash
# tile sweep of the direct 3x3 kernel inside the dlib forward (batch 256): per-config forward time and mean kernel times
for c in 0 1 2; do
  echo "CIS_CNN_DIRECT_CFG=$c"
  CIS_CNN_DIRECT_CFG=$c tools/dlib_timeline.sh 256 2>&1 | grep "direct\|batch" | python3 -c '
import sys,re,collections
d=collections.defaultdict(list)
for l in sys.stdin:
    if l.startswith("batch"): print(l.strip()); continue
    m=re.match(r"(.*?)\s+([\d.]+) us", l)
    if m: d[m.group(1).strip()].append(float(m.group(2)))
for k,v in d.items(): print("  %-44s n=%d mean %.1f us" % (k,len(v),sum(v)/len(v)))
'
done
