#!/bin/bash
# usage (GPU box, repo root): tools/r05_stream_seg.sh -- chunk length of the streaming route on the 200 M index (scan mode 6 + CIS_STREAM_SEG)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for seg in ${SEGS:-65536 52224 32768 16384 8192}; do
  for st in 0 1; do
    echo "== CIS_STREAM_SEG=$seg static=$st"
    rm -rf /tmp/prof_ab
    if [ $st = 1 ]; then export CIS_STREAM_STATIC=1; else unset CIS_STREAM_STATIC; fi
    CIS_STREAM_SEG=$seg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o r -- python tools/bench_c4x.py 200000000 6 1,2 > /tmp/ab.log 2>&1
    python tools/kstats.py /tmp/prof_ab/r_kernel_stats.csv "adc_stream" | grep -v "true>"
    grep "exhaustive" /tmp/ab.log | head -3 | cut -c1-200
  done
done
