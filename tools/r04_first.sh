#!/bin/bash
# round 4, first GPU call: changed tests, the restructured default bench line, CNN per-kernel timelines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cnn_hip_parity.py tests/test_reference_surfaces.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r04a_pytest_subset.txt
cat gpurun_out/r04a_pytest_subset.txt
timeout 900 python bench.py > gpurun_out/r04a_bench_default.log 2>&1
grep '^{' gpurun_out/r04a_bench_default.log | tail -1 > gpurun_out/r04a_bench_line.json
tail -c 600 gpurun_out/r04a_bench_default.log
python - <<PY
import json
d=json.load(open('gpurun_out/r04a_bench_line.json'))
print('c4', round(d['value']), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['stage_ms_per_step'])
for k,v in (d.get('configs') or {}).items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,list))}, v.get('stage_ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('parity_green'))
print('cnn', d['cnn']['value'], d['cnn']['roofline']['frac'], 'dlib', d['dlib']['value'], d['dlib']['roofline']['frac'], d['dlib']['batch_1024'])
print('ingest', {a:b for a,b in d['ingest'].items() if not isinstance(b,(dict,str))})
PY
tools/dlib_timeline.sh 256 > gpurun_out/r04a_dlib_timeline.txt 2>&1
tools/cnn_timeline.sh > gpurun_out/r04a_cnn_timeline.txt 2>&1
tail -70 gpurun_out/r04a_dlib_timeline.txt; tail -30 gpurun_out/r04a_cnn_timeline.txt
