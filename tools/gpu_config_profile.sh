#!/bin/bash
# usage (GPU box, repo root): tools/gpu_config_profile.sh <tag> <config> [pmc]
# bench line of one BASELINE config + rocprofv3 kernel stats of the same command (+ FETCH/WRITE PMC passes -> scan traffic)
tag=$1; cfg=$2; pmc=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --config $cfg --no-cnn > gpurun_out/${tag}_${cfg}_bench_full.log 2>&1
grep '^{' gpurun_out/${tag}_${cfg}_bench_full.log | tail -1 > gpurun_out/${tag}_${cfg}_bench_line.json
tools/gpu_bench_profile.sh ${tag}_${cfg} --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-cnn --no-pcie > gpurun_out/${tag}_${cfg}_summary.txt 2>&1
if [ -n "$pmc" ]; then
  export CIS_BENCH_PIPELINE=1   # the counter passes run one batch at a time: per-kernel counters of isolated launches (7.125 full-launch equivalents)
  for c in FETCH_SIZE WRITE_SIZE; do
    tools/gpu_pmc.sh ${tag}_${cfg}_$c "$c" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie > /dev/null 2>&1
  done
  algo=$(python -c "import json;print(json.load(open('gpurun_out/${tag}_${cfg}_bench_line.json'))['roofline']['algorithmic_bytes_per_launch'])")
  python tools/scan_traffic.py ${tag} ${cfg} 7.125 $algo > /dev/null
  tools/gpu_pmc.sh ${tag}_${cfg}_sq "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie > /dev/null 2>&1
  CIS_BENCH_PIPELINE=1 python bench.py --config $cfg --no-cnn --no-cpu-baseline --no-pcie --steps 10 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${tag}_${cfg}_bench_line_serial.json
  python tools/scan_binding.py ${tag} ${cfg} 7.125 > /dev/null
  unset CIS_BENCH_PIPELINE
fi
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_${cfg}_bench_line.json'))
print('${cfg}: value %.0f q/s  ms/step %.3f  roofline frac %.3f  launch %.3f ms  recall %.3f  cand/q %.0f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['recall_at_10'], d['config']['candidates_per_query']))
print(' stages', {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}, 'encode %.1f M/s' % (d['encode']['value'] / 1e6), 'pcie', d['pcie_inclusive'] and round(d['pcie_inclusive']['value']))
print(' cpu', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in (d['cpu_baseline'] or {}).items() if not k.startswith('sample')}, 'parity', d['parity'])
PY
head -30 gpurun_out/${tag}_${cfg}_summary.txt
