#!/bin/bash
# usage (GPU box, repo root): tools/gpu_config_profile.sh <tag> <config> [pmc]
# bench line of one BASELINE config + rocprofv3 kernel stats of the same command (+ FETCH / WRITE / SQ PMC passes -> scan traffic, binding)
tag=$1; cfg=$2; pmc=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --config $cfg --no-cnn --detail-file gpurun_out/${tag}_${cfg}_bench_detail.json > gpurun_out/${tag}_${cfg}_bench_full.log 2>&1
grep '^{' gpurun_out/${tag}_${cfg}_bench_full.log | tail -1 > gpurun_out/${tag}_${cfg}_bench_line.json
CIS_BENCH_MIN_REPS=2 CIS_BENCH_MIN_TIMED_S=0 tools/gpu_bench_profile.sh ${tag}_${cfg} --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-cnn --no-pcie > gpurun_out/${tag}_${cfg}_summary.txt 2>&1
if [ -n "$pmc" ]; then
  export CIS_BENCH_PIPELINE=1 CIS_BENCH_MIN_REPS=2 CIS_BENCH_MIN_TIMED_S=0   # the counter passes run one batch at a time: counters of isolated launches
  python bench.py --config $cfg --no-cnn --no-cpu-baseline --no-pcie --steps 10 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${tag}_${cfg}_bench_line_serial.json
  for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"; do
    name=$c; [ "${c:0:3}" = "SQ_" ] && name=sq
    rm -rf /tmp/pmc_cfg
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_cfg -o r -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie > /dev/null 2>&1
    head -1 /tmp/pmc_cfg/r_counter_collection.csv > gpurun_out/${tag}_${cfg}_${name}_raw.csv
    grep "k_adc_s" /tmp/pmc_cfg/r_counter_collection.csv >> gpurun_out/${tag}_${cfg}_${name}_raw.csv
  done
  python tools/pmc_scan.py ${tag} ${cfg} gpurun_out/${tag}_${cfg}_bench_line_serial.json
  unset CIS_BENCH_PIPELINE CIS_BENCH_MIN_REPS CIS_BENCH_MIN_TIMED_S
fi
python tools/bench_summary.py gpurun_out/${tag}_${cfg}_bench_line.json
head -30 gpurun_out/${tag}_${cfg}_summary.txt
