#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profile.sh <tag>
# full evidence run: GPU tests, default bench line, rocprofv3 kernel stats of the same command, PMC passes (FETCH / WRITE / SQ)
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${tag}_pytest_gpu.txt
python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
grep '^{' gpurun_out/${tag}_bench_full.log | tail -1 > gpurun_out/${tag}_bench_line.json
tools/gpu_bench_profile.sh ${tag} --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  tools/gpu_pmc.sh ${tag}_$c "$c" --steps 3 --warmup 1 --no-cpu-baseline --no-cnn > /dev/null 2>&1
done
tools/gpu_pmc.sh ${tag}_sq "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" --steps 3 --warmup 1 --no-cpu-baseline --no-cnn > /dev/null 2>&1
tools/gpu_pmc.sh ${tag}_l2 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" --steps 3 --warmup 1 --no-cpu-baseline --no-cnn > /dev/null 2>&1
# the other routes and shapes: limit sweep (select path), small batches, dlib net
{ python tools/bench_limits.py; for nq in 1 63 512; do echo "NQ=$nq"; NQ=$nq LIMITS=10,100,440,1000 python tools/bench_limits.py; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_limits.txt
rm -rf /tmp/prof_lim; LIMITS=1000,10000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lim -o r -- python tools/bench_limits.py > /dev/null 2>&1
python tools/kstats.py /tmp/prof_lim/r_kernel_stats.csv "adc_all|select|emit_sorted|rocprim|cand_layout" > gpurun_out/${tag}_limits_kernels.txt 2>&1
{ python tools/bench_dlib.py 256; python tools/bench_dlib.py 32; python tools/bench_cnn.py; } 2>&1 | grep batch > gpurun_out/${tag}_cnn.txt
tools/gpu_pmc_cnn.sh ${tag} > /dev/null 2>&1
{ for net in cnn dlib; do echo "== tools/bench_$net.py =="; python tools/mfma_pmc_summary.py gpurun_out/${tag}_${net}_mfma_pmc.csv; done; } > gpurun_out/${tag}_mfma_utilisation.txt 2>&1
cat gpurun_out/${tag}_pytest_gpu.txt; cat gpurun_out/${tag}_mfma_utilisation.txt gpurun_out/${tag}_limits.txt gpurun_out/${tag}_limits_kernels.txt gpurun_out/${tag}_cnn.txt; cat gpurun_out/${tag}_summary.txt | head -40; grep adc_scan gpurun_out/${tag}_*_pmc.csv
