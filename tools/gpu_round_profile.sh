#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profile.sh <tag>
# full evidence run of a round: GPU tests, the bench line of every BASELINE config (default = c4 with the CNN legs and CPU baselines),
# rocprofv3 kernel stats of the same commands, PMC passes (FETCH / WRITE -> scan traffic, SQ), limit sweep, CNN, production V, shards
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${tag}_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
grep '^{' gpurun_out/${tag}_bench_default.log | tail -1 > gpurun_out/${tag}_bench_line.json
for c in c4 c2 c3; do timeout 900 tools/gpu_config_profile.sh ${tag} $c pmc > gpurun_out/${tag}_${c}_profile.txt 2>&1; done
{ timeout 300 python tools/bench_limits.py; for nq in 1 63 512; do echo "NQ=$nq"; NQ=$nq LIMITS=10,100,440,1000 timeout 300 python tools/bench_limits.py; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_limits.txt
{ timeout 200 python tools/bench_dlib.py 256; timeout 200 python tools/bench_dlib.py 32; timeout 200 python tools/bench_cnn.py; } 2>&1 | grep batch > gpurun_out/${tag}_cnn.txt
timeout 600 tools/gpu_pmc_cnn.sh ${tag} > /dev/null 2>&1
{ for net in cnn dlib; do echo "== tools/bench_$net.py =="; python tools/mfma_pmc_summary.py gpurun_out/${tag}_${net}_mfma_pmc.csv; done; } > gpurun_out/${tag}_mfma_utilisation.txt 2>&1
{ for a in "2048 10000000 1024" "2048 10000000 8192" "4096 10000000 8192"; do echo "== tools/bench_prodv.py $a =="; timeout 600 python tools/bench_prodv.py $a 2>&1 | grep -v amdgpu | grep "V=\|quota\|parity\|routes"; done; } > gpurun_out/${tag}_prodv.txt
timeout 400 python tools/emulate_shard.py 2>&1 | grep world > gpurun_out/${tag}_shards.txt
{ timeout 300 python tools/bench_ingest.py 2>&1 | grep -v amdgpu | tail -3; timeout 300 python tools/bench_ingest_buffers.py 8192 2>&1 | grep "stages\|buffers"; } > gpurun_out/${tag}_ingest.txt
cat gpurun_out/${tag}_pytest_gpu.txt; for c in c4 c2 c3; do head -4 gpurun_out/${tag}_${c}_profile.txt; done; cat gpurun_out/${tag}_mfma_utilisation.txt gpurun_out/${tag}_cnn.txt gpurun_out/${tag}_prodv.txt gpurun_out/${tag}_shards.txt gpurun_out/${tag}_ingest.txt; tail -20 gpurun_out/${tag}_limits.txt
