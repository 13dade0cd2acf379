#!/bin/bash
# usage (GPU box, repo root): tools/r06_round_profile.sh <tag> -- the round's evidence run (every file profiles/README.md lists)
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${tag}_pytest_gpu.txt
timeout 900 python bench.py --detail-file gpurun_out/${tag}_bench_detail.json > gpurun_out/${tag}_bench_default.log 2> gpurun_out/${tag}_bench_default.err
grep '^{' gpurun_out/${tag}_bench_default.log | tail -1 > gpurun_out/${tag}_bench_line.json
for c in c4 c2 c3; do timeout 900 tools/gpu_config_profile.sh ${tag} $c pmc > gpurun_out/${tag}_${c}_profile.txt 2>&1; done
timeout 600 tools/c4x_pmc.sh ${tag} > gpurun_out/${tag}_c4x_pmc.txt 2>&1
timeout 300 tools/profile_c4x.sh ${tag} 200000000 1,2,4,8 > gpurun_out/${tag}_c4x_kernels.txt 2>&1
timeout 300 tools/r06_call_timeline.sh > gpurun_out/${tag}_c4x_call_timeline.txt 2>&1
timeout 300 tools/r06_single_timeline.sh > gpurun_out/${tag}_single_query_c4.txt 2>&1
timeout 200 tools/probes/stream_probe 200000000 8 > gpurun_out/${tag}_stream_probe_kept_forms.txt 2>&1
for c in c4 c2; do tools/batch_gaps.sh $c > gpurun_out/${tag}_batch_timeline_$c.txt 2>&1; done
for c in c4 c2; do tools/overlap_trace.sh $c 3 > gpurun_out/${tag}_overlap_$c.txt 2>&1; done
{ timeout 200 python tools/bench_dlib.py 256; timeout 200 python tools/bench_dlib.py 1024; timeout 200 python tools/bench_cnn.py; } 2>&1 | grep batch > gpurun_out/${tag}_cnn.txt
{ CIS_CNN_PARTS= ; for a in "dlib 256 3" "cnn 256 3"; do timeout 300 python tools/probe_cnn_lanes.py $a; done; } 2>&1 | grep "in flight\|same\|alone" > gpurun_out/${tag}_cnn_lanes.txt
timeout 600 tools/gpu_pmc_cnn.sh ${tag} > /dev/null 2>&1   # writes gpurun_out/${tag}_mfma_utilisation.txt
{ echo "== dlib forward, one chain, per kernel =="; CIS_CNN_PARTS=1 tools/dlib_timeline.sh 256; echo "== DeepSentibank forward, per kernel =="; tools/cnn_timeline.sh 256; echo "== DeepSentibank forward, batch 1024, per kernel =="; tools/cnn_timeline.sh 1024; } > gpurun_out/${tag}_cnn_timelines.txt 2>&1
{ for a in "2048 10000000 8192" "4096 10000000 8192"; do echo "== tools/bench_prodv.py $a =="; timeout 600 python tools/bench_prodv.py $a 2>&1 | grep -v amdgpu | grep "V=\|quota\|parity\|routes\|in flight"; done; } > gpurun_out/${tag}_prodv.txt
timeout 600 python tools/emulate_shard.py 2>&1 | grep "world\|GPU\|->" > gpurun_out/${tag}_shards.txt
timeout 300 python tools/bench_insert.py 2>&1 | grep resident > gpurun_out/${tag}_insert.txt
timeout 300 python tools/bench_limits.py 2>&1 | grep limit > gpurun_out/${tag}_limits.txt
cat gpurun_out/${tag}_pytest_gpu.txt; python tools/bench_summary.py gpurun_out/${tag}_bench_line.json; for c in c4 c2 c3; do head -3 gpurun_out/${tag}_${c}_profile.txt; done
tail -5 gpurun_out/${tag}_c4x_pmc.txt; cat gpurun_out/${tag}_mfma_utilisation.txt gpurun_out/${tag}_cnn.txt gpurun_out/${tag}_prodv.txt gpurun_out/${tag}_shards.txt gpurun_out/${tag}_insert.txt gpurun_out/${tag}_limits.txt
