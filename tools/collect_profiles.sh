#!/bin/bash
# usage (here, repo root): tools/collect_profiles.sh <tag>  -- copies the evidence set tools/r06_round_profile.sh <tag> left under gpurun_out/ into profiles/
tag=$1
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/${tag}_pytest_gpu.txt $g/${tag}_bench_line.json $g/${tag}_bench_detail.json $p/
for c in c4 c2 c3; do
  cp $g/${tag}_${c}_bench_line.json $g/${tag}_${c}_bench_detail.json $g/${tag}_${c}_bench_line_serial.json $g/${tag}_${c}_kernel_stats.csv $g/${tag}_${c}_summary.txt $g/${tag}_${c}_profile.txt $p/
  cp $g/${tag}_${c}_FETCH_SIZE_raw.csv $g/${tag}_${c}_WRITE_SIZE_raw.csv $g/${tag}_${c}_sq_raw.csv $p/
  cp $g/scan_traffic_${c}.json $g/scan_binding_${c}.json $p/
done
for nq in 1 2 4; do cp $g/${tag}_c4x_nq${nq}_FETCH_SIZE_pmc.csv $g/${tag}_c4x_nq${nq}_WRITE_SIZE_pmc.csv $g/${tag}_c4x_nq${nq}_sq_pmc.csv $g/${tag}_c4x_nq${nq}_kernel_stats.csv $p/; done
cp $g/scan_traffic_c4x.json $g/${tag}_c4x_pmc.txt $g/${tag}_c4x_kernels.txt $g/${tag}_c4x_kernel_stats.csv $g/${tag}_c4x_call_timeline.txt $g/${tag}_single_query_c4.txt $g/${tag}_stream_probe_kept_forms.txt $p/
python tools/pmc_stamp.py $p/scan_traffic_c4x.json $p/scan_traffic_c4.json $p/scan_traffic_c2.json $p/scan_traffic_c3.json $p/scan_binding_c4.json $p/scan_binding_c2.json $p/scan_binding_c3.json
cp $g/${tag}_batch_timeline_c4.txt $g/${tag}_batch_timeline_c2.txt $g/${tag}_overlap_c4.txt $g/${tag}_overlap_c2.txt $p/
cp $g/${tag}_cnn_lanes.txt $p/ 2>/dev/null
cp $g/${tag}_cnn.txt $g/${tag}_mfma_utilisation.txt $g/${tag}_cnn_mfma_pmc.csv $g/${tag}_dlib_mfma_pmc.csv $g/${tag}_cnn_timelines.txt $p/
cp $g/${tag}_prodv.txt $g/${tag}_shards.txt $g/${tag}_insert.txt $g/${tag}_limits.txt $p/
ls $p | grep -c "^${tag}_"
