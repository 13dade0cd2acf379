cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=r02n
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${tag}_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
grep '^{' gpurun_out/${tag}_bench_default.log | tail -1 > gpurun_out/${tag}_bench_line.json
{ timeout 200 python tools/bench_dlib.py 256; timeout 200 python tools/bench_dlib.py 32; timeout 200 python tools/bench_cnn.py; } 2>&1 | grep batch > gpurun_out/${tag}_cnn.txt
timeout 600 tools/gpu_pmc_cnn.sh ${tag} > /dev/null 2>&1
{ for net in cnn dlib; do echo "== tools/bench_$net.py =="; python tools/mfma_pmc_summary.py gpurun_out/${tag}_${net}_mfma_pmc.csv; done; } > gpurun_out/${tag}_mfma_utilisation.txt 2>&1
tools/dlib_timeline.sh 256 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_dlib_timeline.txt
cat gpurun_out/${tag}_pytest_gpu.txt gpurun_out/${tag}_cnn.txt gpurun_out/${tag}_mfma_utilisation.txt; head -c 400 gpurun_out/${tag}_bench_line.json
