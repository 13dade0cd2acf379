cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== c2 A/B"; for m in 2 1; do echo "CIS_SCAN3_TWOPASS=$m"; CIS_SCAN3_TWOPASS=$m tools/gpu_ab.sh "libcis_hip.so" 2 --config c2; done
echo "== fallbacks"; CIS_SCAN4_DEBUG=1 python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-pcie 2>&1 | grep "k_adc_scan4\|^{" | cut -c1-300 | tail -6
echo "== counters"; CIS_LIB_PATH=$PWD/columbiaimagesearch_amd/lib/libcis_s3ctr.so python tools/debug_counters4.py c2 2>&1 | grep -v amdgpu
echo "== parity, scan5 route"; timeout 900 python -m pytest tests/test_lopq_hip_parity.py -x -q -k "scan5 or fuzz or crowd or dupl" 2>&1 | tail -3
} > gpurun_out/r03e_scan4.txt 2>&1
cat gpurun_out/r03e_scan4.txt
