#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_full_size_properties.py -q -m gpu -x -k "rccl or two_ranks or four_ranks or bench_multi or shards_merge" 2>&1 | tail -3 | tee gpurun_out/r04n_pytest_dist.txt
echo "== bench --gpus 1 through the sharded path (world-1 RCCL group)"; CIS_BENCH_FORCE_DIST=1 python bench.py --steps 20 --no-cnn --no-cpu-baseline --no-pcie 2>/dev/null | python tools/bench_summary.py
echo "== plain"; python bench.py --steps 20 --no-cnn --no-cpu-baseline --no-pcie --config c4 2>/dev/null | python tools/bench_summary.py
