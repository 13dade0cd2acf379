#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_full_size_properties.py -q -m gpu -x -k "rccl or two_ranks or four_ranks or bench_multi or shards_merge" 2>&1 | tail -5 | tee gpurun_out/r04m_pytest_dist.txt
timeout 600 python tools/emulate_shard.py 2>&1 | grep "world\|GPU" | tee gpurun_out/r04m_shards.txt
