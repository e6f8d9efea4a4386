#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_hodlr.py -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/t_hodlr.log 2>&1; echo "pytest rc=$?"
tail -60 gpurun_out/t_hodlr.log
timeout 600 python bench.py --workload hodlr --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-600
GEORGE_AMD_HODLR_FENCE=1 timeout 600 python bench.py --workload hodlr --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-300
