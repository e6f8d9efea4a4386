#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== default (depth 1)"; timeout 900 python scripts/size_sweep.py 1024 2048 4096 8192 12288 16384 24576 32768 49152 65536 2>&1 | grep "^| [0-9]" | tee gpurun_out/sweep_d1.md
echo "== depth 0"; GEORGE_AMD_LOOKAHEAD_DEPTH=0 timeout 900 python scripts/size_sweep.py 1024 49152 65536 2>&1 | grep "^| [0-9]"
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -m gpu -q -p no:cacheprovider -x --timeout 900 2>&1 | tail -3
