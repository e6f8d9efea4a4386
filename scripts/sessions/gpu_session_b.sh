#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x --timeout 900 -k "not c3 and not 65536 and not NS and not C3" > gpurun_out/t_solver.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/t_solver.log
echo "== split (new)"; timeout 600 python scripts/size_sweep.py 1024 2048 4096 8192 12288 16384 24576 32768 2>&1 | tee gpurun_out/sweep_split.md
echo "== old arm"; GEORGE_AMD_NO_PANEL_INNER_SPLIT=1 timeout 600 python scripts/size_sweep.py 1024 4096 8192 16384 32768 2>&1 | tee gpurun_out/sweep_nosplit.md
