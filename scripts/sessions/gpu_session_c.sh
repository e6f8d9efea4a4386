#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== default"; timeout 600 python scripts/size_sweep.py 1024 4096 8192 16384 2>&1 | tee gpurun_out/sweep_default.md
echo "== exclusive 32"; GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | tee gpurun_out/sweep_excl32.md
echo "== exclusive 16"; GEORGE_AMD_PANEL_EXCLUSIVE=1 GEORGE_AMD_RESERVE_CUS=16 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | tee gpurun_out/sweep_excl16.md
echo "== exclusive 64"; GEORGE_AMD_PANEL_EXCLUSIVE=1 GEORGE_AMD_RESERVE_CUS=64 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | tee gpurun_out/sweep_excl64.md
