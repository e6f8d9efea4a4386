#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_chain
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== exclusive 32 + inner split"; GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^|"
echo "== exclusive 16 + inner split"; GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 GEORGE_AMD_RESERVE_CUS=16 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^|"
cd /tmp && export TMPDIR=/tmp
N=16384
GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_chain/x$N" -o trace -- python $R/bench.py --n $N --steps 2 --warmup 1 --no-cpu --no-extra > "$R/gpurun_out/prof_chain/x$N.log" 2>&1
f=$(find "$R/gpurun_out/prof_chain/x$N" -name "*.db" | head -1)
echo "== N=$N exclusive+split"; python $R/scripts/chain_stats.py "$f"
find "$R/gpurun_out/prof_chain" -name "*.db" -size +8M -delete
