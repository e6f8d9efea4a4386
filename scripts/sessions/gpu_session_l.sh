#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_chain
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for D in 1 3; do
GEORGE_AMD_LOOKAHEAD_DEPTH=$D timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_chain/d$D" -o trace -- python $R/bench.py --n 16384 --steps 2 --warmup 1 --no-cpu --no-extra > "$R/gpurun_out/prof_chain/d$D.log" 2>&1
f=$(find "$R/gpurun_out/prof_chain/d$D" -name "*.db" | head -1)
echo "== depth $D"; python $R/scripts/stream_busy.py "$f"; python $R/scripts/chain_stats.py "$f" | head -1
done
