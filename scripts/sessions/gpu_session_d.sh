#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_chain
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for N in 8192 16384; do
  timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_chain/t$N" -o trace -- python $R/bench.py --n $N --steps 2 --warmup 1 --no-cpu --no-extra > "$R/gpurun_out/prof_chain/t$N.log" 2>&1
  f=$(find "$R/gpurun_out/prof_chain/t$N" -name "*.db" | head -1)
  echo "== N=$N"; python $R/scripts/chain_stats.py "$f"
  python $R/scripts/summarize_prof.py "$f" "$R/gpurun_out/prof_chain/t$N.md" 8
done
find "$R/gpurun_out/prof_chain" -name "*.db" -size +8M -delete
