#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_hod
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_hod/t" -o hod -- python $R/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu > "$R/gpurun_out/prof_hod/hod.log" 2>&1
f=$(find "$R/gpurun_out/prof_hod/t" -name "*.db" | head -1)
python $R/scripts/hodlr_levels.py "$f"
python $R/scripts/summarize_prof.py "$f" "$R/gpurun_out/prof_hod/hod.md"
