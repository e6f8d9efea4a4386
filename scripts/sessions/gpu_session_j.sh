#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_hod
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_hodlr.py -m gpu -q -p no:cacheprovider --timeout 900 -x > gpurun_out/t_hodlr.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/t_hodlr.log | cut -c1-300
timeout 600 python bench.py --workload hodlr --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-140
GEORGE_AMD_HODLR_SERIAL_LEVELS=1 timeout 600 python bench.py --workload hodlr --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-140
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_hod/t" -o hod -- python $R/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu > "$R/gpurun_out/prof_hod/hod.log" 2>&1
f=$(find "$R/gpurun_out/prof_hod/t" -name "*.db" | head -1)
python $R/scripts/hodlr_levels.py "$f"
