#!/bin/bash
# Round-end artefacts: the default bench line, the size sweep, the potf2 phase table, the queue pattern, then the profiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
T0=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
timeout 600 python scripts/size_sweep.py > $O/size_sweep.md 2>/dev/null; cat $O/size_sweep.md
./scripts/dev/potf2_phases > $O/potf2_phases_v2.txt 2>&1
timeout 300 python scripts/dev/queue_pattern.py 2>/dev/null > $O/queue_pattern.txt
GEORGE_AMD_PRIVATE_STREAMS=1 timeout 300 python scripts/dev/queue_pattern.py 2>/dev/null > $O/queue_pattern_private_streams.txt
bash scripts/profile_r02.sh > $O/profile.log 2>&1; tail -2 $O/profile.log
