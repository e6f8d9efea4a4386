#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
./scripts/dev/potf2_phases > $O/potf2_phases.txt 2>&1; cat $O/potf2_phases.txt
T0=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
cut -c1-1800 $O/bench_default.json
