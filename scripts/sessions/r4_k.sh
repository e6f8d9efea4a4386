#!/bin/bash
# round 4, session k: resident workgroups walking the tile list (persistent form of the main GEMM kernel), A/B in one process
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python scripts/dev/gemm_persist_ab.py > $O/gemm_persist_ab.md 2> $O/gemm_persist_ab.err; echo "ab rc=$?"; cat $O/gemm_persist_ab.md; tail -3 $O/gemm_persist_ab.err
