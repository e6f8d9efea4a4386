#!/bin/bash
# round 4: A/B of one variant of the main GEMM kernel against the tree's (column "b128" = variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python scripts/dev/gemm_ab_tmp.py > $O/gemm_ab.md 2> $O/gemm_ab.err; echo "ab rc=$?"; cat $O/gemm_ab.md; tail -3 $O/gemm_ab.err
