#!/bin/bash
# round 4, session g: paired LDS fragment reads in the k-major GEMM kernels, A/B in one process
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python scripts/dev/gemm_pair_ab.py > $O/gemm_pair_ab.md 2> $O/gemm_pair_ab.err; echo "ab rc=$?"; cat $O/gemm_pair_ab.md; tail -3 $O/gemm_pair_ab.err
