#!/bin/bash
# round 4: whole-factorisation A/B of a variant build of the library (libgeorge_amd_c.so) against the tree's
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python scripts/dev/factor_ab_tmp.py "$@" > $O/factor_ab.md 2> $O/factor_ab.err; echo "ab rc=$?"; cat $O/factor_ab.md; tail -3 $O/factor_ab.err
