#!/bin/bash
# round 4: A/B of a variant build of the library (libgeorge_amd_c.so) against the tree's, one script per argument
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
S=${1:-scripts/dev/factor_ab.py}; shift
timeout 1200 python $S "$@" > $O/ab.md 2> $O/ab.err; echo "ab rc=$?"; cat $O/ab.md; tail -3 $O/ab.err
