#!/bin/bash
# round 4, session h: DMA issued right after the fragment reads, ahead of all 64 MFMAs of the slab (column b128 = new) ("b128" column = new, "b64" column = before)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python scripts/dev/gemm_ab_tmp.py > $O/gemm_ab.md 2> $O/gemm_ab.err; echo "ab rc=$?"; cat $O/gemm_ab.md; tail -3 $O/gemm_ab.err
