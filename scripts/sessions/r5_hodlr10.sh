#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 150 python scripts/dev/hodlr_passes_ab.py 10000 > gpurun_out/hodlr_leaf_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/hodlr_leaf_smoke.log
AB_REPS=6 timeout -s KILL 500 python scripts/dev/hodlr_passes_ab.py 700000 2097152 1000000 > gpurun_out/hodlr_passes_ab10.log 2>&1; echo "ab rc=$?"; tail -14 gpurun_out/hodlr_passes_ab10.log
