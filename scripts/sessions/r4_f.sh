#!/bin/bash
# round 4, session f: the column-priority schedule -- bit-identity against the panel look-ahead, then the A/B sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "trapezoid or layouts or lower" --timeout 600 > $O/gemm.log 2>&1; echo "gemm rc=$?"; tail -3 $O/gemm.log
timeout 1200 python -m pytest tests/test_gpu_solver.py -x -q -k "every_switch" --timeout 900 > $O/switch.log 2>&1; echo "switch rc=$?"; tail -5 $O/switch.log
timeout 1500 python scripts/schedule_ab.py > $O/schedule_ab.md 2> $O/schedule_ab.err; echo "ab rc=$?"; cat $O/schedule_ab.md
