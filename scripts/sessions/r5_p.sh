#!/bin/bash
# round 5, session p: candidate listing without the debugging atomics (every step under its own short hard time limit)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 40 python scripts/dev/dataflow_smoke.py 300 1100 2200 4096 > gpurun_out/r5_p_smoke.log 2>&1
rc=$?
cat gpurun_out/r5_p_smoke.log | tail -8
if [ $rc -ne 0 ]; then echo "smoke rc=$rc: stopping"; exit 0; fi
timeout -s KILL 300 python -m pytest tests/test_gpu_dataflow.py -q --timeout 120 2>&1 | tail -40 > gpurun_out/r5_p_tests.log
tail -5 gpurun_out/r5_p_tests.log
timeout -s KILL 240 python scripts/dev/dataflow_ab.py 4096 8192 12288 16384 20480 > gpurun_out/r5_p_ab.log 2>&1
tail -8 gpurun_out/r5_p_ab.log
timeout -s KILL 120 python scripts/dev/dataflow_trace.py 8192 16384 > gpurun_out/r5_p_trace.log 2>&1
tail -34 gpurun_out/r5_p_trace.log
