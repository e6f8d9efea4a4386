#!/bin/bash
# round 5, session aa: old passes of the next diagonal blocks listed; soon rows first in the bulk
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 60 python scripts/dev/dataflow_smoke.py 300 2200 4096 > gpurun_out/r5_aa_smoke.log 2>&1 || { tail -5 gpurun_out/r5_aa_smoke.log; exit 0; }
grep -v mode gpurun_out/r5_aa_smoke.log | tail -3
timeout -s KILL 200 python -m pytest tests/test_gpu_dataflow.py -q --timeout 120 2>&1 | tail -5 > gpurun_out/r5_aa_tests.log; tail -3 gpurun_out/r5_aa_tests.log
timeout -s KILL 200 python scripts/dev/dataflow_ab.py 4096 8192 12288 16384 20480 > gpurun_out/r5_aa_ab.log 2>&1
tail -8 gpurun_out/r5_aa_ab.log
timeout -s KILL 120 python scripts/dev/dataflow_trace.py 8192 16384 > gpurun_out/r5_aa_trace.log 2>&1
tail -32 gpurun_out/r5_aa_trace.log
