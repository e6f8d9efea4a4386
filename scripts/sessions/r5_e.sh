#!/bin/bash
# round 5, session e: wave-parallel windowed claims, five queues
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dataflow.py -q --timeout 180 2>&1 | tail -60 > gpurun_out/r5_e_tests.log
timeout 900 python scripts/dev/dataflow_ab.py 4096 8192 12288 16384 20480 > gpurun_out/r5_e_ab.log 2>&1
timeout 600 python scripts/dev/dataflow_trace.py 8192 16384 > gpurun_out/r5_e_trace.log 2>&1
tail -5 gpurun_out/r5_e_tests.log; tail -12 gpurun_out/r5_e_ab.log; tail -40 gpurun_out/r5_e_trace.log
