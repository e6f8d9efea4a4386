#!/bin/bash
# round 5, session q: the compare-and-swap herd: idle workers wait a random 0-2 us and look again before they touch a head
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 60 python scripts/dev/dataflow_smoke.py 2200 4096 > gpurun_out/r5_q_smoke.log 2>&1 || { tail -5 gpurun_out/r5_q_smoke.log; exit 0; }
timeout -s KILL 120 python scripts/dev/dataflow_trace.py 8192 > gpurun_out/r5_q_trace.log 2>&1
tail -16 gpurun_out/r5_q_trace.log
timeout -s KILL 120 python scripts/dev/dataflow_ab.py 8192 16384 > gpurun_out/r5_q_ab.log 2>&1
tail -5 gpurun_out/r5_q_ab.log
