#!/bin/bash
# One gpurun call: every step under its own `timeout -s KILL` (shorter than the call's limit: a hang must cost its own
# step, not the call), everything to gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 2000 -- 'bash scripts/gpu_session.sh'
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 1200 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/suite_gpu.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/suite_gpu.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/suite_entry_smoke.log 2>&1; echo "entry smoke rc=$?"; tail -2 gpurun_out/suite_entry_smoke.log
timeout -s KILL 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-3000
