#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 150 python scripts/dev/hodlr_passes_ab.py 4096 > gpurun_out/hodlr_passes_smoke.log 2>&1; echo "smoke rc=$?"; tail -6 gpurun_out/hodlr_passes_smoke.log
timeout -s KILL 400 python -X faulthandler -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py -x -q -m gpu -p no:cacheprovider > gpurun_out/hodlr_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/hodlr_tests.log
timeout -s KILL 400 python scripts/dev/hodlr_passes_ab.py > gpurun_out/hodlr_passes_ab.log 2>&1; echo "ab rc=$?"; tail -20 gpurun_out/hodlr_passes_ab.log
