#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 150 python scripts/dev/hodlr_passes_ab.py 10000 > gpurun_out/hodlr_leaf_smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/hodlr_leaf_smoke.log
timeout -s KILL 600 python -X faulthandler -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py tests/test_gpu_reference_suite.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider > gpurun_out/hodlr_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/hodlr_tests.log
timeout -s KILL 400 python scripts/dev/hodlr_passes_ab.py 50000 262144 700000 > gpurun_out/hodlr_passes_ab9.log 2>&1; echo "ab rc=$?"; tail -14 gpurun_out/hodlr_passes_ab9.log
GEORGE_AMD_HODLR_LEAF_MM=1 timeout -s KILL 200 python scripts/dev/hodlr_passes_ab.py 262144 > gpurun_out/hodlr_passes_ab9_mm.log 2>&1; echo "ab(mm) rc=$?"; tail -5 gpurun_out/hodlr_passes_ab9_mm.log
