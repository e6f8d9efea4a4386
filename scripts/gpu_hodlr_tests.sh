#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/hodlr; export TMPDIR=/tmp
timeout -s KILL 900 python -X faulthandler -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider > gpurun_out/hodlr/tests_all.log 2>&1; echo "hodlr tests rc=$?"; tail -4 gpurun_out/hodlr/tests_all.log
