#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/dense; export TMPDIR=/tmp
timeout -s KILL 1200 python -X faulthandler -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_gemm.py tests/test_gpu_mgpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/dense/tests.log 2>&1; echo "dense tests rc=$?"; tail -4 gpurun_out/dense/tests.log
