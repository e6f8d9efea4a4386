#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/kern; export TMPDIR=/tmp
timeout -s KILL 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference_kernels.py -x -q -m gpu -p no:cacheprovider > gpurun_out/kern/tests.log 2>&1; echo "kernel tests rc=$?"; tail -15 gpurun_out/kern/tests.log | cut -c1-220
timeout -s KILL 300 python scripts/dev/f1_time.py nocpu 2>&1 | tail -4
