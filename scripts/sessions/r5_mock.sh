#!/bin/bash
# session: stand-in RCCL (selftest, grid matrix in both communicator modes), then test_gpu_mgpu as usual
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 120 python scripts/dev/dataflow_smoke.py > gpurun_out/mock_smoke.log 2>&1; echo "smoke rc=$?"
timeout -s KILL 200 python tests/mock_rccl/selftest.py > gpurun_out/mock_selftest.log 2>&1; echo "selftest rc=$?"; tail -3 gpurun_out/mock_selftest.log
timeout -s KILL 900 python -X faulthandler -m pytest tests/test_gpu_mgpu_mock_rccl.py -x -q -m gpu -p no:cacheprovider > gpurun_out/mock_matrix.log 2>&1; echo "matrix rc=$?"; tail -30 gpurun_out/mock_matrix.log
timeout -s KILL 400 python -X faulthandler -m pytest tests/test_gpu_mgpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/mgpu_plain.log 2>&1; echo "plain rc=$?"; tail -5 gpurun_out/mgpu_plain.log
