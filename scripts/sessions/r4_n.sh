#!/bin/bash
# round 4: the GEMM / multi-GPU gpu tests after the grouped tile order
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_mgpu.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
