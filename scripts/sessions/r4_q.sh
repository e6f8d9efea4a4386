#!/bin/bash
# round 4, session q: staircase GEMM -- its test, the multi-GPU forms that use it, and the scale-model traces again
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_mgpu.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
timeout 1500 python scripts/scale_model.py collect $O/scale_traces.json > $O/scale_collect.log 2>&1; echo "collect rc=$?"; cut -c1-200 $O/scale_collect.log | tail -8
gzip -9 -k $O/scale_traces.json
