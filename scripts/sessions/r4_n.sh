#!/bin/bash
# round 4: the HODLR gpu tests (+ the reference suite and the split-tree tests) after the ACA kernel's LDS mirrors
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_gpu_hodlr.py tests/test_gpu_reference_suite.py tests/test_gpu_hodlr_split.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-200
python bench.py --workload hodlr --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' | cut -c1-600
