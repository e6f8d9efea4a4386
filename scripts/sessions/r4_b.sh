#!/bin/bash
# round 4, session b: the re-written ABI multi-GPU solver on virtual devices, the torch driver's new grids, scale-model traces
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python -m pytest tests/test_gpu_mgpu.py -x -q --timeout 400 > $O/mgpu.log 2>&1; echo "mgpu rc=$?"; tail -15 $O/mgpu.log
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q --timeout 400 > $O/dist.log 2>&1; echo "dist rc=$?"; tail -5 $O/dist.log
timeout 1500 python scripts/scale_model.py collect $O/scale_traces.json > $O/scale_collect.log 2>&1; echo "collect rc=$?"; cut -c1-400 $O/scale_collect.log | tail -12
