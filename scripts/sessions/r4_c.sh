#!/bin/bash
# round 4, session c: per-row trailing updates in both multi-GPU forms; scale-model traces again; the fp64 ceiling microbenchmark
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -s -k "ceiling or microbench" > $O/ceiling.log 2>&1; echo "ceiling rc=$?"; grep -a "ceiling\]\|suite\]\|passed\|failed" $O/ceiling.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_mgpu.py -x -q --timeout 400 > $O/mgpu.log 2>&1; echo "mgpu rc=$?"; tail -3 $O/mgpu.log
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q --timeout 400 > $O/dist.log 2>&1; echo "dist rc=$?"; tail -3 $O/dist.log
timeout 1500 python scripts/scale_model.py collect $O/scale_traces.json > $O/scale_collect.log 2>&1; echo "collect rc=$?"; cut -c1-330 $O/scale_collect.log | tail -8
