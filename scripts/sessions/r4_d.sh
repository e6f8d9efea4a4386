#!/bin/bash
# round 4, session d: HODLR sweep with update fused into the next level's reduce (A/B + parity), the fp64 ceiling in inline assembly
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -s -k "ceiling" > $O/ceiling.log 2>&1; echo "ceiling rc=$?"; grep -a "ceiling\]\|passed\|failed" $O/ceiling.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py -x -q --timeout 900 > $O/hodlr.log 2>&1; echo "hodlr rc=$?"; tail -8 $O/hodlr.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c4" --timeout 900 > $O/c4.log 2>&1; echo "c4 rc=$?"; tail -3 $O/c4.log
for arm in fused twopass; do
  if [ $arm = twopass ]; then export GEORGE_AMD_HODLR_NO_FUSED_SWEEP=1; else unset GEORGE_AMD_HODLR_NO_FUSED_SWEEP; fi
  for rep in 1 2; do timeout 300 python bench.py --workload hodlr --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm', d['ms_per_step'], d['log_likelihood'])"; done
done
unset GEORGE_AMD_HODLR_NO_FUSED_SWEEP
