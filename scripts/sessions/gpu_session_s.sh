#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 60 ./scripts/dev/potf2_phases > $O/potf2_phases_v2.txt 2>&1; cat $O/potf2_phases_v2.txt
timeout 60 ./scripts/dev/potf2_phases_v1 2>&1 | grep -E "max|total"
for n in 4096 8192 16384; do
  timeout 300 python bench.py --n $n --steps 5 --warmup 2 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N', d['config']['N'], 'ms', d['ms_per_step'], 'TF', d['value'], 'parity', d.get('parity'))"
done
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_hodlr.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5
