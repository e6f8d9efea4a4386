#!/bin/bash
# round 4, session j: panel width at the headline size with the round-4 GEMM kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
for nb in 1024 1536 2048 1024; do
  timeout 600 python bench.py --no-cpu --no-extra --steps 3 --warmup 1 --nb $nb --detail "" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nb=$nb', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['with_overlapped_block_column_launches']['achieved'], d['phases_ms'])"
done
