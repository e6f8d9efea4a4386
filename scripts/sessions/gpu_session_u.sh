#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for T in 64 32 6432; do
  export GEORGE_AMD_K128_TILE=$T
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr$T -o trace -- python $R/bench.py --n 16384 --steps 3 --warmup 1 --no-cpu --no-extra > $O/tr$T.log 2>&1
  cd $R
  echo "tile $T"; grep '^{' $O/tr$T.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms under rocprof', d['ms_per_step'])"
  f=$(find $O/tr$T -name "*.db" | head -1); python scripts/chain_stats.py "$f" | head -7
  python bench.py --n 16384 --steps 5 --warmup 2 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  N', d['config']['N'], 'ms', d['ms_per_step'])"
done
find $O -name "*.db" -delete
