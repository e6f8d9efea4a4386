#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr16k -o trace -- python $R/bench.py --n 16384 --steps 3 --warmup 1 --no-cpu --no-extra > $O/tr16k.log 2>&1
cd $R
f=$(find $O/tr16k -name "*.db" | head -1); python scripts/chain_stats.py "$f" | head -12
GEORGE_AMD_POTF2=v1 python bench.py --n 16384 --steps 5 --warmup 2 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v1 N', d['config']['N'], 'ms', d['ms_per_step'])"
python bench.py --workload hodlr --steps 10 --warmup 3 --no-cpu 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 ms', d['ms_per_step'])"
find $O -name "*.db" -size +6M -delete
