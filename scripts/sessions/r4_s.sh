#!/bin/bash
# round 4, session s: the whole -m gpu suite + smoke once more on the last tree (staircase table of 128 groups)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --no-cpu --no-extra --steps 5 --warmup 2 2>/dev/null | grep '^{' | cut -c1-400
