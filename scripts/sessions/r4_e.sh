#!/bin/bash
# round 4, session e: the whole -m gpu suite after the pruning, smoke, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cat $O/bench_line.json | cut -c1-1500
