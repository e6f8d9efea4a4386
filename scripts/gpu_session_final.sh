#!/bin/bash
# Round-end validation on a gpurun box: full -m gpu suite, smoke(), the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
T0=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
cut -c1-1500 $O/bench_default.json
