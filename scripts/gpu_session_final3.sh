#!/bin/bash
# last refresh after HODLR-only changes: full -m gpu suite, smoke, default bench line, C4 trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; P=$R/gpurun_out/prof_r02; mkdir -p $O $P
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $P/traceC4
timeout 600 rocprofv3 --kernel-trace --stats -d $P/traceC4 -o trace -- python $R/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu > $P/traceC4.log 2>&1
cd $R
f=$(find $P/traceC4 -name "*.db" | head -1); python scripts/summarize_prof.py "$f" $P/traceC4.md; python scripts/hodlr_levels.py "$f" > $P/hodlr_levels_C4.txt; head -30 $P/hodlr_levels_C4.txt
