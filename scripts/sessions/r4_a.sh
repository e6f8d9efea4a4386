#!/bin/bash
# round 4, session a: the reference's own package on the HIP solvers (new test), the compact bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 900 python -m pytest tests/test_gpu_reference_suite.py -x -q > $O/refsuite.log 2>&1; echo "refsuite rc=$?"; tail -5 $O/refsuite.log
timeout 1200 python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; tail -3 $O/bench.err
cat $O/bench_line.json
