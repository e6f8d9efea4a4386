#!/bin/bash
# gpurun session: full -m gpu suite, then the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout 900 > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/t_all.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
