#!/bin/bash
# round 5: the whole -m gpu suite (with the new reference-kernel boundary test and the dataflow tests), smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -15 > gpurun_out/r5_suite.log
tail -15 gpurun_out/r5_suite.log
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
