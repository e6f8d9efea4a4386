#!/bin/bash
# round 4, session u: the whole -m gpu suite + smoke on the final tree, then the round's rocprofv3 / PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
bash scripts/profile_r04.sh > $O/profile.log 2>&1; echo "profile rc=$?"; tail -30 $O/profile.log | cut -c1-300
