#!/bin/bash
# round 4, session p: scale-model traces again on the final tree (grouped tile order in the per-row update launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python scripts/scale_model.py collect $O/scale_traces.json > $O/scale_collect.log 2>&1; echo "collect rc=$?"; cut -c1-330 $O/scale_collect.log | tail -8
gzip -9 -k $O/scale_traces.json; ls -la $O
