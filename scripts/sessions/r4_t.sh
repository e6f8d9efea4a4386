#!/bin/bash
# round 4, session t: traces of the 2 x 1 and 4 x 1 snake grids, merged into the committed collection (predicted scaling curve)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python scripts/scale_model.py collect $O/scale_traces.json --configs 2x1x1024,4x1x1024 --merge profiles/r04/scale_traces_N65536.json.gz > $O/scale_collect.log 2>&1; echo "collect rc=$?"; cut -c1-260 $O/scale_collect.log | tail -4
gzip -9 -k $O/scale_traces.json; ls -la $O | tail -3
