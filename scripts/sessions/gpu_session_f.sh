#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_chain
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== exclusive crit, side prio, no inner"; GEORGE_AMD_PANEL_SIDE_PRIO=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^|"
echo "== exclusive crit, side prio, inner split"; GEORGE_AMD_PANEL_SIDE_PRIO=1 GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^|"
echo "== same, 16 CUs"; GEORGE_AMD_RESERVE_CUS=16 GEORGE_AMD_PANEL_SIDE_PRIO=1 GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^|"
echo "== same, 8 CUs"; GEORGE_AMD_RESERVE_CUS=8 GEORGE_AMD_PANEL_SIDE_PRIO=1 GEORGE_AMD_PANEL_INNER_SPLIT=1 GEORGE_AMD_PANEL_EXCLUSIVE=1 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^|"
