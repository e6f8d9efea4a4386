#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GEORGE_AMD_LOOKAHEAD_DEPTH=1
echo "== depth 1, exclusive chain CUs + split bcol"; timeout 600 python scripts/size_sweep.py 2048 4096 8192 12288 16384 20480 24576 32768 2>&1 | grep "^| [0-9]"
echo "== no exclusive"; GEORGE_AMD_PANEL_EXCLUSIVE=0 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^| [0-9]"
echo "== no bcol split"; GEORGE_AMD_NO_BCOL_SPLIT=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^| [0-9]"
echo "== side masked"; GEORGE_AMD_PANEL_SIDE_MASKED=1 timeout 600 python scripts/size_sweep.py 4096 8192 16384 2>&1 | grep "^| [0-9]"
echo "== depth 2"; GEORGE_AMD_LOOKAHEAD_DEPTH=2 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^| [0-9]"
echo "== reserve 16"; GEORGE_AMD_RESERVE_CUS=16 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^| [0-9]"
echo "== reserve 64"; GEORGE_AMD_RESERVE_CUS=64 timeout 600 python scripts/size_sweep.py 8192 16384 2>&1 | grep "^| [0-9]"
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x --timeout 900 -k "not 65536 and not NS and not C3 and not c3" 2>&1 | tail -3
