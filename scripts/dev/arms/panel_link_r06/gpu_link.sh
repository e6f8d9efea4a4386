#!/bin/bash
# the one-launch chain link: against the three-launch chain, A/B timings, kernel timeline at N = 2048
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -k "link_kernel" 2>&1 | tail -5
timeout 900 python scripts/dev/panel_link_ab.py ${SIZES:-1024 4096 16384} 2>&1 | tee gpurun_out/panel_link_ab.txt | tail -40
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lk && timeout 300 rocprofv3 --kernel-trace -d /tmp/lk -o lk -- python $GRAFT_REPO_ROOT/scripts/dev/no_torch_step.py 2048 > /tmp/lk.log 2>&1; echo "rocprof rc=$?"
python $GRAFT_REPO_ROOT/scripts/dev/step_timeline.py "$(find /tmp/lk -name '*.db' | head -1)" 2>&1 | tail -60 > $GRAFT_REPO_ROOT/gpurun_out/link_timeline.txt
tail -45 $GRAFT_REPO_ROOT/gpurun_out/link_timeline.txt | cut -c1-150
