#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 150 python scripts/dev/hodlr_passes_ab.py 4096 262144 > gpurun_out/hodlr_passes_ab2.log 2>&1; echo "ab rc=$?"; tail -9 gpurun_out/hodlr_passes_ab2.log
cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/hodlr_prof -o hp -- python /root/repo/scripts/dev/hodlr_prof_step.py 262144 > /root/repo/gpurun_out/hodlr_prof.log 2>&1; echo "prof rc=$?"
cd /root/repo; f=$(ls gpurun_out/hodlr_prof/*/*kernel_stats.csv gpurun_out/hodlr_prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -40 "$f"
