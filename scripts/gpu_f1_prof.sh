#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/f1; O=/root/repo/gpurun_out/f1; export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/trace -o f1 -- python /root/repo/scripts/dev/f1_time.py nocpu > $O/f1.log 2>&1; echo "rc=$?"; tail -5 $O/f1.log
cd /root/repo
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
find $O -name "*.db" -size +6M -delete
