#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/hodlr_prof50k; cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/hodlr_prof50k -o hp -- python /root/repo/scripts/dev/hodlr_prof_step.py 50000 > /root/repo/gpurun_out/hodlr_prof50k.log 2>&1; echo "prof rc=$?"
