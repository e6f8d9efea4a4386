#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 300 python -X faulthandler -m pytest tests/test_gpu_hodlr.py -x -q -m gpu -p no:cacheprovider -k "blocked or shared or positive_definite" > gpurun_out/hodlr_newtests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/hodlr_newtests.log
rm -rf gpurun_out/hodlr_prof700k; cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/hodlr_prof700k -o hp -- python /root/repo/scripts/dev/hodlr_prof_step.py 700000 > /root/repo/gpurun_out/hodlr_prof700k.log 2>&1; echo "prof rc=$?"
