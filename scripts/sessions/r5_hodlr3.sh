#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 150 python scripts/dev/hodlr_passes_ab.py 4096 > gpurun_out/hodlr_passes_smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/hodlr_passes_smoke.log
timeout -s KILL 400 python -X faulthandler -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py -x -q -m gpu -p no:cacheprovider > gpurun_out/hodlr_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/hodlr_tests.log
timeout -s KILL 400 python scripts/dev/hodlr_passes_ab.py > gpurun_out/hodlr_passes_ab7.log 2>&1; echo "ab rc=$?"; tail -18 gpurun_out/hodlr_passes_ab7.log
rm -rf gpurun_out/hodlr_prof; cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/hodlr_prof -o hp -- python /root/repo/scripts/dev/hodlr_prof_step.py 262144 > /root/repo/gpurun_out/hodlr_prof.log 2>&1; echo "prof rc=$?"
cd /root/repo; ls gpurun_out/hodlr_prof | head
