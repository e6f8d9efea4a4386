#!/bin/bash
# HODLR work session: the wave-ACA tests, the A/B, the C4 timeline.   gpurun --timeout 1500 -- 'bash scripts/gpu_hodlr_session.sh'
cd /root/repo; mkdir -p gpurun_out/hodlr; export TMPDIR=/tmp
O=gpurun_out/hodlr
timeout -s KILL 600 python -X faulthandler -m pytest tests/test_gpu_hodlr.py -x -q -m gpu -p no:cacheprovider -k "wave or shared_passes or golden or blocked" > $O/tests.log 2>&1; echo "hodlr tests rc=$?"; tail -4 $O/tests.log
timeout -s KILL 400 python scripts/dev/hodlr_wave_ab.py 262144 32768 50000 1048576 > $O/wave_ab.md 2>&1; echo "ab rc=$?"; cat $O/wave_ab.md
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /root/repo/$O/traceC4 -o trace -- python /root/repo/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu > /root/repo/$O/traceC4.log 2>&1; echo "trace rc=$?"
cd /root/repo
f=$(find $O/traceC4 -name "*.db" | head -1); [ -n "$f" ] && python scripts/dev/hodlr_timeline.py "$f" > $O/hodlr_timeline_C4.txt && python scripts/hodlr_levels.py "$f" > $O/hodlr_levels_C4.txt
find $O -name "*.db" -size +6M -delete
head -80 $O/hodlr_timeline_C4.txt
