#!/bin/bash
# HODLR: tests of the solver + C4 / other sizes timed (median of 15) on the tree as built
cd /root/repo; mkdir -p gpurun_out/hodlr; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_hodlr.py tests/test_gpu_hodlr_split.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
python - <<'PY'
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, bench, torch
for n in (262144, 50000, 1048576):
    job = bench.HodlrJob(n, 0)
    for i in range(3): job.step()
    ts = []
    for i in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ll = job.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("N = %7d: %.3f / %.3f / %.3f ms  ll %.12g" % (n, min(ts), float(np.median(ts)), max(ts), ll), flush=True)
    del job
PY
