#!/bin/bash
# dense solver: its GPU tests + a size sweep on the tree as built
cd /root/repo; export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
python - <<'PY'
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, bench, torch
for n in (1024, 2048, 4096, 8192, 16384, 24064, 65536):
    job = bench.DenseJob(n, 0, 0, profile=False)
    for i in range(3): job.step()
    ts = []
    for i in range(15 if n <= 24064 else 4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ll = job.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("N = %6d: %.3f / %.3f ms  ll %.12g" % (n, min(ts), float(np.median(ts)), ll), flush=True)
    job.close()
PY
