"""compute()+log_likelihood() (bench.DenseJob) with the kernel-matrix split diagonal kernel (1) or on the main stream (0):
gh_debug_set_potf2_split, one process.  python scripts/dev/build_on_chain_ab.py [sizes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 16384, 32768]
print("| N | split diagonal kernel | ms min / median | log-likelihood |\n|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(2):
        for mode in (0, 1):
            N.lib.gh_debug_set_potf2_split(mode)
            job = bench.DenseJob(n, 0, 0, profile=False)
            ts = []
            for rep in range(3 + (14 if n <= 16384 else 5)):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).extend(ts); res[(mode, "ll")] = float(v)
            job.close()
    for mode in (0, 1):
        print("| %d | %d | %.3f / %.3f | %.15g |" % (n, mode, min(res[mode]), float(np.median(res[mode])), res[(mode, "ll")]), flush=True)
N.lib.gh_debug_set_potf2_split(1)
