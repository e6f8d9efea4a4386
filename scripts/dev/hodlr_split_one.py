"""One configuration of the HODLR split, every compute() timed on its own.  usage: hodlr_split_one.py N P [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zoo
import george_amd
from george_amd import kernels, MultiGPUHODLRSolver

n, P = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ndev = max(george_amd.device_count(), 1)
x, yerr, y = zoo.bench_data(n)
kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
X = np.ascontiguousarray(x[:, None])
s = MultiGPUHODLRSolver(kernel, devices=[i % ndev for i in range(P)], tol=1e-10, min_size=100, seed=42)
for it in range(reps):
    t0 = time.perf_counter()
    s.compute(X, yerr)
    t1 = time.perf_counter()
    q = s.dot_solve(y)
    t2 = time.perf_counter()
    print("compute %.2f ms, dot_solve %.2f ms, ll part %.9f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, -0.5 * (q + s.log_determinant)), flush=True)
