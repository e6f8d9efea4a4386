"""Time HODLR compute()+log_likelihood() with the tree split over P sub-trees -- on ONE GPU the same device is
listed P times ("virtual devices"): what is measured is the cost of the protocol (host threads, barriers, the
pinned-memory sums of the top levels), not a speed-up.  usage: hodlr_split_time.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import zoo
import george_amd
from george_amd import kernels, HODLRSolver, MultiGPUHODLRSolver

ndev_phys = max(george_amd.device_count(), 1)
for n in [int(a) for a in sys.argv[1:]] or [262144, 2097152]:
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    X = np.ascontiguousarray(x[:, None])
    kw = dict(tol=1e-10, min_size=100, seed=42)
    rows = []
    for P in (0, 1, 2, 4, 8):
        if P == 0:
            s = HODLRSolver(kernel, **kw)
        else:
            s = MultiGPUHODLRSolver(kernel, devices=[i % ndev_phys for i in range(P)], **kw)
        best = 1e30
        for it in range(6):
            t0 = time.perf_counter()
            s.compute(X, yerr)
            ll = -0.5 * (s.dot_solve(y) + s.log_determinant)
            dt = time.perf_counter() - t0
            if it >= 2:
                best = min(best, dt)
        rows.append((P, best * 1e3, ll))
    print("N = %d (%d physical device%s)" % (n, ndev_phys, "" if ndev_phys == 1 else "s"))
    print("| solver | ms per compute()+log_likelihood() | log-likelihood kernel part |")
    print("|---|---|---|")
    for P, ms, ll in rows:
        print("| %s | %.2f | %.9f |" % ("HODLRSolver (gh_hodlr_*)" if P == 0 else "split over %d sub-tree%s" % (P, "" if P == 1 else "s"), ms, ll))
