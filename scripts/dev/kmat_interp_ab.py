"""Kernel-matrix build of the hyper.rst kernel (through the postfix walker): the build's HIP-event time inside compute() (gh_chol
profile) and the wall time of grad_log_likelihood().  python scripts/dev/kmat_interp_ab.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from george_amd import GP, kernels, BasicSolver, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t, y = bench.f1_data(n)
for mode in (0, 1):
    gp = GP(bench.f1_kernel(kernels), mean=float(np.mean(y)), white_noise=np.log(0.19 ** 2), solver=BasicSolver, profile=True)
    for rep in range(3):
        gp.compute(t); ll = gp.log_likelihood(y)
    b = gp.solver.profile()["ms_build"]
    import time
    g = gp.grad_log_likelihood(y)
    t0 = time.perf_counter(); g = gp.grad_log_likelihood(y); tg = (time.perf_counter() - t0) * 1e3
    print("rep %d: build %.3f ms  grad %.2f ms  ll %.15g  grad0 %.15g" % (mode, b, tg, ll, g[0]), flush=True)
    del gp
