"""A process that never imports torch: GP.compute(x, yerr) + GP.log_likelihood(y) on NumPy arrays.  usage: no_torch_step.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from george_amd import GP, kernels
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192, 16384]:
    rng = np.random.RandomState(1234)
    x = np.sort(rng.uniform(0, 10, n)); yerr = 0.1 * np.ones(n); y = np.sin(x)
    gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
    for i in range(4):
        gp.compute(x, yerr); ll = gp.log_likelihood(y)
    t0 = time.perf_counter()
    for i in range(10):
        gp.compute(x, yerr); ll = gp.log_likelihood(y)
    print("N=%6d GP.compute+log_likelihood %.3f ms (torch imported: %s)  ll %.6f" % (n, (time.perf_counter() - t0) / 10 * 1e3, "torch" in sys.modules, ll), flush=True)
