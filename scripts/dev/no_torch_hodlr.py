"""HODLR C4 through the NumPy facade in a process that never imports torch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from george_amd import GP, kernels, HODLRSolver
n = 262144
rng = np.random.RandomState(1234)
x = np.sort(rng.uniform(0, 10, n)); yerr = 0.1 * np.ones(n); y = np.sin(x)
gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0), solver=HODLRSolver, tol=1e-10, min_size=100, seed=42)
for i in range(4):
    gp.compute(x, yerr); ll = gp.log_likelihood(y)
t0 = time.perf_counter()
for i in range(10):
    gp.compute(x, yerr); ll = gp.log_likelihood(y)
print("C4 through GP(solver=HODLRSolver) on NumPy arrays: %.3f ms per compute+log_likelihood (torch imported: %s; prime: %s)  ll %.6f" % (
    (time.perf_counter() - t0) / 10 * 1e3, "torch" in sys.modules, "no" if os.environ.get("GEORGE_AMD_NO_NULL_PRIME") else "yes", ll))
