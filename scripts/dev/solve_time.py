"""apply_inverse(y) (one right-hand side: forward + backward chained sweeps) at a few sizes.  usage: solve_time.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from george_amd import BasicSolver, kernels
for n in [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]:
    x, yerr, y = bench.make_inputs(n)
    s = BasicSolver(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
    s.compute(x[:, None], np.sqrt(yerr ** 2 + 1.25e-12))
    for _ in range(3):
        a = s.apply_inverse(y)
    t0 = time.perf_counter()
    for _ in range(10):
        a = s.apply_inverse(y)
    dt = (time.perf_counter() - t0) / 10
    print("N=%6d apply_inverse(y) %.3f ms (host clock, H2D + two sweeps + D2H)  |alpha|max %.6e" % (n, dt * 1e3, np.abs(a).max()))
    del s
