"""gh_mgpu_* with ONE device (RCCL world of one) against gh_chol_* on the same problem: what the sharded
driver's loop costs on a single GPU with no communication (per-tile-column GEMMs instead of one wide SYRK,
no look-ahead), and the virtual-device grids (copy transport, every rank on device 0) for the ordering
logic at full size.    python scripts/mgpu_one_device.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from george_amd import BasicSolver, MultiGPUSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
x, yerr, y = bench.make_inputs(n)
kernel = bench.make_kernel("expsquared", np.var(y))
X = x[:, None]
sig = np.sqrt(yerr ** 2 + 1.25e-12)
d = BasicSolver(kernel)
d.compute(X, sig)
t0 = time.perf_counter(); d.compute(X, sig); q0 = d.dot_solve(y); t_d = time.perf_counter() - t0
for devices, transport in (([0], "rccl"), ([0, 0], "copy"), ([0, 0, 0, 0], "copy")):
    s = MultiGPUSolver(kernel, devices=devices, transport=transport)
    s.compute(X, sig)
    t0 = time.perf_counter(); s.compute(X, sig); q = s.dot_solve(y); t_s = time.perf_counter() - t0
    pr, pc, nb = s.grid_shape()
    print("N=%d  gh_mgpu %d rank(s) %s grid %dx%d nb=%d: %.1f ms (%.1f TFLOP/s) | gh_chol: %.1f ms | ratio %.2f | "
          "rel log-det diff %.1e, rel quad diff %.1e"
          % (n, len(devices), transport, pr, pc, nb, t_s * 1e3, bench.flops_alg(n) / t_s * 1e-12, t_d * 1e3, t_s / t_d,
             abs(s.log_determinant - d.log_determinant) / abs(d.log_determinant), abs(q - q0) / abs(q0)), flush=True)
    del s
