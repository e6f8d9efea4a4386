"""The multi-GPU tile driver (george_amd/distributed.py) as a world of ONE rank on the leased GPU: what the
2-D block-cyclic loop costs against the single-GPU solver on the same problem -- per-tile-column GEMM
launches instead of one wide SYRK, Python issue overhead -- with no communication at all.
    python scripts/tile_driver_world1.py [N] [nb]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from george_amd.distributed import DistributedDenseJob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
job = DistributedDenseJob(n, nb, 0, bench.make_inputs, kernel=bench.make_kernel)
ll = job.step()
job.reset_profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 2
for _ in range(steps):
    ll = job.step()
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / steps
ms, fl, calls = job.chol.update_profile()
d = bench.DenseJob(n, 0, 0, profile=False)
d.step()
t0 = time.perf_counter()
ll1 = d.step()
sec1 = time.perf_counter() - t0
print("N=%d nb=%d: tile driver (world of one) %.1f ms = %.1f TFLOP/s | single-GPU solver %.1f ms = %.1f TFLOP/s | ratio %.3f | "
      "rel diff of log-likelihood %.2e | trailing updates: %.1f ms per step in %d sweeps, %.1f TFLOP/s"
      % (n, job.nb, sec * 1e3, bench.flops_alg(n) / sec * 1e-12, sec1 * 1e3, bench.flops_alg(n) / sec1 * 1e-12, sec / sec1,
         abs(ll - ll1) / abs(ll1), ms / steps, calls // steps, fl / (ms * 1e-3) * 1e-12 if ms else 0.0))
