"""Does the time of a mid-size step depend on what touched the HIP runtime before the first solver handle was made?
usage: stream_order_probe.py {torch_first|george_first} [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
import bench
from george_amd import GP, kernels
if mode == "torch_first":
    import torch
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
x, yerr, y = bench.make_inputs(256)
gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
gp.compute(x, yerr)                         # the first solver handle of the process (creates the shared streams)
gp.log_likelihood(y)
for m in (n, 8192):
    job = bench.DenseJob(m, 0, 0, profile=False)
    for i in range(5):
        job.step()
    t0 = time.perf_counter()
    for i in range(20):
        job.step()
    print("%s: N=%d raw ABI step %.3f ms" % (mode, m, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    job.close()
