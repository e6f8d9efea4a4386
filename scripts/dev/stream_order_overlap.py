"""Overlap matrix of the process-wide streams (gh_debug_stream_overlap: 0.3 = run side by side, 0.6 = share a hardware
queue) depending on what touched the HIP runtime first.  usage: stream_order_overlap.py {torch_first|george_first}"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
mode = sys.argv[1]
import bench
from george_amd import GP, kernels, _native as N
if mode == "torch_first":
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
x, yerr, y = bench.make_inputs(256)
gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
gp.compute(x, yerr); gp.log_likelihood(y)
job = bench.DenseJob(8192, 0, 0, profile=False)
for i in range(3):
    job.step()
out = (C.c_double * 36)()
N.check(N.lib.gh_debug_stream_overlap(job.h, out, 36))
names = ["null", "main", "chain", "rows", "near", "masked"]
print(mode, " ".join("%s|%s=%.2f" % (names[a], names[b], out[a * 6 + b]) for a in range(6) for b in range(a + 1, 6)))
t0 = time.perf_counter()
for i in range(10):
    job.step()
print(mode, "N=8192 step %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
job.close()
job = bench.DenseJob(8192, 0, 0, profile=True)
for i in range(4):
    job.step()
p = job.profile()
print(mode, "profiled last step: total %.2f build %.2f panel %.2f trailing %.2f [%d launches] solve %.2f union %.2f ms" % (
    p.ms_total, p.ms_build, p.ms_panel, p.ms_trailing, p.n_trailing, p.ms_solve, p.ms_update_union))
iv = job.update_intervals()
print(mode, "update launches (start, end ms):", " ".join("%.2f-%.2f" % (a, b) for a, b, f in iv[:24]))
