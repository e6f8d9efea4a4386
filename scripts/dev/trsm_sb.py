"""K^-1 B for many right-hand sides (device-resident B): time of gh_chol_solve.  GEORGE_AMD_TRSM_SB = tiles per super-block."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from george_amd import _native as N
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
job = bench.DenseJob(n, 0, 0, profile=False)
job.step()
B = torch.randn(n, m, dtype=torch.float64, device="cuda")
out = torch.empty_like(B)
for i in range(2):
    N.check(N.lib.gh_chol_solve(job.h, B.data_ptr(), m, out.data_ptr()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    N.check(N.lib.gh_chol_solve(job.h, B.data_ptr(), m, out.data_ptr()))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("SB=%s N=%d M=%d: K^-1 B in %.1f ms = %.1f TFLOP/s (2 N^2 M)  checksum %.9e" % (os.environ.get("GEORGE_AMD_TRSM_SB", "4"), n, m, dt * 1e3, 2.0 * n * n * m / dt * 1e-12, float(out.abs().sum())))
