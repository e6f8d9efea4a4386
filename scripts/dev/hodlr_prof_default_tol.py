"""four steps of bench.HodlrJob(N, tol = 0.1) -- the reference's own benchmark configuration (docs/tutorials/scaling.rst) -- for
rocprofv3 --kernel-trace:  rocprofv3 --kernel-trace -d /tmp/d -o t -- python scripts/dev/hodlr_prof_default_tol.py 50000"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
job = bench.HodlrJob(n, 0, tol=float(sys.argv[2]) if len(sys.argv) > 2 else 0.1)
for i in range(4): job.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): job.step()
torch.cuda.synchronize(); print("N = %d: %.3f ms per step" % (n, (time.perf_counter() - t0) / 10 * 1e3))
