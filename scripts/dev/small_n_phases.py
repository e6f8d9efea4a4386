"""Where a small-N compute()+log_likelihood() goes: device phases from the handle's HIP events against the host clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
for n in (1024, 2048, 4096, 8192):
    job = bench.DenseJob(n, 0, 0, profile=True)
    for _ in range(5): job.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): job.step()
    torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / 20 * 1e3
    p = job.profile()
    job2 = bench.DenseJob(n, 0, 0, profile=False)
    for _ in range(5): job2.step()
    t0 = time.perf_counter()
    for _ in range(20): job2.step()
    torch.cuda.synchronize()
    host2 = (time.perf_counter() - t0) / 20 * 1e3
    import ctypes as C
    N = job2.N
    ld, q = C.c_double(0), C.c_double(0)
    t0 = time.perf_counter()
    for _ in range(20):
        N.check(N.lib.gh_chol_compute(job2.h, job2.dk.handle, job2.x.data_ptr(), n, 1, job2.yerr.data_ptr(), C.byref(ld)))
    tc = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        N.check(N.lib.gh_chol_dot_solve(job2.h, job2.y.data_ptr(), C.byref(q)))
    ts = (time.perf_counter() - t0) / 20 * 1e3
    print("N=%5d  host clock per step %.3f ms (profile on) / %.3f (off) = compute %.3f + dot_solve %.3f | device: compute %.3f ms "
          "(build %.3f, panels %.3f, trailing %.3f), solve %.3f" % (n, host, host2, tc, ts, p.ms_total, p.ms_build, p.ms_panel, p.ms_trailing, p.ms_solve))
