"""What the Python facade adds to a small-N step: GP.compute(x, yerr) + GP.log_likelihood(y) as an optimiser loop
issues them (new parameter vector every iterate), against the raw ABI calls of bench.DenseJob; cProfile of the former."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from george_amd import GP, kernels

for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]:
    x, yerr, y = bench.make_inputs(n)
    gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
    p0 = gp.get_parameter_vector()

    def step(i):
        gp.set_parameter_vector(p0 + 1e-6 * (i % 7))
        gp.compute(x, yerr)
        return gp.log_likelihood(y)
    for i in range(20):
        step(i)
    t0 = time.perf_counter()
    for i in range(200):
        step(i)
    t_facade = (time.perf_counter() - t0) / 200
    def nll(i):
        return gp.nll(p0 + 1e-6 * (i % 7), y)
    for i in range(20):
        nll(i)
    t0 = time.perf_counter()
    for i in range(200):
        nll(i)
    t_nll = (time.perf_counter() - t0) / 200
    job = bench.DenseJob(n, 0, 0, profile=False)
    for i in range(20):
        job.step()
    t0 = time.perf_counter()
    for i in range(200):
        job.step()
    t_raw = (time.perf_counter() - t0) / 200
    job.close()
    print("N=%5d  GP.compute+log_likelihood %.3f ms | gp.nll(p, y) (fused objective) %.3f ms | raw ABI, device-resident inputs %.3f ms" % (n, t_facade * 1e3, t_nll * 1e3, t_raw * 1e3))
    if n == 1024:
        pr = cProfile.Profile()
        pr.enable()
        for i in range(300):
            step(i)
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(14)
