"""BASELINE config C5: N=32768, 3-D isotropic metric, Matern52 + ConstantKernel, predict() mean+var at
M=4096 test points and grad_log_likelihood, fp64, one MI355X."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from george_amd import GP, kernels

def main(n=32768, m=4096):
    rng = np.random.RandomState(1234)
    x = rng.uniform(0, 1, (n, 3)); x = x[np.argsort(x[:, 0])]
    y = np.sin(x.sum(axis=1))
    t = rng.uniform(0, 1, (m, 3))
    kernel = kernels.Matern52Kernel(0.5, ndim=3) + kernels.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    gp = GP(kernel)
    out = {"N": n, "M": m}
    for rep in range(2):
        t0 = time.perf_counter(); gp.compute(x, 0.1); ll = gp.log_likelihood(y); out["compute_loglike_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); mu, var = gp.predict(y, t, return_var=True); out["predict_var_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); g = gp.grad_log_likelihood(y); out["grad_s"] = time.perf_counter() - t0
    out.update(loglike=float(ll), grad=[float(v) for v in g], mu0=float(mu[0]), var0=float(var[0]))
    print(json.dumps(out))

if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
