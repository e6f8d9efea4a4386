"""TEST INFRASTRUCTURE ONLY.  Reference scalars at the FULL sizes of BASELINE.json's configs.

Runs the REAL reference in the build container -- its compiled C++ kernel evaluator
(oracle/_ref, built from /root/reference/src/george/kernel_interface.cpp) for K and the very
LAPACK calls of ``src/george/solvers/basic.py:68,87`` (SciPy ``cholesky(lower=False)`` /
``cho_solve``) -- and writes ``tests/golden/large.json``: log-determinant, log-likelihood and a
strided sample of ``alpha = K^-1 y`` for

  C2      N=16384  1-D ExpSquared                      (BASELINE configs[1])
  M32_20k N=20480  1-D Matern32
  NS      N=65536  1-D ExpSquared                      (north-star target / bench.py headline)
  C3      N=65536  1-D Matern32                        (BASELINE configs[2])
  C5      N=32768  3-D Matern52 + Constant             (BASELINE configs[4]): + predict mean/var at
                                                       64 of the 4096 test points, + gradient

so that the ``-m gpu`` tests can compare the HIP path with the reference at the sizes the claims
are made on without running minutes of LAPACK on the GPU box.

Memory: the reference hands ``cholesky`` a C-contiguous K, which f2py copies into Fortran order
(68.7 GB at N=65536).  K is exactly symmetric (``kernel_interface.cpp:62-77`` writes both halves),
so the Fortran-ordered *view* ``K.T`` holds the same values and the same ``dpotrf('U')`` runs in place.

    OPENBLAS_NUM_THREADS=6 python -m oracle.gen_golden_large [names...]
"""
import json
import os
import sys
import time

import numpy as np
from scipy.linalg import cholesky, cho_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_loader  # noqa: E402
import zoo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "large.json")
TINY = 1.25e-12                      # src/george/gp.py:19


def _save(res):
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def dense_case(george, kernel, x, yerr, y, want_inverse_for_grad=False, t=None):
    """GP.compute + log_likelihood (gp.py:303-337,369-397) with basic.py's LAPACK calls, in place."""
    n = len(x)
    x2 = np.ascontiguousarray(x.reshape(n, -1))
    t0 = time.time()
    K = kernel.get_value(x2)                                         # basic.py:64 -> value_symmetric
    t_build = time.time() - t0
    K[np.diag_indices_from(K)] += yerr ** 2 + TINY                   # basic.py:65 with gp.py:330
    t0 = time.time()
    U = cholesky(K.T, overwrite_a=True, lower=False, check_finite=False)     # basic.py:68
    assert np.shares_memory(U, K)
    t_fac = time.time() - t0
    logdet = 2 * np.sum(np.log(np.diag(U)))                          # basic.py:69
    alpha = cho_solve((U, False), y, check_finite=False)             # basic.py:87
    q = float(np.dot(y, alpha))                                      # basic.py:102
    ll = -0.5 * (n * np.log(2 * np.pi) + logdet) - 0.5 * q           # gp.py:333-335,396
    out = {"n": n, "logdet": float(logdet), "loglike": float(ll), "quad": q,
           "alpha_stride": max(n // 64, 1), "alpha": [float(v) for v in alpha[::max(n // 64, 1)]],
           "seconds_build": t_build, "seconds_factor": t_fac}
    if t is not None:                                                # gp.py:532-541
        Kxs = kernel.get_value(t, x2)
        mu = np.dot(Kxs, alpha)
        KinvKxs = cho_solve((U, False), Kxs.T, check_finite=False)
        var = kernel.get_value(t, diag=True) - np.sum(Kxs.T * KinvKxs, axis=0)
        out["t"] = [[float(v) for v in row] for row in t]
        out["mu"] = [float(v) for v in mu]
        out["var"] = [float(v) for v in var]
    if want_inverse_for_grad:                                        # gp.py:436-437,465-466, blocked over rows
        t0 = time.time()
        Kinv = np.eye(n)
        Kinv = cho_solve((U, False), Kinv, overwrite_b=True, check_finite=False)    # basic.py:121
        del U, K
        ki = kernel.kernel
        which = np.ones(kernel.full_size, dtype=np.uint32)
        g = np.zeros(kernel.full_size)
        step = 1024
        for i0 in range(0, n, step):
            Kg = ki.gradient_general(which, x2[i0:i0 + step], x2)    # same entries as gradient_symmetric's
            A = np.outer(alpha[i0:i0 + step], alpha) - Kinv[i0:i0 + step]
            g += 0.5 * np.einsum("ijk,ij", Kg, A)
        out["grad"] = [float(v) for v in g]
        out["grad_names"] = list(kernel.get_parameter_names(include_frozen=True))
        out["seconds_grad"] = time.time() - t0
    return out


def main():
    george = ref_loader.load_reference()
    if george is None:
        raise SystemExit("reference not available (need /root/reference and oracle/_ref built: make -C oracle)")
    K = george.kernels
    want = sys.argv[1:] or ["C2", "M32_20k", "NS", "C3", "C5"]
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        t0 = time.time()
        if name == "C5":
            x, yerr, y = zoo.bench_data(32768, ndim=3)
            kernel = K.Matern52Kernel(0.5, ndim=3) + K.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
            t = np.random.RandomState(4321).uniform(0, 1, (4096, 3))[:64].copy()
            res[name] = dense_case(george, kernel, x, yerr, y, want_inverse_for_grad=True, t=t)
        else:
            n, cls = {"C2": (16384, K.ExpSquaredKernel), "M32_20k": (20480, K.Matern32Kernel),
                      "NS": (65536, K.ExpSquaredKernel), "C3": (65536, K.Matern32Kernel)}[name]
            x, yerr, y = zoo.bench_data(n)
            res[name] = dense_case(george, np.var(y) * cls(1.0), x, yerr, y)
        res[name]["seconds_total"] = time.time() - t0
        res[name]["generator"] = "oracle/gen_golden_large.py (reference C++ evaluator + scipy %s LAPACK)" % (
            __import__("scipy").__version__)
        print(name, {k: v for k, v in res[name].items() if not isinstance(v, list)}, flush=True)
        _save(res)


if __name__ == "__main__":
    main()
