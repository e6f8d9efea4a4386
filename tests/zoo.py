"""Kernel / GP configurations shared by the golden-vector generator (which builds them with the
REFERENCE's ``george.kernels``) and the tests (which build them with ``george_amd.kernels``)."""
import numpy as np


def _spd(ndim, seed):
    rng = np.random.RandomState(seed)
    L = rng.randn(ndim, ndim)
    L[np.diag_indices(ndim)] = np.exp(L[np.diag_indices(ndim)])
    L[np.triu_indices(ndim, 1)] = 0.0
    return np.dot(L, L.T)


def kernel_zoo(K):
    """(name, kernel) pairs covering every leaf, operator, metric type, axes and block
    (mirrors the reference's tests/test_kernels.py:19-63,83-128 and tests/test_metrics.py)."""
    z = [
        ("const1", K.ConstantKernel(log_constant=0.1)),
        ("const5", K.ConstantKernel(log_constant=5.0, ndim=5)),
        ("dot5ax0", K.DotProductKernel(ndim=5, axes=0)),
        ("dot2", K.DotProductKernel(ndim=2)),
        ("cos1", K.CosineKernel(log_period=1.0)),
        ("cos5ax23", K.CosineKernel(log_period=0.75, ndim=5, axes=[2, 3])),
        ("es2a", K.ExpSine2Kernel(gamma=0.4, log_period=1.0)),
        ("es2b", K.ExpSine2Kernel(gamma=13.7, log_period=-0.75, ndim=5, axes=[2, 3])),
        ("es2neg", K.ExpSine2Kernel(gamma=-0.7, log_period=0.75, ndim=5, axes=[2, 3])),
        ("lg1", K.LocalGaussianKernel(log_width=0.5, location=1.0)),
        ("lg5", K.LocalGaussianKernel(log_width=2.0, location=0.75, ndim=5, axes=[2, 3])),
        ("lin0", K.LinearKernel(order=0, log_gamma2=0.0)),
        ("lin2", K.LinearKernel(order=2, log_gamma2=0.0)),
        ("lin3ax2", K.LinearKernel(order=3, log_gamma2=-1.0, ndim=5, axes=2)),
        ("linsum", K.LinearKernel(order=0, log_gamma2=0.0) + K.LinearKernel(order=1, log_gamma2=-1.0)
         + K.LinearKernel(order=2, log_gamma2=-2.0)),
        ("poly0", K.PolynomialKernel(order=0, log_sigma2=-10.0)),
        ("poly2", K.PolynomialKernel(order=2, log_sigma2=0.0)),
        ("poly3ax2", K.PolynomialKernel(order=3, log_sigma2=-1.0, ndim=5, axes=2)),
        ("es2scaled", 12. * K.ExpSine2Kernel(gamma=0.4, log_period=1.0, ndim=5)),
        ("c5like", 12. * K.ExpSquaredKernel(0.4, ndim=3) + 0.1),
        ("empty_sum", K.Matern32Kernel(2.0, ndim=2) + K.EmptyKernel(ndim=2)),
    ]
    stationary = [("exp", K.ExpKernel, {}), ("expsq", K.ExpSquaredKernel, {}), ("m32", K.Matern32Kernel, {}),
                  ("m52", K.Matern52Kernel, {}), ("rq1", K.RationalQuadraticKernel, dict(log_alpha=np.log(1.0))),
                  ("rq01", K.RationalQuadraticKernel, dict(log_alpha=np.log(0.1)))]
    for nm, cls, kw in stationary:
        z += [
            (nm + "_iso01", cls(metric=0.1, **kw)),
            (nm + "_iso10", cls(metric=10.0, **kw)),
            (nm + "_axis", cls(metric=[1.0, 0.1, 10.0], ndim=3, **kw)),
            (nm + "_iso3", cls(metric=1.0, ndim=3, **kw)),
            (nm + "_ax2", cls(metric=1.0, ndim=3, axes=2, **kw)),
            (nm + "_block", cls(metric=1.0, ndim=3, axes=2, block=(-0.1, 0.1), **kw)),
            (nm + "_general", cls(metric=_spd(3, 7), ndim=3, **kw)),
        ]
    z += [
        ("prod_mixed", K.Matern32Kernel([1.0, 0.1, 10.0], ndim=3) * K.Matern52Kernel(0.3, ndim=3, axes=2, block=(-0.5, 0.5))),
        ("deep", (0.5 * K.ExpSquaredKernel(1.3, ndim=2) + K.CosineKernel(log_period=0.3, ndim=2, axes=1))
         * (K.Matern52Kernel([0.7, 1.9], ndim=2) + 2.0) + K.DotProductKernel(ndim=2)),
    ]
    return z


def scaling_data(n):
    """The reference's only benchmark inputs (docs/tutorials/scaling.rst:56-59): first n points of the
    sorted 50 000-point array."""
    rng = np.random.RandomState(1234)
    x = np.sort(rng.uniform(0, 10, 50000))
    y = np.sin(x)
    return x[:n], 0.1 * np.ones(n), y[:n], np.var(y)      # kernel amplitude is var of ALL 50 000 y (scaling.rst:67)


def bench_data(n, ndim=1, seed=1234):
    """SURVEY.md 8(d) synthetic inputs for the configs C1-C5."""
    rng = np.random.RandomState(seed)
    if ndim == 1:
        x = np.sort(rng.uniform(0, 10, n))
        return x, 0.1 * np.ones(n), np.sin(x)
    x = rng.uniform(0, 1, (n, ndim))
    x = x[np.argsort(x[:, 0])]
    return x, 0.1 * np.ones(n), np.sin(x.sum(axis=1))


def gp_configs(K):
    """Reduced-size versions of BASELINE.json's configs: name -> (kernel, x, yerr, y)."""
    out = {}
    x, yerr, y, amp = scaling_data(100)
    out["scaling100"] = (amp * K.ExpSquaredKernel(1.0), x, yerr, y)          # scaling.rst:76 golden 133.946394912
    x, yerr, y = bench_data(1024)
    out["C1"] = (np.var(y) * K.ExpSquaredKernel(1.0), x, yerr, y)
    x, yerr, y = bench_data(1500)
    out["C2small"] = (np.var(y) * K.ExpSquaredKernel(1.0), x, yerr, y)
    out["C3small"] = (np.var(y) * K.Matern32Kernel(1.0), x, yerr, y)
    x, yerr, y = bench_data(700, ndim=3)
    out["C5small"] = (K.Matern52Kernel(0.5, ndim=3) + K.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3), x, yerr, y)
    return out


def hodlr_configs(K):
    """HODLR cases: name -> (kernel, x, yerr, y, dict(min_size, tol, seed)).  Shapes follow the
    reference's HODLR tests (tests/test_solvers.py:29-75, tests/test_gp.py HODLR parametrisations,
    docs/tutorials/scaling.rst) and BASELINE config C4 at reduced size."""
    out = {}
    rng = np.random.RandomState(1234)
    x = np.sort(10 * rng.randn(1000))
    out["solver1000"] = (1.0 * K.ExpSquaredKernel(1.0), x, np.ones(1000), np.sin(x), dict(min_size=100, tol=1e-10, seed=42))
    x, yerr, y, amp = scaling_data(2000)
    out["scaling2000_default"] = (amp * K.ExpSquaredKernel(1.0), x, yerr, y, dict(min_size=100, tol=0.1, seed=42))
    for n in (4096, 8192):
        x, yerr, y = bench_data(n)
        out["C4_%d" % n] = (np.var(y) * K.ExpSquaredKernel(1.0), x, yerr, y, dict(min_size=100, tol=1e-10, seed=42))
    # the point DENSITY of the full C4 (262144 points on [0, 10]): the deep blocks span so little of the
    # kernel's length scale that every residual row falls under 1e-14 after three or four terms, and
    # the reference takes its exact "trivial factorisation" there (rank = block size)
    rng = np.random.RandomState(99)
    x = np.sort(rng.uniform(0, 10 * 4096 / 262144.0, 4096))
    out["C4_density_4096"] = (0.5 * K.ExpSquaredKernel(1.0), x, 0.1 * np.ones(4096), np.sin(40 * x), dict(min_size=100, tol=1e-10, seed=42))
    x, yerr, y = bench_data(3000)
    out["C4_3000_tol1e-4_seed7"] = (np.var(y) * K.ExpSquaredKernel(1.0), x, yerr, y, dict(min_size=64, tol=1e-4, seed=7))
    # exactly low-rank block (Matern-3/2 on sorted 1-D inputs is rank 2 off the diagonal): with a tight
    # tolerance every remaining residual row falls under the 1e-14 pivot threshold, the reference runs
    # out of rows and returns the exact block ("trivial factorisation", hodlr.h:160-176)
    x, yerr, y = bench_data(1200)
    out["m32_exhausted"] = (np.var(y) * K.Matern32Kernel(1.0), x, yerr, y, dict(min_size=100, tol=1e-8, seed=3))
    # ranks grow with the dimension (docs/user/solvers.rst:40-42)
    x, yerr, y = bench_data(2000, ndim=3)
    out["c5like3d"] = (K.Matern52Kernel(0.5, ndim=3) + K.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3), x, yerr, y,
                       dict(min_size=100, tol=1e-6, seed=42))
    # rank beyond 256 (400 at the root in the reference build)
    x, yerr, y = bench_data(4096, ndim=3)
    out["c5like3d_4096_rank400"] = (K.Matern52Kernel(0.5, ndim=3) + K.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3), x, yerr, y,
                                    dict(min_size=100, tol=1e-7, seed=42))
    x, yerr, y = bench_data(2000, ndim=2)
    out["expsq2d"] = (0.7 * K.ExpSquaredKernel([0.3, 0.5], ndim=2), x, yerr, y, dict(min_size=100, tol=1e-8, seed=11))
    return out
