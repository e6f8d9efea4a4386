"""Parity at the sizes the claims are made on (``-m gpu``).

``tests/golden/large.json`` holds log-determinant, log-likelihood, a strided sample of
``alpha = K^-1 y`` (and, for C5, predictive mean / variance and the gradient) computed by the REAL
reference in the build container -- its compiled C++ kernel evaluator + the SciPy LAPACK calls of
``basic.py:68,87`` (``oracle/gen_golden_large.py``).  The HIP path is compared with those numbers
directly: north-star bound 1e-6 relative on the log-likelihood (``BASELINE.json``), asserted here
at 1e-9; the log-determinant on its own at 1e-10 relative (a skipped or duplicated diagonal block
would shift it by O(1e2) out of O(1e5))."""
import json
import os

import numpy as np
import pytest

import zoo
from george_amd import kernels, GP, HODLRSolver

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "large.json")) as _f:
    LARGE = json.load(_f)

CASES = {"C2": (16384, kernels.ExpSquaredKernel), "M32_20k": (20480, kernels.Matern32Kernel),
         "NS": (65536, kernels.ExpSquaredKernel), "C3": (65536, kernels.Matern32Kernel)}


@pytest.mark.parametrize("name", list(CASES))
def test_dense_1d_against_reference_scalars(name):
    if name not in LARGE:
        pytest.skip("no reference scalars committed for %s" % name)
    g = LARGE[name]
    n, cls = CASES[name]
    assert g["n"] == n
    x, yerr, y = zoo.bench_data(n)
    gp = GP(np.var(y) * cls(1.0))
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)
    assert abs(gp.solver.log_determinant - g["logdet"]) <= 1e-10 * abs(g["logdet"]), (gp.solver.log_determinant, g["logdet"])
    assert abs(ll - g["loglike"]) <= 1e-9 * abs(g["loglike"]), (ll, g["loglike"])
    assert abs(gp.solver.dot_solve(y) - g["quad"]) <= 1e-7 * abs(g["quad"])
    alpha = gp.apply_inverse(y)[::g["alpha_stride"]]
    ref = np.array(g["alpha"])
    # alpha = K^-1 y amplifies rounding by cond(K) ~ 1e6: compare on the scale of the vector
    assert np.abs(alpha - ref).max() <= 1e-6 * np.abs(ref).max()


def test_c5_full_size_against_reference():
    if "C5" not in LARGE:
        pytest.skip("no reference scalars committed for C5")
    g = LARGE["C5"]
    x, yerr, y = zoo.bench_data(32768, ndim=3)
    kernel = kernels.Matern52Kernel(0.5, ndim=3) + kernels.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    gp = GP(kernel)
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)
    assert abs(gp.solver.log_determinant - g["logdet"]) <= 1e-10 * abs(g["logdet"])
    assert abs(ll - g["loglike"]) <= 1e-9 * abs(g["loglike"]), (ll, g["loglike"])
    mu, var = gp.predict(y, np.array(g["t"]), return_var=True)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, g["var"], rtol=1e-6, atol=1e-10)
    grad = gp.grad_log_likelihood(y)
    assert list(gp.get_parameter_names()) == ["kernel:" + s for s in g["grad_names"]]
    np.testing.assert_allclose(grad, g["grad"], rtol=1e-7, atol=1e-6)


@pytest.mark.parametrize("name", ["NS", "C3"])
def test_dense_vs_hodlr_logdet_at_65536(name):
    """Independent on-device cross-check of the dense log-determinant at N = 65536: the HODLR solver
    shares no factorisation code with the dense path (ACA + Woodbury cores + 128-row leaves)."""
    n, cls = CASES[name]
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * cls(1.0)
    gd = GP(kernel)
    gd.compute(x, yerr)
    gh = GP(kernel, solver=HODLRSolver, tol=1e-12)
    gh.compute(x, yerr)
    assert abs(gd.solver.log_determinant - gh.solver.log_determinant) <= 1e-8 * abs(gd.solver.log_determinant)
    assert abs(gd.log_likelihood(y) - gh.log_likelihood(y)) <= 1e-7 * abs(gd.log_likelihood(y))


def test_c4_full_size_against_reference_hodlr():
    """BASELINE configs[3] at its stated size: N = 262144, HODLRSolver(tol=1e-10, min_size=100, seed=42)
    against the reference's own hodlr.h (oracle/_ref/_hodlr: hodlr.h:75-103 compute, :237-254
    apply_inverse; 59 s on one core in the build container, oracle/gen_golden_large.py C4).  Both are
    tol = 1e-10 approximations of the same dense answer from different pivot rows (per-node RNG
    streams here, one mt19937 threaded through the construction there), so they agree to a small
    multiple of the tolerance, far inside the north-star bound of 1e-6 on the log-likelihood."""
    if "C4" not in LARGE:
        pytest.skip("no reference scalars committed for C4")
    g = LARGE["C4"]
    n = g["n"]
    assert n == 262144 and g["tol"] == 1e-10 and g["min_size"] == 100 and g["seed"] == 42
    x, yerr, y = zoo.bench_data(n)
    gp = GP(np.var(y) * kernels.ExpSquaredKernel(1.0), solver=HODLRSolver, tol=g["tol"], min_size=g["min_size"], seed=g["seed"])
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)
    assert abs(gp.solver.log_determinant - g["logdet"]) <= 1e-9 * abs(g["logdet"]), (gp.solver.log_determinant, g["logdet"])
    assert abs(ll - g["loglike"]) <= 1e-9 * abs(g["loglike"]), (ll, g["loglike"])
    assert abs(gp.solver.dot_solve(y) - g["quad"]) <= 1e-7 * abs(g["quad"])
    alpha = gp.apply_inverse(y)[::g["alpha_stride"]]
    ref = np.array(g["alpha"])
    # alpha amplifies the two solvers' O(tol) differences by cond(K) ~ 1e6 (measured: 1.5e-6 of the vector's scale)
    assert np.abs(alpha - ref).max() <= 2e-5 * np.abs(ref).max()
    # rank profile: per level within a few of the reference's, except where the reference ran out of
    # rows and took its rank-min(rows, cols) fallback (hodlr.h:160-176: levels 8 and 9 here, 512 / 256)
    mine = gp.solver.ranks()
    at = 0
    for lvl, r_ref in enumerate(g["rank_per_level"]):
        r = max(mine[at:at + (1 << lvl)])
        at += 1 << lvl
        size = n >> lvl
        if r_ref >= size // 2 - 1:
            assert r <= r_ref
        else:
            assert abs(r - r_ref) <= max(4, r_ref // 4), (lvl, r, r_ref)


@pytest.mark.parametrize("ndev", [2, 8])
def test_c4_full_size_split_over_devices(ndev):
    """The same C4 with the tree split over `ndev` sub-trees (gh_hodlr_mgpu_*; the one GPU of the
    box listed ndev times): against the reference scalars, and node for node the ranks of the single-GPU solver."""
    if "C4" not in LARGE:
        pytest.skip("no reference scalars committed for C4")
    from george_amd import MultiGPUHODLRSolver
    g = LARGE["C4"]
    n = g["n"]
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    kw = dict(tol=g["tol"], min_size=g["min_size"], seed=g["seed"])
    gp = GP(kernel, solver=MultiGPUHODLRSolver, devices=[0] * ndev, **kw)
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)
    assert abs(gp.solver.log_determinant - g["logdet"]) <= 1e-9 * abs(g["logdet"]), (gp.solver.log_determinant, g["logdet"])
    assert abs(ll - g["loglike"]) <= 1e-9 * abs(g["loglike"]), (ll, g["loglike"])
    alpha = gp.apply_inverse(y)[::g["alpha_stride"]]
    ref = np.array(g["alpha"])
    assert np.abs(alpha - ref).max() <= 2e-5 * np.abs(ref).max()
    one = HODLRSolver(kernel, **kw)
    one.compute(x[:, None], yerr)
    assert gp.solver.ranks() == one.ranks()
    # (same pivots, but the clusters of the ACA kernel differ in size between the two forms: other summation orders,
    #  amplified by cond(K) ~ 1e6 -- measured 3e-11)
    assert abs(gp.solver.log_determinant - one.log_determinant) <= 1e-9 * abs(one.log_determinant)
    assert [r[1] for r in gp.solver.rows()] == [n // ndev] * ndev


def test_fused_objective_against_reference_c5():
    """gp.py:470-480 at the full C5 size: ONE fused device call (gh_chol_objective) against the
    reference's log-likelihood and gradient directly (not via the separate HIP calls)."""
    if "C5" not in LARGE:
        pytest.skip("no reference scalars committed for C5")
    g = LARGE["C5"]
    x, yerr, y = zoo.bench_data(32768, ndim=3)
    kernel = kernels.Matern52Kernel(0.5, ndim=3) + kernels.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    gp = GP(kernel)
    gp.compute(x, yerr)
    gp.kernel.dirty = True                            # as after a parameter change: the fused call re-factorises
    v, grad = gp.nll_and_grad(gp.get_parameter_vector(), y)
    assert abs(-v - g["loglike"]) <= 1e-9 * abs(g["loglike"]), (v, g["loglike"])
    np.testing.assert_allclose(-grad, g["grad"], rtol=1e-7, atol=1e-6)
