"""Pin the oracle (CPU, no GPU): NumPy restatement vs the committed golden vectors produced by
the real reference, vs the reference's published golden scalars, and -- where the compiled
reference evaluator (oracle/_ref) or /root/reference is present -- vs the reference itself."""
import numpy as np
import pytest

import zoo
from oracle import kernels_np, solver_np, hodlr_np, ref_loader
import george_amd.kernels as AK

RTOL, ATOL = 1e-11, 1e-13      # libm-vs-NumPy rounding on exp/sin/pow of O(1..100) arguments

ZOO = zoo.kernel_zoo(AK)


@pytest.mark.parametrize("name,kernel", ZOO, ids=[n for n, _ in ZOO])
def test_numpy_restatement_matches_reference_goldens(name, kernel, golden_kernels):
    g = golden_kernels
    t1, t2 = g[name + "/t1"], g[name + "/t2"]
    assert list(g[name + "/names"]) == list(kernel.get_parameter_names(include_frozen=True))
    np.testing.assert_allclose(g[name + "/vector"], kernel.get_parameter_vector(include_frozen=True), rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(kernels_np.value_symmetric(kernel, t1), g[name + "/vsym"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(kernels_np.value_general(kernel, t1, t2), g[name + "/vgen"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(kernels_np.value_diagonal(kernel, t1, t1[::-1].copy()), g[name + "/vdiag"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(kernels_np.gradient_general(kernel, t1, t2), g[name + "/ggen"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(kernels_np.gradient_symmetric(kernel, t1), g[name + "/gsym"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(kernels_np.x1_gradient_general(kernel, t1, t2), g[name + "/x1"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(kernels_np.x2_gradient_general(kernel, t1, t2), g[name + "/x2"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name,kernel", ZOO[::5], ids=[n for n, _ in ZOO[::5]])
def test_compiled_reference_evaluator_accepts_our_specs(name, kernel, golden_kernels):
    """oracle/_ref/kernel_interface*.so is the reference's own C++; it must evaluate OUR spec
    objects (same attribute protocol) to the goldens bit for bit."""
    KI = ref_loader.load_kernel_interface()
    if KI is None:
        pytest.skip("oracle/_ref not built")
    g = golden_kernels
    ki = KI(kernel)
    assert np.array_equal(ki.value_symmetric(g[name + "/t1"]), g[name + "/vsym"])
    assert np.array_equal(ki.value_general(g[name + "/t1"], g[name + "/t2"]), g[name + "/vgen"])


def test_published_golden_scalars(golden_gp):
    # docs/tutorials/scaling.rst:76,91 -> 133.946394912 (Basic and HODLR)
    assert abs(float(golden_gp["scaling100/loglike"]) - 133.946394912) < 5e-9
    kernel, x, yerr, y = zoo.gp_configs(AK)["scaling100"]
    ll = solver_np.gp_log_likelihood(solver_np.DenseOracle(kernel, force_port=True), x[:, None], yerr, y)
    assert abs(ll - 133.946394912) < 5e-9
    # N=100 < 2*min_size: the HODLR tree is a single leaf -> exact (scaling.rst:91)
    ll_h = solver_np.gp_log_likelihood(hodlr_np.HODLROracle(kernel, force_port=True), x[:, None], yerr, y)
    assert abs(ll_h - 133.946394912) < 5e-9


@pytest.mark.parametrize("name", ["C1", "C3small", "C5small"])
def test_dense_oracle_matches_reference_gp(name, golden_gp):
    g = golden_gp
    kernel, x, yerr, y = zoo.gp_configs(AK)[name]
    X = x[:, None] if x.ndim == 1 else x
    s = solver_np.DenseOracle(kernel, force_port=True)
    ll = solver_np.gp_log_likelihood(s, X, yerr, y)
    assert abs(ll - float(g[name + "/loglike"])) <= 1e-9 * abs(float(g[name + "/loglike"]))
    t = g[name + "/t"]
    T = t[:, None] if t.ndim == 1 else t
    mu, var = solver_np.gp_predict(s, kernel, X, y, T, force_port=True)
    np.testing.assert_allclose(mu, g[name + "/mu"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, g[name + "/var"], rtol=1e-6, atol=1e-9)
    grad, _ = solver_np.gp_grad_log_likelihood(s, kernel, X, y, force_port=True)
    np.testing.assert_allclose(grad, g[name + "/grad"], rtol=1e-6, atol=1e-6)


def test_hodlr_oracle_against_dense():
    """The criterion of the reference's own HODLR tests (tests/test_solvers.py:29-62)."""
    kernel = 1.0 * AK.ExpSquaredKernel(1.0)
    rng = np.random.RandomState(1234)
    N = 300
    x = np.sort(10 * rng.randn(N))[:, None]
    yerr = np.ones(N)
    h = hodlr_np.HODLROracle(kernel, tol=1e-10, force_port=True)
    h.compute(x, yerr)
    d = solver_np.DenseOracle(kernel, force_port=True)
    d.compute(x, yerr)
    assert np.allclose(h.log_determinant, d.log_determinant)
    y = np.sin(x[:, 0])
    assert np.allclose(h.apply_inverse(y), d.apply_inverse(y))
    K = solver_np.kernel_matrix(kernel, x, force_port=True)
    K[np.diag_indices_from(K)] += yerr ** 2
    assert np.allclose(h.apply_inverse(K), np.eye(N))


def test_mt19937_matches_std():
    r = hodlr_np.MT19937(42)
    assert [r(), r()] == [1608637542, 3421126067]     # std::mt19937 seeded with 42


def test_flattener_accepts_reference_kernel_objects():
    """Drop-in claim of INTEGRATION.md section 1: our spec flattener reads the reference's own
    `george.kernels` objects (same attribute protocol as parser.h) into the same POD program."""
    george = ref_loader.load_reference()
    if george is None:
        pytest.skip("/root/reference not present (GPU box)")
    from george_amd import program
    ours = zoo.kernel_zoo(AK)
    theirs = zoo.kernel_zoo(george.kernels)
    for (name, ka), (_, kr) in zip(ours, theirs):
        assert bytes(program.flatten(ka)) == bytes(program.flatten(kr)), name
        dk = program.DeviceKernel(kr)
        assert (dk.ndim, dk.size) == (kr.ndim, kr.full_size)
    # and the reference GP accepts our solver classes as its `solver=` plug-in (construction only here)
    import george_amd
    gp = george.GP(1.0 * george.kernels.ExpSquaredKernel(1.0), solver=george_amd.BasicSolver)
    assert gp.solver_type is george_amd.BasicSolver


def test_against_live_reference_when_available():
    george = ref_loader.load_reference()
    if george is None:
        pytest.skip("/root/reference not present (GPU box)")
    rng = np.random.RandomState(5)
    for name, k_ref in zoo.kernel_zoo(george.kernels)[::7]:
        t1 = rng.randn(11, k_ref.ndim)
        np.testing.assert_allclose(kernels_np.value_symmetric(k_ref, t1), k_ref.get_value(t1), rtol=RTOL, atol=ATOL)
