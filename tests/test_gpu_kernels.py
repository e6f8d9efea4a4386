"""Parity of the HIP kernel-function evaluator (through the C ABI) with the reference: committed
golden vectors from the real reference, the oracle on larger / ragged / empty inputs, and the
reference's own finite-difference tests (tests/test_kernels.py:65-128)."""
import numpy as np
import pytest

import zoo
from oracle import kernels_np
import george_amd
import george_amd.kernels as AK

pytestmark = pytest.mark.gpu

# fp64; device libm (ocml) vs glibc differ by <= 1-2 ulp in exp/sin/cos/pow, amplified by the
# argument magnitude (|r2| up to ~1e2 in the zoo): stated tolerance
RTOL, ATOL = 2e-12, 1e-14
ZOO = zoo.kernel_zoo(AK)


@pytest.mark.parametrize("name,kernel", ZOO, ids=[n for n, _ in ZOO])
def test_golden_vectors(name, kernel, golden_kernels):
    g = golden_kernels
    t1, t2 = g[name + "/t1"], g[name + "/t2"]
    ki = kernel.kernel
    which = np.ones(kernel.full_size, dtype=np.uint32)
    np.testing.assert_allclose(ki.value_symmetric(t1), g[name + "/vsym"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ki.value_general(t1, t2), g[name + "/vgen"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ki.value_diagonal(t1, t1[::-1].copy()), g[name + "/vdiag"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ki.gradient_general(which, t1, t2), g[name + "/ggen"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ki.gradient_symmetric(which, t1), g[name + "/gsym"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ki.x1_gradient_general(t1, t2), g[name + "/x1"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ki.x2_gradient_general(t1, t2), g[name + "/x2"], rtol=1e-10, atol=1e-12)
    # exact symmetry of the symmetric build (kernel_interface.cpp:68-74 mirrors one evaluation)
    v = ki.value_symmetric(t1)
    assert np.array_equal(v, v.T)


@pytest.mark.parametrize("name,kernel", ZOO[::3], ids=[n for n, _ in ZOO[::3]])
def test_ragged_sizes_vs_oracle(name, kernel):
    rng = np.random.RandomState(99)
    for n1, n2 in [(1, 1), (130, 67), (257, 3), (64, 64)]:
        a, b = rng.randn(n1, kernel.ndim), rng.randn(n2, kernel.ndim)
        np.testing.assert_allclose(kernel.get_value(a, b), kernels_np.value_general(kernel, a, b), rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(kernel.get_value(a), kernels_np.value_symmetric(kernel, a), rtol=RTOL, atol=ATOL)
    # masked gradients: masked slots are dropped exactly as kernels.py:127 does
    if kernel.full_size >= 2:
        kernel.freeze_parameter(kernel.get_parameter_names()[0])
        a = rng.randn(33, kernel.ndim)
        g = kernel.get_gradient(a)
        full = kernels_np.gradient_symmetric(kernel, a)
        np.testing.assert_allclose(g, full[:, :, kernel.unfrozen_mask], rtol=1e-10, atol=1e-12)
        kernel.thaw_all_parameters()


@pytest.mark.parametrize("cls", [AK.ExpSquaredKernel, AK.Matern32Kernel, AK.Matern52Kernel, AK.ExpKernel])
@pytest.mark.parametrize("ndim,metric", [(1, 0.7), (2, 1.9), (3, 0.5), (3, [1.0, 0.1, 10.0]), (2, [0.3, 2.0])])
def test_interior_tile_kernel_vs_oracle(cls, ndim, metric):
    """kmat_interior_kernel: the specialised 64x64 tiles of a + b F(r^2) on <= 3 plain coordinates, which only
    matrices with at least one tile off the edge and off the diagonal reach -- symmetric, general and
    offset builds at ragged sizes against the oracle, and against the generic tile path
    (GEORGE_AMD_NO_KMAT_INTERIOR in a subprocess) to a few ulps."""
    rng = np.random.RandomState(5 + ndim)
    kernel = 0.8 * cls(metric, ndim=ndim) + 0.05
    for n1, n2 in [(333, 200), (128, 192), (700, 65)]:
        a, b = rng.uniform(0, 3, (n1, ndim)), rng.uniform(0, 3, (n2, ndim))
        np.testing.assert_allclose(kernel.get_value(a, b), kernels_np.value_general(kernel, a, b), rtol=RTOL, atol=ATOL)
        v = kernel.get_value(a)
        np.testing.assert_allclose(v, kernels_np.value_symmetric(kernel, a), rtol=RTOL, atol=ATOL)
        assert np.array_equal(v, v.T)
    # the factorisation's own build (lower 128-tiles, yerr on the diagonal, identity padding): log-det vs NumPy
    from george_amd import BasicSolver
    x = np.sort(rng.uniform(0, 3, (900, ndim)), axis=0)
    s = BasicSolver(kernel)
    s.compute(x, 0.3 * np.ones(900))
    Kd = kernels_np.value_symmetric(kernel, x) + 0.09 * np.eye(900)
    assert abs(s.log_determinant - np.linalg.slogdet(Kd)[1]) <= 1e-9 * 900


def test_interior_tile_kernel_matches_generic_path():
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import numpy as np\n"
            "import george_amd.kernels as K\n"
            "rng = np.random.RandomState(3); x = rng.uniform(0, 3, (500, 3))\n"
            "k = 0.8 * K.Matern52Kernel([1.0, 0.1, 10.0], ndim=3) + 0.05\n"
            "np.save(sys.argv[1], k.get_value(x))\n") % root
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for i, env in enumerate(({}, {"GEORGE_AMD_NO_KMAT_INTERIOR": "1"})):
            e = dict(os.environ); e.update(env)
            f = os.path.join(d, "v%d.npy" % i)
            r = subprocess.run([sys.executable, "-c", code, f], env=e, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(np.load(f))
    assert not np.array_equal(outs[0], np.zeros_like(outs[0]))
    np.testing.assert_allclose(outs[0], outs[1], rtol=4e-16, atol=1e-17)


def test_empty_and_bad_inputs():
    k = AK.ExpSquaredKernel(1.0, ndim=2)
    assert k.get_value(np.zeros((0, 2)), np.zeros((5, 2))).shape == (0, 5)
    assert k.get_value(np.zeros((0, 2))).shape == (0, 0)
    with pytest.raises(RuntimeError):
        k.kernel.value_symmetric(np.zeros((4, 3)))            # "dimension mismatch", kernel_interface.cpp:65


def test_large_tile_coverage_vs_oracle():
    """A size that is not a tile multiple and spans many workgroups (1-D and 3-D)."""
    rng = np.random.RandomState(5)
    k1 = 0.7 * AK.ExpSquaredKernel(1.3)
    x = np.sort(rng.uniform(0, 10, 1500))[:, None]
    np.testing.assert_allclose(k1.get_value(x), kernels_np.value_symmetric(k1, x), rtol=RTOL, atol=ATOL)
    k3 = AK.Matern52Kernel(0.5, ndim=3) + AK.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    x3 = rng.uniform(0, 1, (700, 3))
    t3 = rng.uniform(0, 1, (333, 3))
    np.testing.assert_allclose(k3.get_value(t3, x3), kernels_np.value_general(k3, t3, x3), rtol=RTOL, atol=ATOL)


STATIONARY = [
    (AK.ExpKernel, {}), (AK.ExpSquaredKernel, {}), (AK.Matern32Kernel, {}), (AK.Matern52Kernel, {}),
    (AK.RationalQuadraticKernel, dict(log_alpha=np.log(1.0))),
    (AK.RationalQuadraticKernel, dict(log_alpha=np.log(0.1))),
    (AK.RationalQuadraticKernel, dict(log_alpha=np.log(10.0))),
]


def _fd_checks(kernel, N=20, seed=123, eps=1.32e-6):
    np.random.seed(seed)
    t1 = np.random.randn(N, kernel.ndim)
    kernel.test_gradient(t1, eps=eps)
    kernel.test_gradient(t1, t1[:1], eps=eps)
    kernel.test_x1_gradient(t1, eps=eps)
    kernel.test_x1_gradient(t1, np.array(t1[:1]), eps=eps)
    kernel.test_x2_gradient(t1, eps=eps)
    kernel.test_x2_gradient(np.array(t1[:1]), t1, eps=eps)


@pytest.mark.parametrize("cls,kw", STATIONARY)
def test_stationary_finite_differences(cls, kw):
    """tests/test_kernels.py:83-128"""
    for metric, more in [(0.1, {}), (1.0, {}), (10.0, {}), ([1.0, 0.1, 10.0], dict(ndim=3)), (1.0, dict(ndim=3)),
                         (1.0, dict(ndim=3, axes=2)), (1.0, dict(ndim=3, axes=2, block=(-0.1, 0.1)))]:
        _fd_checks(cls(metric=metric, **dict(kw, **more)))


@pytest.mark.parametrize("name,kernel", ZOO[:21], ids=[n for n, _ in ZOO[:21]])
def test_kernel_finite_differences(name, kernel):
    """tests/test_kernels.py:65-80"""
    _fd_checks(kernel)


def test_general_metric_closed_form():
    """tests/test_metrics.py:40-88"""
    rng = np.random.RandomState(1234)
    ndim = 3
    L = rng.randn(ndim, ndim)
    L[np.diag_indices(ndim)] = np.exp(L[np.diag_indices(ndim)])
    L[np.triu_indices(ndim, 1)] = 0.0
    for metric in (np.eye(ndim), L @ L.T):
        kernel = 0.1 * AK.ExpSquaredKernel(metric, ndim=ndim)
        x = rng.rand(50, ndim)
        M0 = kernel.get_value(x)
        d = x[:, None, :] - x[None, :, :]
        r2 = np.einsum("ijk,kl,ijl->ij", d, np.linalg.inv(metric), d)
        assert np.allclose(M0, 0.1 * np.exp(-0.5 * r2))
        assert np.allclose(george_amd.GP(kernel).get_matrix(x), M0)


FAST_FORMS = [
    ("scaled_expsq", lambda: 10.0 * AK.ExpSquaredKernel(1.3)),
    ("bare_matern32", lambda: AK.Matern32Kernel(0.7)),
    ("offset_matern52", lambda: 2.0 * AK.Matern52Kernel(0.4) + 0.3),
    ("ratquad_axis", lambda: 0.5 * AK.RationalQuadraticKernel(log_alpha=0.1, metric=[0.5, 2.0], ndim=2)),
    ("exp_subspace", lambda: 1.7 * AK.ExpKernel(2.0, ndim=3, axes=[0, 2])),
    ("nested", lambda: 3.0 * (2.0 * AK.ExpSquaredKernel(0.9) + 1.0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", FAST_FORMS, ids=[f[0] for f in FAST_FORMS])
def test_fast_affine_form_matches_interpreter(name, make, monkeypatch):
    """Kernels of the shape a + b*F(r^2) take the interpreter-free evaluator (gh_eval.h GhFast);
    it must agree with the postfix interpreter to rounding (<= 4 ulp of the largest entry)."""
    from george_amd.kernel_interface import KernelInterface
    kernel = make()
    rng = np.random.default_rng(5)
    x = rng.uniform(-3, 3, size=(301, kernel.ndim))
    x2 = rng.uniform(-3, 3, size=(77, kernel.ndim))
    fast = KernelInterface(kernel)
    monkeypatch.setenv("GEORGE_AMD_NO_FAST_KERNEL", "1")
    slow = KernelInterface(kernel)
    monkeypatch.delenv("GEORGE_AMD_NO_FAST_KERNEL")
    for a, b in ((fast.value_symmetric(x), slow.value_symmetric(x)),
                 (fast.value_general(x, x2), slow.value_general(x, x2))):
        assert np.max(np.abs(a - b)) <= 1e-15 * np.max(np.abs(b))
    ks = fast.value_symmetric(x)
    assert np.array_equal(ks, ks.T)
    assert np.allclose(ks, kernels_np.value_symmetric(kernel, x), rtol=1e-13, atol=1e-14)


def _hyper_kernel():
    """docs/tutorials/hyper.rst:91-95: k1 + k2 * ExpSine2 + k3 + k4, eleven parameters (the survey's f1 workload)"""
    kernels = AK
    k1 = 66.0 ** 2 * kernels.ExpSquaredKernel(metric=67.0 ** 2)
    k2 = 2.4 ** 2 * kernels.ExpSquaredKernel(90.0 ** 2) * kernels.ExpSine2Kernel(gamma=2.0 / 1.3 ** 2, log_period=0.0)
    k3 = 0.66 ** 2 * kernels.RationalQuadraticKernel(log_alpha=np.log(0.78), metric=1.2 ** 2)
    k4 = 0.18 ** 2 * kernels.ExpSquaredKernel(1.6 ** 2)
    return k1 + k2 + k3 + k4


@pytest.mark.gpu
def test_hyper_tutorial_kernel_against_the_oracle():
    """The survey's f1 workload (docs/tutorials/hyper.rst:91-104): the 11-parameter kernel through the postfix walker -- kernel
    matrix against the oracle (the reference's C++ evaluator where oracle/_ref travelled) at 2e-12 of the largest entry, and the
    fused gradient reduction 1/2 sum A_ij dK_ij/dtheta against the reference formula (gp.py:465-466) on the oracle's tensors."""
    from george_amd.kernel_interface import KernelInterface
    from george_amd import BasicSolver
    from oracle import solver_np
    kernel = _hyper_kernel()
    rng = np.random.default_rng(3)
    n = 400
    x = np.sort(1958.0 + 52.0 * rng.uniform(0, 1, n))[:, None]
    ks = KernelInterface(kernel).value_symmetric(x)
    ref = kernels_np.value_symmetric(kernel, x)
    assert np.max(np.abs(ks - ref)) <= 2e-12 * np.max(np.abs(ref))
    assert np.array_equal(ks, ks.T)
    y = 340.0 + 3.0 * np.sin(2 * np.pi * x[:, 0]) - 345.0
    yerr = 0.19 * np.ones(n)
    s = BasicSolver(kernel)
    s.compute(x, yerr)
    which = np.ones(len(kernel.get_parameter_vector(include_frozen=True)), dtype=np.uint32)
    g, alpha, diagA = s.grad(y, which)
    d = solver_np.DenseOracle(kernel)
    d.compute(x, yerr)
    gref, A = solver_np.gp_grad_log_likelihood(d, kernel, x, y)
    assert np.max(np.abs(g - gref) / np.maximum(np.abs(gref), 1e-6 * np.abs(gref).max())) <= 1e-7
    assert np.allclose(diagA, np.diag(A), rtol=1e-7, atol=1e-9 * np.abs(np.diag(A)).max())
