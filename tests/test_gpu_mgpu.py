"""The sharded dense solver behind the C ABI (gh_mgpu_*, george_amd/csrc/gh_mgpu.hip) on the one GPU of
the test box:

* transport "rccl" with n_dev = 1 -- communicator creation (ncclCommInitAll), the all-reduce self-check
  and the whole driver as a world of one;
* transport "copy" with the SAME device listed 2, 4 and 8 times ("virtual devices") -- the ownership,
  ordering and hand-over logic of a 1x2, 2x2 and 2x4 grid with real tile kernels, against the
  single-GPU solver (itself pinned to the reference, tests/test_gpu_fullsize.py).

What this cannot cover is RCCL between two physical devices; see DESIGN.md section 7."""
import numpy as np
import pytest

import zoo
from george_amd import kernels, GP, BasicSolver, MultiGPUSolver

pytestmark = pytest.mark.gpu


def _case(n, ndim=1):
    x, yerr, y = zoo.bench_data(n, ndim=ndim)
    if ndim == 1:
        kernel = np.var(y) * kernels.Matern32Kernel(1.0)
    else:
        kernel = kernels.Matern52Kernel(0.5, ndim=3) + kernels.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    X = np.ascontiguousarray(x.reshape(n, -1))
    d = BasicSolver(kernel)
    d.compute(X, yerr)
    return kernel, X, yerr, y, d


@pytest.mark.parametrize("devices,transport,n,nb,grid", [
    ([0], "rccl", 2500, 512, None),
    ([0], "copy", 1000, 128, None),
    ([0, 0], "copy", 2500, 256, None),
    ([0, 0, 0, 0], "copy", 3000, 256, None),
    ([0, 0, 0, 0], "copy", 1700, 128, (4, 1)),
    ([0, 0, 0, 0], "copy", 1700, 128, (1, 4)),
    ([0] * 8, "copy", 4100, 256, None),
    ([0] * 6, "copy", 2900, 128, (2, 3)),
])
def test_sharded_solver_matches_single_gpu(devices, transport, n, nb, grid):
    kernel, X, yerr, y, d = _case(n)
    s = MultiGPUSolver(kernel, devices=devices, nb=nb, grid=grid, transport=transport)
    s.compute(X, yerr)
    pr, pc, nb_used = s.grid_shape()
    assert pr * pc == len(devices) and nb_used == nb
    if grid is None:
        assert (pr, pc) == {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}[len(devices)]
    assert s.computed
    assert abs(s.log_determinant - d.log_determinant) <= 1e-11 * abs(d.log_determinant)
    q = d.dot_solve(y)
    assert abs(s.dot_solve(y) - q) <= 1e-10 * abs(q)
    a = d.apply_inverse(y)
    np.testing.assert_allclose(s.apply_inverse(y), a, rtol=0, atol=1e-9 * np.abs(a).max())
    Y2 = np.stack([y, np.cos(3 * X[:, 0])], axis=1)
    a2 = d.apply_inverse(Y2)
    got = s.apply_inverse(Y2)
    assert got.shape == (n, 2)
    np.testing.assert_allclose(got, a2, rtol=0, atol=1e-9 * np.abs(a2).max())
    # a second compute on the same handle (an optimiser iterate), other size: buffers regrow
    s.compute(X[: n // 2], yerr[: n // 2])
    d2 = BasicSolver(kernel)
    d2.compute(X[: n // 2], yerr[: n // 2])
    assert abs(s.log_determinant - d2.log_determinant) <= 1e-11 * abs(d2.log_determinant)


def test_sharded_solver_through_gp_and_3d():
    kernel, X, yerr, y, d = _case(2000, ndim=3)
    gp = GP(kernel, solver=MultiGPUSolver, devices=[0, 0, 0, 0], transport="copy", nb=256)
    gp.compute(X, yerr)
    ref = GP(kernel)
    ref.compute(X, yerr)
    ll, ll0 = gp.log_likelihood(y), ref.log_likelihood(y)
    assert abs(ll - ll0) <= 1e-10 * abs(ll0)
    t = X[:37] + 0.01
    mu, var = gp.predict(y, t, return_var=True)            # the generic NumPy path of GP.predict on apply_inverse
    mu0, var0 = ref.predict(y, t, return_var=True)
    np.testing.assert_allclose(mu, mu0, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(var, var0, rtol=1e-6, atol=1e-9)
    with pytest.raises(NotImplementedError):
        gp.solver.apply_sqrt(y)
    inv = MultiGPUSolver(kernel, devices=[0, 0], transport="copy", nb=128)
    inv.compute(X[:300], yerr[:300])
    dd = BasicSolver(kernel)
    dd.compute(X[:300], yerr[:300])
    np.testing.assert_allclose(inv.get_inverse(), dd.get_inverse(), rtol=1e-7, atol=1e-8)


def test_sharded_solver_errors():
    x = np.linspace(0, 3, 700)
    bad = kernels.CosineKernel(log_period=0.0)                              # singular without noise
    for devices in ([0], [0, 0, 0, 0]):
        s = MultiGPUSolver(bad, devices=devices, transport="copy", nb=128)
        with pytest.raises(np.linalg.LinAlgError):
            s.compute(x[:, None], np.zeros(700))
        assert not s.computed
        with pytest.raises(RuntimeError):
            s.dot_solve(np.sin(x))
        # the handle is usable afterwards
        s.kernel = 1.0 * kernels.ExpSquaredKernel(1.0)
        s.compute(x[:, None], 0.1 * np.ones(700))
        assert s.computed and np.isfinite(s.log_determinant)
    s = MultiGPUSolver(1.0 * kernels.ExpSquaredKernel(1.0, ndim=2), devices=[0, 0], transport="copy")
    with pytest.raises(RuntimeError):
        s.compute(x[:, None], 0.1)                                           # dimension mismatch
    with pytest.raises(ValueError):
        MultiGPUSolver(bad, devices=[0, 0], transport="rccl").compute(x[:, None], 0.1)      # RCCL: one rank per device
    with pytest.raises(ValueError):
        MultiGPUSolver(bad, devices=[0, 0, 0], grid=(2, 2), transport="copy").compute(x[:, None], 0.1)
    with pytest.raises(ValueError):
        MultiGPUSolver(bad, devices=[97], transport="copy").compute(x[:, None], 0.1)
