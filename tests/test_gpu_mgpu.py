"""The sharded dense solver behind the C ABI (gh_mgpu_*, george_amd/csrc/gh_mgpu.hip) on the one GPU of
the test box:

* transport "rccl" with n_dev = 1 -- communicator creation (ncclCommInitAll), the all-reduce self-check
  and the whole driver as a world of one;
* transport "copy" with the SAME device listed 2, 4, 6 and 8 times ("virtual devices") -- the ownership,
  ordering and hand-over logic of the default P x 1 snake grid and of 1x2 / 2x2 / 2x3 / 2x4 / 4x1 / 1x4 grids
  with real tile kernels, against the single-GPU solver (itself pinned to the reference,
  tests/test_gpu_fullsize.py): factorisation, the sweeps with one and many right-hand sides, apply_sqrt,
  get_inverse, predict; the chain-only and trace modes behind profiles/r04/scale_model.md.

With GEORGE_AMD_TEST_VIRTUAL_TRANSPORT=rccl and GEORGE_AMD_RCCL_LIB=tests/mock_rccl/libmock_rccl.so (what
tests/test_gpu_mgpu_mock_rccl.py does in a child process) every virtual-device case below runs through the RCCL
branch of the driver instead -- ncclSend / ncclRecv / groups -- against a stand-in library that pairs every send
with its receive and fails on a count, type, peer or ordering mismatch or an operation nobody answers.

What this cannot cover is RCCL's own transport between two physical devices; see DESIGN.md section 7."""
import ctypes
import os

import numpy as np
import pytest

import zoo
from george_amd import kernels, GP, BasicSolver, MultiGPUSolver

pytestmark = pytest.mark.gpu

VIRTUAL = os.environ.get("GEORGE_AMD_TEST_VIRTUAL_TRANSPORT", "copy")        # transport of the virtual-device cases
MOCK = os.environ.get("GEORGE_AMD_RCCL_LIB") if VIRTUAL == "rccl" else None
ONE_COMM = bool(int(os.environ.get("GEORGE_AMD_TEST_ONE_COMM", "0")))


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
    ([0], VIRTUAL, 1000, 128, None),
    ([0, 0], VIRTUAL, 2500, 256, None),
    ([0, 0, 0, 0], VIRTUAL, 3000, 256, None),
    ([0, 0, 0, 0], VIRTUAL, 1700, 128, (2, 2)),
    ([0, 0, 0, 0], VIRTUAL, 1700, 128, (1, 4)),
    ([0] * 8, VIRTUAL, 4100, 256, None),
    ([0] * 8, VIRTUAL, 4100, 256, (2, 4)),
    ([0] * 8, VIRTUAL, 2600, 128, (4, 2)),
    ([0] * 6, VIRTUAL, 2900, 128, (2, 3)),
    ([0] * 3, VIRTUAL, 2000, 128, None),
])
def test_sharded_solver_matches_single_gpu(devices, transport, n, nb, grid):
    kernel, X, yerr, y, d = _case(n)
    s = MultiGPUSolver(kernel, devices=devices, nb=nb, grid=grid, transport=transport, one_comm=ONE_COMM)
    s.compute(X, yerr)
    mode = s.comm_mode()
    assert (mode == 0) if transport == "copy" else (mode == 1 if ONE_COMM else mode in (1, 2))
    pr, pc, nb_used = s.grid_shape()
    assert pr * pc == len(devices) and nb_used == nb
    if grid is None:
        assert (pr, pc) == (len(devices), 1)                     # whole tile rows per rank ...
        W = len(devices)
        snake = [t if t < W else 2 * W - 1 - t for t in range(2 * W)]
        assert [s.owner(i, 0) for i in range(4 * W)] == [snake[i % (2 * W)] for i in range(4 * W)]      # ... in snake order
    else:
        assert [s.owner(i, j) for i in range(3) for j in range(3)] == [(i % pr) * pc + j % pc for i in range(3) for j in range(3)]
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
    gp = GP(kernel, solver=MultiGPUSolver, devices=[0, 0, 0, 0], transport=VIRTUAL, nb=256)
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
    mu, cov = gp.predict(y, t)
    mu0, cov0 = ref.predict(y, t)
    np.testing.assert_allclose(cov, cov0, rtol=1e-6, atol=1e-9)
    assert type(gp.solver) is MultiGPUSolver


@pytest.mark.parametrize("devices,grid,nb,n", [([0, 0, 0, 0], None, 128, 1100), ([0] * 6, (2, 3), 128, 1500), ([0] * 8, None, 256, 4000),
                                               ([0, 0], (1, 2), 256, 900), ([0], None, 256, 700)])
def test_whole_protocol_on_the_sharded_factor(devices, grid, nb, n):
    """basic.py:72-121 and gp.py:482-545 as tile sweeps: many right-hand sides at once, apply_sqrt, get_inverse, predict"""
    kernel, X, yerr, y, d = _case(n)
    s = MultiGPUSolver(kernel, devices=devices, transport=VIRTUAL, nb=nb, grid=grid)
    s.compute(X, yerr)
    rng = np.random.RandomState(5)
    B = rng.randn(n, 131)                                        # not a multiple of anything
    a = d.apply_inverse(B)
    np.testing.assert_allclose(s.apply_inverse(B), a, rtol=0, atol=1e-9 * np.abs(a).max())
    B2 = B.copy()
    assert s.apply_inverse(B2, in_place=True) is B2
    np.testing.assert_allclose(B2, a, rtol=0, atol=1e-9 * np.abs(a).max())
    R = rng.randn(5, n)
    u = d.apply_sqrt(R)
    np.testing.assert_allclose(s.apply_sqrt(R), u, rtol=0, atol=1e-11 * np.abs(u).max())
    np.testing.assert_allclose(s.apply_sqrt(R[0]), u[0], rtol=0, atol=1e-11 * np.abs(u).max())
    Ki = d.get_inverse()
    np.testing.assert_allclose(s.get_inverse(), Ki, rtol=0, atol=1e-8 * np.abs(Ki).max())
    t = np.ascontiguousarray(X[::7] + 0.013)
    r = y - 0.1
    mu0, var0, _ = d.predict(kernel, r, t, return_var=True)
    _, _, cov0 = d.predict(kernel, r, t, return_cov=True)
    mu, var, _ = s.predict(kernel, r, t, return_var=True)
    np.testing.assert_allclose(mu, mu0, rtol=0, atol=1e-9 * np.abs(mu0).max())
    np.testing.assert_allclose(var, var0, rtol=0, atol=1e-9)
    mu2, _, cov = s.predict(kernel, r, t, return_cov=True)
    np.testing.assert_allclose(mu2, mu0, rtol=0, atol=1e-9 * np.abs(mu0).max())
    np.testing.assert_allclose(cov, cov0, rtol=0, atol=1e-9)
    other = 0.7 * kernels.ExpSquaredKernel(2.0)                  # predict with ANOTHER kernel object (gp.py:482: the `kernel` argument)
    m1, v1, _ = d.predict(other, r, t, return_var=True)
    m2, v2, _ = s.predict(other, r, t, return_var=True)
    np.testing.assert_allclose(m2, m1, rtol=0, atol=1e-9 * np.abs(m1).max())
    np.testing.assert_allclose(v2, v1, rtol=0, atol=1e-9)


def test_chain_only_and_trace_modes():
    """the two timing aids behind profiles/r04/scale_model.md: the trace leaves the numbers alone, chain-only does not raise"""
    kernel, X, yerr, y, d = _case(3000)
    s = MultiGPUSolver(kernel, devices=[0] * 4, transport=VIRTUAL, nb=256, trace=True)
    s.compute(X, yerr)
    assert abs(s.log_determinant - d.log_determinant) <= 1e-11 * abs(d.log_determinant)
    tr = s.trace()
    nt = -(-3000 // 256)
    assert tr.shape[1] == 5 and set(tr[:, 2].astype(int)) >= {0, 1, 2, 3, 4, 6, 7}
    assert (tr[:, 3] > 0).all()
    potrf = tr[tr[:, 2] == 0]
    assert len(potrf) == nt and sorted(potrf[:, 1].astype(int)) == list(range(nt))             # one diagonal tile per step ...
    assert [int(r_) for r_ in potrf[np.argsort(potrf[:, 1]), 0]] == [s.owner(k, k) for k in range(nt)]     # ... on its owner
    c = MultiGPUSolver(kernel, devices=[0] * 4, transport=VIRTUAL, nb=256, chain_only=True, trace=True)
    c.compute(X, yerr)                                           # (garbage numbers, no exception)
    assert not c.computed
    with pytest.raises(RuntimeError):
        c.dot_solve(y)
    assert 3 not in set(c.trace()[:, 2].astype(int)) and 7 not in set(c.trace()[:, 2].astype(int))


def test_sharded_solver_errors():
    x = np.linspace(0, 3, 700)
    bad = kernels.CosineKernel(log_period=0.0)                              # singular without noise
    for devices in ([0], [0, 0, 0, 0]):
        s = MultiGPUSolver(bad, devices=devices, transport=VIRTUAL, nb=128)
        with pytest.raises(np.linalg.LinAlgError):
            s.compute(x[:, None], np.zeros(700))
        assert not s.computed
        with pytest.raises(RuntimeError):
            s.dot_solve(np.sin(x))
        # the handle is usable afterwards
        s.kernel = 1.0 * kernels.ExpSquaredKernel(1.0)
        s.compute(x[:, None], 0.1 * np.ones(700))
        assert s.computed and np.isfinite(s.log_determinant)
    s = MultiGPUSolver(1.0 * kernels.ExpSquaredKernel(1.0, ndim=2), devices=[0, 0], transport=VIRTUAL)
    with pytest.raises(RuntimeError):
        s.compute(x[:, None], 0.1)                                           # dimension mismatch
    if MOCK is None:
        with pytest.raises(ValueError):
            MultiGPUSolver(bad, devices=[0, 0], transport="rccl").compute(x[:, None], 0.1)      # RCCL: one rank per device
    with pytest.raises(ValueError):
        MultiGPUSolver(bad, devices=[0, 0, 0], grid=(2, 2), transport=VIRTUAL).compute(x[:, None], 0.1)
    with pytest.raises(ValueError):
        MultiGPUSolver(bad, devices=[97], transport=VIRTUAL).compute(x[:, None], 0.1)


@pytest.mark.skipif(MOCK is None, reason="only under the stand-in RCCL library (tests/test_gpu_mgpu_mock_rccl.py)")
def test_stand_in_paired_every_operation():
    """runs LAST in this module: what the stand-in library saw over all the cases above"""
    lib = ctypes.CDLL(MOCK)                                      # (the handle gh_mgpu.hip opened: same path, same library)
    st = (ctypes.c_longlong * 8)()
    lib.mock_rccl_stats(st)
    sets, pairs, nbytes, reduces, errors, pending, biggest, groups = list(st)
    msg = ctypes.create_string_buffer(1024)
    lib.mock_rccl_last_error(msg, 1024)
    assert errors == 0, msg.value
    assert pending == 0
    assert sets >= 20 and pairs > 1000 and nbytes > 100e6 and reduces >= sets and groups > 100 and biggest >= 2
    mode = MultiGPUSolver(1.0 * kernels.ExpSquaredKernel(1.0), devices=[0, 0], transport="rccl", one_comm=ONE_COMM).comm_mode()
    line = ("stand-in RCCL: %d communicator sets, %d matched pairs, %.2f GB, %d all-reduces, %d groups (largest %d operations), "
            "0 errors, 0 left over; communicators in flight per rank: %d" % (sets, pairs, nbytes / 1e9, reduces, groups, biggest, mode))
    print(line)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "mock_rccl_stats_one_comm%d.txt" % int(ONE_COMM)), "w") as f:
            f.write(line + "\n")
