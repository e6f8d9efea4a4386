"""The HODLR tree split over several devices (gh_hodlr_mgpu_*) on the one GPU of the test box:
the SAME device listed 2, 4 and 8 times ("virtual devices") runs the whole protocol -- the ACA of the top
nodes on their runner devices, the rows of their factors dealt to the sub-trees, one sub-tree handle per
rank, the sums of the top levels completed through pinned host memory in every factorisation and solve.

Checked against the single-GPU solver (itself pinned to the reference's hodlr.h: tests/test_gpu_hodlr.py),
with which it shares the node-by-node random streams: same ranks node for node, log-determinant and solves
equal to rounding.  What this cannot cover is the peer copy between two physical devices."""
import numpy as np
import pytest

import zoo
from george_amd import kernels, GP, BasicSolver, HODLRSolver, MultiGPUHODLRSolver

pytestmark = pytest.mark.gpu

HCONF = zoo.hodlr_configs(kernels)


def _agree(s, ref, y, rel=1e-9):
    assert s.computed
    assert abs(s.log_determinant - ref.log_determinant) <= rel * abs(ref.log_determinant), (s.log_determinant, ref.log_determinant)
    a = ref.apply_inverse(y)
    np.testing.assert_allclose(s.apply_inverse(y), a, rtol=0, atol=rel * 10 * np.abs(a).max())
    q = ref.dot_solve(y)
    assert abs(s.dot_solve(y) - q) <= rel * 10 * abs(q)


@pytest.mark.parametrize("name,ndev", [
    ("solver1000", 2), ("solver1000", 4),
    ("C4_4096", 2), ("C4_4096", 4), ("C4_8192", 8),
    ("C4_3000_tol1e-4_seed7", 4),              # odd sizes: the sub-trees differ in size and shape
    ("c5like3d", 2), ("c5like3d_4096_rank400", 4), ("expsq2d", 4),
    ("scaling2000_default", 2),
    ("C4_4096", 1),                            # a split of one: the plain solver behind the same entry points
])
def test_split_matches_single_gpu(name, ndev):
    kernel, x, yerr, y, kw = HCONF[name]
    X = np.ascontiguousarray(x.reshape(len(x), -1))
    ref = HODLRSolver(kernel, **kw)
    ref.compute(X, yerr)
    s = MultiGPUHODLRSolver(kernel, devices=[0] * ndev, **kw)
    s.compute(X, yerr)
    rows = s.rows()
    assert len(rows) == ndev and rows[0][0] == 0 and sum(r[1] for r in rows) == len(x)
    for (a0, an), (b0, _) in zip(rows[:-1], rows[1:]):
        assert a0 + an == b0
    # same tree, same random streams: the ranks agree node for node
    assert s.ranks() == ref.ranks()
    tol = kw["tol"]
    # (the sums of the top levels are added up in a different order: rounding, amplified by cond(K) at most)
    _agree(s, ref, y, rel=1e-9 if tol <= 1e-6 else 1e-7)
    Y2 = np.stack([y, np.cos(3 * X[:, 0])], axis=1)
    a2 = ref.apply_inverse(Y2)
    got = s.apply_inverse(Y2)
    assert got.shape == Y2.shape
    np.testing.assert_allclose(got, a2, rtol=0, atol=1e-6 * np.abs(a2).max())


def test_split_against_dense_and_reference_golden():
    """tests/test_solvers.py:61-75 (HODLR at tol = 1e-10 against the dense answer) on four sub-trees, and the
    reference's own hodlr.h on the same inputs (tests/golden/hodlr.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hodlr.npz"))
    kernel, x, yerr, y, kw = HCONF["C4_8192"]
    X = x[:, None]
    s = MultiGPUHODLRSolver(kernel, devices=[0, 0, 0, 0], **kw)
    s.compute(X, yerr)
    d = BasicSolver(kernel)
    d.compute(X, yerr)
    assert abs(s.log_determinant - d.log_determinant) <= 1e-9 * abs(d.log_determinant)
    assert abs(s.log_determinant - float(g["C4_8192/logdet"])) <= 1e-9 * abs(d.log_determinant)
    a = d.apply_inverse(y)
    np.testing.assert_allclose(s.apply_inverse(y), a, rtol=0, atol=1e-6 * np.abs(a).max())
    np.testing.assert_allclose(s.apply_inverse(y), g["C4_8192/alpha"], rtol=0, atol=1e-6 * np.abs(a).max())


def test_split_many_right_hand_sides_and_repeat():
    """More columns than one pass of a top level takes (256), twice on one handle, then new hyper-parameters."""
    kernel, x, yerr, y, kw = HCONF["C4_4096"]
    X = x[:, None]
    ref = HODLRSolver(kernel, **kw)
    ref.compute(X, yerr)
    s = MultiGPUHODLRSolver(kernel, devices=[0, 0, 0, 0], **kw)
    s.compute(X, yerr)
    rng = np.random.RandomState(5)
    B = rng.randn(len(x), 300)
    want = ref.apply_inverse(B)
    for _ in range(2):
        np.testing.assert_allclose(s.apply_inverse(B), want, rtol=0, atol=1e-7 * np.abs(want).max())
    k2 = 0.3 * kernels.ExpSquaredKernel(2.5)
    ref2 = HODLRSolver(k2, **kw)
    ref2.compute(X, yerr)
    s.kernel = k2
    s.compute(X, yerr)
    _agree(s, ref2, y)


def test_split_in_gp_and_errors():
    kernel, x, yerr, y, kw = HCONF["C4_4096"]
    gp = GP(kernel, solver=MultiGPUHODLRSolver, devices=[0, 0], **kw)
    gp.compute(x, yerr)
    ref = GP(kernel, solver=HODLRSolver, **kw)
    ref.compute(x, yerr)
    assert abs(gp.log_likelihood(y) - ref.log_likelihood(y)) <= 1e-9 * abs(ref.log_likelihood(y))
    # too few points for that many devices: a top node would be a leaf
    s = MultiGPUHODLRSolver(kernel, devices=[0] * 8, min_size=100, tol=1e-10)
    with pytest.raises(ValueError):
        s.compute(x[:700, None], yerr[:700])         # level 2: 175 points, half 87 < min_size
    s = MultiGPUHODLRSolver(kernel, devices=[0, 0], **kw)
    with pytest.raises(RuntimeError):
        s.compute(np.zeros((len(x), 2)), yerr)             # dimension mismatch
    with pytest.raises(RuntimeError):
        s.dot_solve(y)                                     # not computed


def test_split_failure_on_one_device_is_reported_and_the_handle_recovers():
    """A leaf that is not positive definite on ONE sub-tree (forty copies of one point, no noise on them): its rank fails inside its own
    part while the other one already waits in the top level's exchange -- the abort flag must release it, the error must
    be the failing rank's, and the same solver must then factor a healthy matrix (barrier counts and top nodes of the
    aborted run are not carried over)."""
    kernel, x, yerr, y, kw = HCONF["C4_4096"]
    bad = x.copy()
    bad[3000:3040] = bad[3000]                              # a singular block inside sub-tree 1 of 2
    s = MultiGPUHODLRSolver(kernel, devices=[0, 0], **kw)
    noise = yerr.copy()
    noise[3000:3040] = 0.0
    with pytest.raises((np.linalg.LinAlgError, ValueError)) as e:
        s.compute(bad[:, None], noise)
    assert "sub-tree 1" in str(e.value) and "positive definite" in str(e.value)
    assert not s.computed
    ref = HODLRSolver(kernel, **kw)
    ref.compute(x[:, None], yerr)
    for _ in range(2):
        s.compute(x[:, None], yerr)
        _agree(s, ref, y)
