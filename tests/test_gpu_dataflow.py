"""The dataflow factorisation (george_amd/csrc/gh_dflow.hip: one persistent launch of tile tasks) against the launch
chain (gh_chol.hip, factor_lookahead_deep): the SAME bits -- factor, diagonal inverses, log-determinant, log-likelihood --
at every size, on repeated computes of one handle, with the reference's numbers where the golden file has them.
The reference call both arms replace: /root/reference/src/george/solvers/basic.py:68-69 (cholesky + log-det).
"""
import json
import os

import numpy as np
import pytest

import zoo
import george_amd
from george_amd import kernels, GP, BasicSolver, _native as N

pytestmark = pytest.mark.gpu


class arm:
    """with arm(1): the dataflow factorisation for every Np >= 256; arm(0): the launch chain."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = N.lib.gh_debug_set_dataflow(self.mode)

    def __exit__(self, *a):
        N.lib.gh_debug_set_dataflow(self.prev)


def _factor(kernel, x, yerr):
    s = BasicSolver(kernel)
    s.compute(x, yerr)
    L, dinv = s.__getstate__()["_factor_state"]
    return s.log_determinant, np.array(L), np.array(dinv), s


@pytest.mark.parametrize("n", [129, 256, 300, 1000, 1024, 1100, 1536, 2200, 3072, 4096, 5000])
def test_same_bits_as_the_launch_chain(n):
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    with arm(0):
        ld0, L0, d0, s0 = _factor(kernel, x[:, None], yerr)
        a0 = s0.apply_inverse(y)
    with arm(1):
        ld1, L1, d1, s1 = _factor(kernel, x[:, None], yerr)
        a1 = s1.apply_inverse(y)
    assert ld0 == ld1
    assert np.array_equal(L0, L1), int(np.argmax(L0 != L1))
    assert np.array_equal(d0, d1)
    assert np.array_equal(a0, a1)


@pytest.mark.parametrize("n,kname", [(8192, "expsq"), (12288, "m32"), (16384, "expsq")])
def test_same_bits_at_the_sizes_it_is_for(n, kname):
    x, yerr, y = zoo.bench_data(n)
    k = kernels.ExpSquaredKernel(1.0) if kname == "expsq" else kernels.Matern32Kernel(1.0)
    kernel = np.var(y) * k
    out = []
    for mode in (0, 1, 1):
        with arm(mode):
            gp = GP(kernel)
            gp.compute(x, yerr)
            out.append((gp.solver.log_determinant, gp.log_likelihood(y)))
    assert out[0] == out[1] == out[2], out
    if n == 16384 and kname == "expsq":                      # BASELINE configs[1]: the reference's value
        g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "large.json")))["C2"]
        assert abs(out[1][1] - g["loglike"]) <= 1e-9 * abs(g["loglike"])


def test_one_handle_many_computes_and_sizes():
    """The counters and the queues of a handle are reused: sizes up and down, every result equal to the launch chain's."""
    kernel = 0.7 * kernels.Matern32Kernel(1.3)
    want = {}
    with arm(0):
        s = BasicSolver(kernel)
        for n in (700, 2500, 1300):
            x, yerr, _ = zoo.bench_data(n)
            s.compute(x[:, None], yerr)
            want[n] = s.log_determinant
    with arm(1):
        s = BasicSolver(kernel)
        for rep in range(3):
            for n in (700, 2500, 1300, 2500):
                x, yerr, _ = zoo.bench_data(n)
                s.compute(x[:, None], yerr)
                assert s.log_determinant == want[n], (rep, n)


def test_not_positive_definite_is_reported_as_by_the_launch_chain():
    n = 1500
    x = np.linspace(0, 1, n)
    kernel = 1.0 * kernels.ExpSquaredKernel(100.0)           # numerically singular without a diagonal
    yerr = np.zeros(n)
    infos = []
    for mode in (0, 1):
        with arm(mode):
            s = BasicSolver(kernel)
            with pytest.raises(np.linalg.LinAlgError) as e:
                s.compute(x[:, None], yerr)
            infos.append(str(e.value))
            # the handle still works afterwards
            s.compute(x[:, None], 0.1 * np.ones(n))
            assert np.isfinite(s.log_determinant)
    assert infos[0] == infos[1], infos


def test_fused_objective_through_the_dataflow_arm():
    x, yerr, y = zoo.bench_data(3000)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    vals = []
    for mode in (0, 1):
        with arm(mode):
            gp = GP(kernel)
            gp.compute(x, yerr)
            vals.append((gp.nll(gp.get_parameter_vector(), y), tuple(gp.grad_nll(gp.get_parameter_vector(), y))))
    assert vals[0] == vals[1]


def test_two_threads_factor_at_once():
    """One dataflow factorisation per device at a time (its workers fill the chip): two threads serialise, both finish."""
    import threading
    x, yerr, y = zoo.bench_data(2600)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    with arm(0):
        ref = _factor(kernel, x[:, None], yerr)[0]
    got, err = [], []

    def work():
        try:
            s = BasicSolver(kernel)
            for _ in range(4):
                s.compute(x[:, None], yerr)
                got.append(s.log_determinant)
        except Exception as e:                       # pragma: no cover
            err.append(e)

    with arm(1):
        ts = [threading.Thread(target=work) for _ in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
    assert not err and len(got) == 8 and all(v == ref for v in got)
