"""Host-side logic (CPU): modeling protocol, kernel spec algebra, metric parsing, GP bookkeeping.
Mirrors the reference's tests/test_modeling.py and the host parts of tests/test_kernels.py,
tests/test_pickle.py."""
import pickle

import numpy as np
import pytest

import george_amd
from george_amd import kernels, GP, Metric
from george_amd.modeling import Model, ModelSet, ConstantModel, CallableModel
from george_amd.solvers import TrivialSolver


class LinearModel(Model):
    parameter_names = ("m", "b")

    def get_value(self, x):
        return self.m * x + self.b


def test_model_protocol():
    m = LinearModel(m=0.5, b=1.0)
    assert m.full_size == 2 and len(m) == 2
    assert m.get_parameter_names() == ("m", "b")
    assert np.allclose(m.get_parameter_vector(), [0.5, 1.0])
    m.freeze_parameter("m")
    assert len(m) == 1 and m.get_parameter_names() == ("b",)
    m.set_parameter_vector([3.0])
    assert m.b == 3.0 and m.m == 0.5
    m.thaw_all_parameters()
    m["m"] = 2.0
    assert m[0] == 2.0 and m["b"] == 3.0
    assert m.get_parameter_dict()["m"] == 2.0
    with pytest.raises(ValueError):
        LinearModel(m=1.0)
    with pytest.raises(ValueError):
        LinearModel(1.0, 2.0, b=3.0)
    m2 = LinearModel(1.0, 2.0, bounds=dict(m=(0.0, 2.0)))
    assert m2.log_prior() == 0.0
    m2.set_parameter("m", 5.0)
    assert m2.log_prior() == -np.inf
    assert m2.check_parameter_vector([1.0, 0.0]) and not m2.check_parameter_vector([9.0, 0.0])
    x = np.linspace(0, 1, 5)
    g = m.get_gradient(x)
    assert g.shape == (2, 5) and np.allclose(g[0], x, atol=1e-4) and np.allclose(g[1], 1.0, atol=1e-4)


def test_modelset_names_and_dirty():
    ms = ModelSet([("a", LinearModel(1.0, 2.0)), ("c", ConstantModel(3.0))])
    assert ms.get_parameter_names() == ("a:m", "a:b", "c:value")
    ms.set_parameter("a:b", 7.0)
    assert ms.get_parameter("a:b") == 7.0 and ms.dirty
    ms.dirty = False
    assert not ms.dirty
    ms.freeze_parameter("a:m")
    assert ms.vector_size == 2 and list(ms.unfrozen_mask) == [False, True, True]
    with pytest.raises(ValueError):
        ms.set_parameter("nope:x", 1.0)
    cm = CallableModel(lambda x: 2 * x)
    assert np.allclose(cm.get_value(np.ones(3)), 2.0) and cm.full_size == 0


def test_kernel_algebra_and_names():
    k = 12. * kernels.ExpSquaredKernel(0.4, ndim=3) + 0.1
    assert not k.is_kernel and k.operator_type == 0
    assert k.k1.kernel_type == 8 and np.isclose(k.k1.log_constant, np.log(0.1 / 3))   # scalar -> Constant(log(b/ndim))
    assert k.k2.operator_type == 1 and k.k2.k2.kernel_type == 9
    assert k.get_parameter_names() == ("k1:log_constant", "k2:k1:log_constant", "k2:k2:metric:log_M_0_0")
    assert np.isclose(k.k2.k2.metric.log_M_0_0, np.log(0.4))
    k2 = np.float64(2.0) * kernels.Matern32Kernel(1.0)          # numpy scalars (kernels.py:39-50)
    assert k2.operator_type == 1
    k.set_parameter_vector(k.get_parameter_vector() + 0.1)
    assert k.dirty
    k.dirty = False
    assert not k.dirty
    with pytest.raises(ValueError):
        kernels.ExpSquaredKernel()                               # missing metric
    with pytest.raises(ValueError):
        kernels.LinearKernel(log_gamma2=0.0)                     # missing constant 'order'
    with pytest.raises(ValueError):
        kernels.ExpSquaredKernel([1.0, 0.1, 10.0, 500], ndim=3)  # tests/test_kernels.py:117-118
    with pytest.raises(ValueError):
        kernels.ExpSquaredKernel(1.0, ndim=2, axes=5)
    b = kernels.Matern52Kernel(1.0, ndim=3, axes=2, block=(-0.1, 0.1))
    assert b.blocked and b.block == [(-0.1, 0.1)]
    assert "Matern52Kernel" in repr(b) and "+" in repr(k)


def test_metric_types():
    assert Metric(2.0, ndim=3).metric_type == 0
    m1 = Metric([1.0, 4.0], ndim=2)
    assert m1.metric_type == 1 and np.allclose(m1.to_matrix(), np.diag([1.0, 4.0]))
    A = np.array([[2.0, 0.3], [0.3, 1.0]])
    m2 = Metric(A, ndim=2)
    assert m2.metric_type == 2 and m2.parameter_names == ("log_L_0_0", "L_0_1", "log_L_1_1")
    assert np.allclose(m2.to_matrix(), A)
    with pytest.raises(ValueError):
        Metric([1.0, -1.0], ndim=2)
    copy = Metric(m2)
    assert np.allclose(copy.get_parameter_vector(), m2.get_parameter_vector())


def test_gp_bookkeeping_with_trivial_solver():
    """EmptyKernel -> TrivialSolver: the whole GP facade runs on the CPU (gp.py:125-130)."""
    gp = GP(mean=0.5, fit_mean=True, white_noise=np.log(0.1), fit_white_noise=True)
    assert gp.solver_type is TrivialSolver
    assert gp.get_parameter_names() == ("mean:value", "white_noise:value")
    x = np.linspace(0, 1, 20)
    y = 0.5 + 0.1 * np.sin(7 * x)
    gp.compute(x, 0.05)
    ll = gp.log_likelihood(y)
    var = 0.05 ** 2 + 0.1
    expect = -0.5 * np.sum((y - 0.5) ** 2 / var) - 0.5 * 20 * np.log(2 * np.pi * var)
    assert np.isclose(ll, expect)
    with pytest.raises(ValueError):
        gp.log_likelihood(y[:-1])
    with pytest.raises(RuntimeError):
        GP().recompute()
    assert gp.nll(gp.get_parameter_vector(), y) == -ll


def test_pickle_kernel_and_solver_state():
    k = 2.0 * kernels.ExpSquaredKernel(1.5, ndim=2) + kernels.CosineKernel(log_period=0.3, ndim=2, axes=1)
    k2 = pickle.loads(pickle.dumps(k, -1))
    assert k2.get_parameter_names() == k.get_parameter_names()
    assert np.allclose(k2.get_parameter_vector(), k.get_parameter_vector())
    s = george_amd.BasicSolver(k)
    s2 = pickle.loads(pickle.dumps(s, -1))
    assert not s2.computed                     # device factors are dropped (hodlr.py:69-76 precedent)
    h = george_amd.HODLRSolver(k, tol=1e-8)
    h2 = pickle.loads(pickle.dumps(h, -1))
    assert h2.tol == 1e-8 and not h2.computed
    with pytest.raises(NotImplementedError):
        h.apply_sqrt(np.zeros(3))
