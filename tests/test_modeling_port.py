"""The reference's tests/test_modeling.py re-stated for george_amd: the modeling protocol around the
GP (mean / white-noise models, freezing, bounds) and the GP-level gradient checks, which run the
HIP solver (the reference's single-point GPs plus 50-point variants)."""
import numpy as np
import pytest

from george_amd import GP, kernels
from george_amd.utils import check_gradient
from george_amd.modeling import Model, ConstantModel, CallableModel


def test_constant_mean():
    check_gradient(ConstantModel(5.0), np.zeros(4))


def test_callable_mean():
    check_gradient(CallableModel(lambda x: 5.0 * x), np.zeros(4))


class LinearWhiteNoise(Model):
    parameter_names = ("m", "b")

    def get_value(self, x):
        return self.m * x + self.b

    @Model.parameter_sort
    def compute_gradient(self, x):
        return dict(m=x, b=np.ones(len(x)))


def test_parameters():
    kernel = 10 * kernels.ExpSquaredKernel(1.0)
    kernel += 0.5 * kernels.RationalQuadraticKernel(log_alpha=0.1, metric=5.0)
    gp = GP(kernel, white_noise=LinearWhiteNoise(1.0, 0.1))
    n = len(gp.get_parameter_vector())
    assert n == len(gp.get_parameter_names())
    assert n - 2 == len(kernel.get_parameter_names())
    gp.freeze_parameter(gp.get_parameter_names()[0])
    assert n - 1 == len(gp.get_parameter_names()) == len(gp.get_parameter_vector())
    gp.freeze_all_parameters()
    assert len(gp.get_parameter_names()) == 0 and len(gp.get_parameter_vector()) == 0
    gp.kernel.thaw_all_parameters()
    gp.white_noise.thaw_all_parameters()
    assert n == len(gp.get_parameter_vector()) == len(gp.get_parameter_names())
    assert np.allclose(kernel[0], np.log(10.))


def test_bounds():
    kernel = 10 * kernels.ExpSquaredKernel(1.0, metric_bounds=[(None, 4.0)])
    kernel += 0.5 * kernels.RationalQuadraticKernel(log_alpha=0.1, metric=5.0)
    gp = GP(kernel, white_noise=LinearWhiteNoise(1.0, 0.1))
    assert len(gp.get_parameter_bounds()) == len(gp.get_parameter_vector())
    gp.freeze_all_parameters()
    gp.thaw_parameter("white_noise:m")
    assert len(gp.get_parameter_bounds()) == len(gp.get_parameter_vector())
    with pytest.raises(ValueError):
        kernels.ExpSine2Kernel(gamma=0.1, log_period=5.0, bounds=[10.0])


def _data(N, seed=1234):
    np.random.seed(seed)
    x = np.sort(np.random.uniform(0, 5, N)) if N > 1 else np.random.uniform(0, 5)
    return x, 5 + np.sin(x)


def _yerr(N):
    # the reference runs these checks on ONE point (tests/test_modeling.py:33-40: N is unused there);
    # the 50-point variants add yerr so that K is well conditioned (without it cond(K) ~ 6e19 and the
    # reference itself fails its own finite-difference check on that data).
    return 0.0 if N == 1 else 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 50])
def test_gp_mean(N):
    x, y = _data(N)
    gp = GP(10. * kernels.ExpSquaredKernel(1.3), mean=5.0, fit_mean=True)
    gp.compute(x, _yerr(N))
    check_gradient(gp, y)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 50])
def test_gp_callable_mean(N):
    x, y = _data(N)
    gp = GP(10. * kernels.ExpSquaredKernel(1.3), mean=CallableModel(lambda x: 5.0 * x))
    gp.compute(x, _yerr(N))
    check_gradient(gp, y)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 50])
def test_gp_white_noise(N):
    x, y = _data(N)
    gp = GP(10. * kernels.ExpSquaredKernel(1.3), mean=5.0, fit_mean=True, white_noise=0.1, fit_white_noise=True)
    gp.compute(x, _yerr(N))
    check_gradient(gp, y)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 50])
def test_gp_callable_white_noise(N):
    x, y = _data(N)
    gp = GP(10. * kernels.ExpSquaredKernel(1.3), mean=5.0, white_noise=LinearWhiteNoise(-6, 0.01), fit_white_noise=True)
    gp.compute(x, _yerr(N))
    check_gradient(gp, y)
    gp.freeze_parameter("white_noise:m")
    check_gradient(gp, y)


@pytest.mark.gpu
def test_dtype_coercion(seed=123):                          # tests/test_kernels.py:11-17
    np.random.seed(seed)
    kernel = 0.1 * kernels.ExpSquaredKernel(1.5)
    kernel.pars = [1, 2]
    gp = GP(kernel)
    gp.compute(np.random.rand(100), 1e-2)
