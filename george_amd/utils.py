"""Small host-side helpers (sampling, sorting, finite differences) with the
signatures of the reference's ``src/george/utils.py:11-92``.  Not on the hot path."""
import numpy as np

__all__ = ["multivariate_gaussian_samples", "nd_sort_samples", "numerical_gradient", "check_gradient"]


def multivariate_gaussian_samples(matrix, N, mean=None):
    """Draw ``N`` samples from N(mean, matrix); one sample is returned 1-D."""
    mean = np.zeros(len(matrix)) if mean is None else mean
    draws = np.random.multivariate_normal(mean, matrix, N)
    return draws[0] if N == 1 else draws


def nd_sort_samples(samples):
    """Indices ordering N-dimensional points by distance from the first one (KD-tree
    query, as the reference does), which keeps HODLR off-diagonal blocks low rank."""
    from scipy.spatial import cKDTree
    samples = np.asarray(samples)
    assert samples.ndim == 2
    _, order = cKDTree(samples).query(samples[0], k=len(samples))
    return np.atleast_1d(order)


def numerical_gradient(f, x, dx=1.234e-6):
    x = np.array(x, dtype=np.float64)
    g = np.empty_like(x)
    for i in range(len(x)):
        x[i] += dx
        fp = f(x)
        x[i] -= 2 * dx
        fm = f(x)
        x[i] += dx
        g[i] = 0.5 * (fp - fm) / dx
    return g


def check_gradient(obj, *args, **kwargs):
    eps = kwargs.pop("eps", 1.23e-5)
    g0 = obj.get_gradient(*args, **kwargs)
    theta = obj.get_parameter_vector()
    for i, t in enumerate(theta):
        theta[i] = t + eps
        obj.set_parameter_vector(theta)
        fp = obj.get_value(*args, **kwargs)
        theta[i] = t - eps
        obj.set_parameter_vector(theta)
        fm = obj.get_value(*args, **kwargs)
        theta[i] = t
        obj.set_parameter_vector(theta)
        assert np.allclose(0.5 * (fp - fm) / eps, g0[i])
