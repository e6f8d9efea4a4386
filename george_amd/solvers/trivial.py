"""Solver for a GP without a kernel: the covariance is the diagonal ``diag(sigma^2)`` and every
operation of the solver protocol is an element-wise scaling (the reference's counterpart is
``src/george/solvers/trivial.py:11-35``).  Host NumPy by design -- there is no matrix to factor, so
this is not part of the device path."""
import numpy as np

__all__ = ["TrivialSolver"]

_KERNEL_TYPE_EMPTY = 4          # kernel_type of kernels.EmptyKernel


def _check_kernel_free(kernel):
    if kernel is None:
        return
    if getattr(kernel, "kernel_type", None) != _KERNEL_TYPE_EMPTY:
        raise ValueError("the trivial solver doesn't work with a kernel")


class TrivialSolver(object):
    """``K = diag(sigma_i^2)``: weights ``w_i = 1 / sigma_i^2`` are all the state there is."""

    def __init__(self, kernel=None):
        _check_kernel_free(kernel)
        self._weights = None
        self._logdet = None

    # -- protocol attributes
    @property
    def computed(self):
        return self._weights is not None

    @computed.setter
    def computed(self, flag):
        if not flag:
            self._weights = None

    @property
    def log_determinant(self):
        return self._logdet

    # -- protocol methods
    def compute(self, x, yerr):
        sigma = np.asarray(yerr, dtype=np.float64)
        self._logdet = float(np.log(sigma).sum() * 2.0)          # log prod sigma_i^2
        self._weights = np.reciprocal(np.square(sigma))

    def _scale(self, y, in_place):
        w = self._weights if np.ndim(y) == 1 else self._weights[:, None]
        if in_place:
            np.multiply(y, w, out=y)
            return y
        return np.multiply(y, w)

    def apply_inverse(self, y, in_place=False):
        return self._scale(y, in_place)

    def dot_solve(self, y):
        y = np.asarray(y)
        return float(np.dot(self._weights, np.square(y)))

    def apply_sqrt(self, r):
        # same convention as the reference (trivial.py:34-35): scaled by the inverse standard deviation
        return np.sqrt(self._weights) * r

    def get_inverse(self):
        return np.diag(self._weights)
