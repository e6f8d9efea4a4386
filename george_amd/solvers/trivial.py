"""Diagonal-only solver for the kernel-less GP (reference
``src/george/solvers/trivial.py:11-35``).  O(N) NumPy on the host by design: it
is not on the hot path and has no matrix to factor."""
import numpy as np

__all__ = ["TrivialSolver"]

_EMPTY_KERNEL_TYPE = 4


class TrivialSolver(object):

    def __init__(self, kernel=None):
        if kernel is not None and kernel.kernel_type != _EMPTY_KERNEL_TYPE:
            raise ValueError("the trivial solver doesn't work with a kernel")
        self.computed = False
        self.log_determinant = None

    def compute(self, x, yerr):
        self._ivar = 1.0 / yerr ** 2
        self.log_determinant = 2 * np.sum(np.log(yerr))
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        out = y if in_place else np.array(y)
        out[:] *= self._ivar
        return out

    def dot_solve(self, y):
        return np.sum(y ** 2 * self._ivar)

    def apply_sqrt(self, r):
        return r * np.sqrt(self._ivar)
