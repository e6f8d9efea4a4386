"""``KernelInterface`` -- HIP-backed drop-in for the reference's pybind11 class
of the same name (``src/george/kernel_interface.cpp:10-35, 47-167``).

Same seven methods, same argument order, shapes, dtypes and error behaviour;
the work happens in ``gh_kernel_*`` (george_amd/csrc/gh_kmat.hip).
"""
import numpy as np

from . import _native as N
from .program import DeviceKernel


class KernelInterface(object):

    def __init__(self, kernel_spec):
        self._spec = kernel_spec
        self._dk = DeviceKernel(kernel_spec)

    # kernel_interface.cpp:17-18
    def ndim(self):
        return self._dk.ndim

    def size(self):
        return self._dk.size

    @property
    def handle(self):
        return self._dk.handle

    def _x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != self._dk.ndim:
            raise RuntimeError("dimension mismatch")          # kernel_interface.cpp:51
        return x

    def value_general(self, x1, x2):                          # kernel_interface.cpp:47-60
        x1, x2 = self._x(x1), self._x(x2)
        out = np.empty((len(x1), len(x2)), dtype=np.float64)
        N.check(N.lib.gh_kernel_value_general(self.handle, N.ptr(x1), len(x1), N.ptr(x2), len(x2), N.ptr(out)))
        return out

    def value_symmetric(self, x):                             # kernel_interface.cpp:62-77
        x = self._x(x)
        out = np.empty((len(x), len(x)), dtype=np.float64)
        N.check(N.lib.gh_kernel_value_symmetric(self.handle, N.ptr(x), len(x), N.ptr(out)))
        return out

    def value_diagonal(self, x1, x2):                         # kernel_interface.cpp:79-90
        x1, x2 = self._x(x1), self._x(x2)
        if len(x1) != len(x2):
            raise RuntimeError("dimension mismatch")
        out = np.empty(len(x1), dtype=np.float64)
        N.check(N.lib.gh_kernel_value_diagonal(self.handle, N.ptr(x1), N.ptr(x2), len(x1), N.ptr(out)))
        return out

    def _which(self, which):
        which = np.ascontiguousarray(which, dtype=np.uint32)
        if which.shape != (self._dk.size,):
            raise RuntimeError("dimension mismatch")
        return which

    def gradient_general(self, which, x1, x2):                # kernel_interface.cpp:92-107
        which, x1, x2 = self._which(which), self._x(x1), self._x(x2)
        out = np.empty((len(x1), len(x2), self._dk.size), dtype=np.float64)
        if out.size:
            N.check(N.lib.gh_kernel_gradient_general(self.handle, N.ptr(which), N.ptr(x1), len(x1),
                                                     N.ptr(x2), len(x2), N.ptr(out)))
        return out

    def gradient_symmetric(self, which, x):                   # kernel_interface.cpp:109-125
        which, x = self._which(which), self._x(x)
        out = np.empty((len(x), len(x), self._dk.size), dtype=np.float64)
        if out.size:
            N.check(N.lib.gh_kernel_gradient_symmetric(self.handle, N.ptr(which), N.ptr(x), len(x), N.ptr(out)))
        return out

    def x1_gradient_general(self, x1, x2):                    # kernel_interface.cpp:127-141
        x1, x2 = self._x(x1), self._x(x2)
        out = np.empty((len(x1), len(x2), self._dk.ndim), dtype=np.float64)
        N.check(N.lib.gh_kernel_x1_gradient_general(self.handle, N.ptr(x1), len(x1), N.ptr(x2), len(x2), N.ptr(out)))
        return out

    def x2_gradient_general(self, x1, x2):                    # kernel_interface.cpp:143-157
        x1, x2 = self._x(x1), self._x(x2)
        out = np.empty((len(x1), len(x2), self._dk.ndim), dtype=np.float64)
        N.check(N.lib.gh_kernel_x2_gradient_general(self.handle, N.ptr(x1), len(x1), N.ptr(x2), len(x2), N.ptr(out)))
        return out

    # pickle support mirrors kernel_interface.cpp:159-167 (state == the spec)
    def __reduce__(self):
        return (KernelInterface, (self._spec,))
