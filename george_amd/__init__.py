"""george_amd -- an MI355X-native Gaussian-process solver backend with the
``george.GP`` / ``BasicSolver`` / ``HODLRSolver`` API surface.

Everything numeric runs in hand-written HIP behind a C ABI
(``include/george_amd.h`` -> ``george_amd/csrc/libgeorge_amd.so``); importing
this package fails loudly when that library has not been built, and every call
fails loudly when no MI355X is visible.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _native                     # noqa: F401  (raises ImportError if the HIP library is missing)
from . import kernels
from .gp import GP
from .metrics import Metric
from .solvers import TrivialSolver, BasicSolver, HODLRSolver, MultiGPUSolver, MultiGPUHODLRSolver
from .kernel_interface import KernelInterface

__all__ = ["__version__", "kernels", "GP", "Metric", "TrivialSolver", "BasicSolver",
           "HODLRSolver", "MultiGPUSolver", "MultiGPUHODLRSolver", "KernelInterface"]


def device_count():
    return _native.lib.gh_device_count()
