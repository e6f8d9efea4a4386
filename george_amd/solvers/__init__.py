"""Solver plugins for :class:`george_amd.GP` (duck-typed, as in the reference:
``docs/user/solvers.rst:12-21``, ``src/george/gp.py:125-133,327``).  They also
plug into the *reference's own* ``george.GP(kernel, solver=...)`` unchanged."""
from .trivial import TrivialSolver
from .basic import BasicSolver
from .hodlr import HODLRSolver
from .multigpu import MultiGPUSolver, MultiGPUHODLRSolver

__all__ = ["TrivialSolver", "BasicSolver", "HODLRSolver", "MultiGPUSolver", "MultiGPUHODLRSolver"]
