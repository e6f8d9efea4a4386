"""``HODLRSolver`` -- level-batched HODLR factorisation on one MI355X.

Drop-in for the reference's ``HODLRSolver`` (``src/george/solvers/hodlr.py:13-76``
over ``_hodlr.cpp`` / ``include/george/hodlr.h``): same constructor keywords
(``min_size=100, tol=0.1, seed=42``), same methods, ``apply_sqrt`` raises
``NotImplementedError`` (hodlr.py:62-64), pickling drops the factor (:69-76).
"""
import atexit
import ctypes as C
import warnings

import numpy as np

from .. import _native as N
from ..program import DeviceKernel
from .basic import BasicSolver

__all__ = ["HODLRSolver"]


class HODLRSolver(BasicSolver):

    # hodlr.h:147 lets a block's rank grow to min(rows, cols); the level-batched ACA here stops at 1024
    # columns.  A block that needs more (3-D inputs at tight tolerances: docs/user/solvers.rst:40-42) is
    # not low-rank in any useful sense -- the reference then spends O(N r^2) on it -- and up to this many
    # points the solver answers with the EXACT dense device factorisation instead (BasicSolver on the
    # same GPU: every tolerance is met; ``dense_fallback`` is set and a warning is issued); beyond it,
    # or when an explicit ``max_rank`` is too small, compute() raises ValueError.  Never a truncation.
    DENSE_FALLBACK_MAX_N = 98304          # 8 N^2 = 77 GB of the 288 GB

    def __init__(self, kernel, min_size=100, tol=0.1, seed=42, device=0, max_rank=0):
        # max_rank = 0: ranks grow as far as ``tol`` asks (hodlr.h:147), up to the solver's ceiling of
        # 1024 (then: dense fallback, above); an explicit ``max_rank`` that is too small raises ValueError.
        self.dense_fallback = False
        self._dense = None
        self.min_size = min_size
        self.tol = tol
        self.seed = seed
        self._hopts = dict(device=int(device), max_rank=int(max_rank))
        super(HODLRSolver, self).__init__(kernel, device=device)

    # A native handle keeps its tree, its per-level job tables on the device and the measured schedule of its
    # last compute(): inside an optimiser loop (GP makes a NEW solver object per compute, gp.py:327) a fresh handle
    # per iterate cost 3 of the 8.4 ms of a C4 step through this class.  Handles of dropped solvers are parked per
    # option set (at most _HPOOL_MAX each) and picked up by the next solver with the same options.
    _HPOOL = {}
    _HPOOL_MAX = 2
    _HPOOL_TOTAL = 4           # over all option sets: a scan over tolerances must not leave a trail of parked handles

    def _hkey(self):
        return (self._hopts["device"], int(self.min_size), int(self.seed), self._hopts["max_rank"], float(self.tol))

    def _ensure_handle(self):
        if self._handle is None:
            self._handle_key = self._hkey()
            free = HODLRSolver._HPOOL.get(self._handle_key)
            if free:
                self._handle = free.pop()
                if not free:
                    del HODLRSolver._HPOOL[self._handle_key]
                return self._handle
            o = N.gh_hodlr_opts()
            o.device, o.min_size, o.seed = self._hopts["device"], int(self.min_size), int(self.seed)
            o.max_rank, o.tol = self._hopts["max_rank"], float(self.tol)
            h = N._vp()
            N.check(N.lib.gh_hodlr_create(C.byref(o), C.byref(h)))
            self._handle = h
        return self._handle

    def _park_handle(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h is not None and h.value:
            try:
                key = getattr(self, "_handle_key", None)
                pool = HODLRSolver._HPOOL
                if key is None:
                    N.lib.gh_hodlr_destroy(h)
                    return
                free = pool.pop(key, [])                       # (re-inserted last: the dict keeps the option sets oldest first)
                if len(free) >= HODLRSolver._HPOOL_MAX:
                    N.lib.gh_hodlr_destroy(free.pop(0))
                free.append(h)
                pool[key] = free
                while sum(len(v) for v in pool.values()) > HODLRSolver._HPOOL_TOTAL:
                    oldest = next(iter(pool))
                    if pool[oldest]:
                        N.lib.gh_hodlr_destroy(pool[oldest].pop(0))
                    if not pool[oldest]:
                        del pool[oldest]
            except Exception:
                pass

    def __del__(self):
        self._park_handle()

    @classmethod
    def release_pool(cls):
        """Destroy every parked native handle (frees their device memory)."""
        for free in cls._HPOOL.values():
            while free:
                try:
                    N.lib.gh_hodlr_destroy(free.pop())
                except Exception:
                    pass
        cls._HPOOL.clear()

    def compute(self, x, yerr):
        x = N.as_f64(x)
        if x.ndim != 2:
            raise ValueError("x must be (nsamples, ndim)")
        yerr = N.as_f64(np.zeros(len(x)) + yerr)
        self._computed = False
        self._dk = DeviceKernel(self.kernel)
        if x.shape[1] != self._dk.ndim:
            raise RuntimeError("dimension mismatch")
        if self._handle is not None and getattr(self, "_handle_key", None) != self._hkey():
            self._park_handle()               # options were changed on the instance
        h = self._ensure_handle()
        logdet = C.c_double(0.0)
        self.dense_fallback, self._dense = False, None
        try:
            try:
                N.check(N.lib.gh_hodlr_compute(h, self._dk.handle, N.ptr(x), len(x), x.shape[1], N.ptr(yerr), C.byref(logdet)))
            except MemoryError:
                # memory that dead solvers left parked (dense handle pool, HODLR handle pool, native block cache) goes back
                # to the device, and the call is tried once more
                self._handle = None
                N.lib.gh_hodlr_destroy(h)
                BasicSolver.release_pool()
                HODLRSolver.release_pool()
                N.lib.gh_release_caches(self._hopts["device"])
                h = self._ensure_handle()
                N.check(N.lib.gh_hodlr_compute(h, self._dk.handle, N.ptr(x), len(x), x.shape[1], N.ptr(yerr), C.byref(logdet)))
        except N.RankCeilingError as e:
            if len(x) > HODLRSolver.DENSE_FALLBACK_MAX_N:
                raise
            warnings.warn("HODLRSolver: %s -- answering with the dense device solver (exact)" % (e,), RuntimeWarning)
            N.lib.gh_hodlr_destroy(self._handle)          # (frees the HODLR scratch before the N x N matrix is built)
            self._handle = None
            self._dense = BasicSolver(self.kernel, device=self._hopts["device"])
            self._dense.compute(x, yerr)
            self.dense_fallback = True
            self._n = len(x)
            self._log_det = self._dense.log_determinant
            self.computed = True
            return
        self._n = len(x)
        self._log_det = logdet.value
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        # the reference's pybind/Eigen binding always returns a fresh array
        if self._computed and self._dense is not None:
            return self._dense.apply_inverse(y, in_place=False)
        h = self._need()
        y = np.asarray(y, dtype=np.float64)
        if y.shape[0] != self._n or y.ndim > 2:
            raise ValueError("dimension mismatch")
        yc = np.ascontiguousarray(y)
        nrhs = 1 if yc.ndim == 1 else yc.shape[1]
        out = np.empty_like(yc)
        if nrhs > 0:
            N.check(N.lib.gh_hodlr_solve(h, N.ptr(yc), nrhs, N.ptr(out)))
        return out

    def dot_solve(self, y):
        if self._computed and self._dense is not None:
            return self._dense.dot_solve(y)
        h = self._need()
        y = N.as_f64(y).reshape(-1)
        if len(y) != self._n:
            raise ValueError("dimension mismatch")
        out = C.c_double(0.0)
        N.check(N.lib.gh_hodlr_dot_solve(h, N.ptr(y), C.byref(out)))
        return out.value

    def apply_sqrt(self, r):
        raise NotImplementedError("apply_sqrt is not implemented for the HODLRSolver")

    def get_inverse(self):
        if self._computed and self._dense is not None:
            return self._dense.get_inverse()
        h = self._need()
        out = np.empty((self._n, self._n), dtype=np.float64)
        N.check(N.lib.gh_hodlr_get_inverse(h, N.ptr(out)))
        return out

    def ranks(self):
        if self._computed and self._dense is not None:
            return []                             # (no low-rank blocks: the dense factorisation answered)
        buf = (C.c_int32 * 65536)()
        cnt = C.c_int32(0)
        N.check(N.lib.gh_hodlr_ranks(self._need(), buf, 65536, C.byref(cnt)))
        return list(buf[:cnt.value])

    # pickling drops the factor and flags the solver un-computed (hodlr.py:69-76)
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = None
        state["_dk"] = None
        state["_factor_state"] = None
        state["_computed"] = False
        state["_dense"] = None
        state["dense_fallback"] = False          # (it described the factor that is not pickled)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _need(self):
        if not self._computed or self._handle is None:
            raise RuntimeError("you must call 'compute' first")
        return self._handle

    # the fused dense-only extensions do not apply
    predict = None
    grad = None
    profile = None
    objective = None


atexit.register(HODLRSolver.release_pool)
