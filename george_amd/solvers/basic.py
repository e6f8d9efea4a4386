"""``BasicSolver`` -- dense Cholesky on one MI355X.

Drop-in for the reference's ``BasicSolver`` (``src/george/solvers/basic.py``):
same constructor, methods, shapes and error behaviour, but ``compute`` builds
the covariance matrix on the device from ``(kernel, x)`` and factorises it there
(blocked fp64-MFMA Cholesky, george_amd/csrc/gh_chol.hip); the N x N matrix
never visits the host.  Extra, optional entry points (``predict``, ``grad``)
keep the GP glue of ``gp.py:482-545`` / ``:429-466`` device-resident;
:class:`george_amd.GP` uses them when present, the reference GP simply ignores
them.
"""
import ctypes as C

import numpy as np

from .. import _native as N
from ..program import DeviceKernel

__all__ = ["BasicSolver"]

import atexit


class BasicSolver(object):

    def __init__(self, kernel, device=0, nb=0, profile=False, lookahead=True):
        self.kernel = kernel
        self._computed = False
        self._log_det = None
        self._opts = dict(device=int(device), nb=int(nb), profile=bool(profile), lookahead=bool(lookahead))
        self._handle = None
        self._dk = None
        self._factor_state = None

    # -- properties (basic.py:25-49)
    @property
    def computed(self):
        return self._computed

    @computed.setter
    def computed(self, v):
        self._computed = v

    @property
    def log_determinant(self):
        return self._log_det

    @log_determinant.setter
    def log_determinant(self, v):
        self._log_det = v

    # -- lifetime.  `GP.compute` instantiates a NEW solver on every call (gp.py:327) -- thousands of
    # times inside an optimiser loop -- so native handles (and the N x N device buffers they own)
    # are recycled through a small per-configuration pool instead of being hipMalloc'ed each time.
    # A handle keeps every buffer it ever grew (the factor, and after grad / get_inverse / predict up
    # to three more N x N work arrays: ~100 GB at N = 65536).  It is parked AS IT IS while the pool
    # stays under _POOL_MAX_BYTES of device memory in total (an optimiser iterate drops its solver
    # and the next one picks the same handle up: freeing the work arrays in between meant two or
    # three hipFree + hipMalloc of 8 N^2 bytes per iterate -- the fused objective ran 2x slower than
    # the separate calls in round 2's driver run); above that budget it is trimmed to its factor,
    # and above it still it is destroyed.  At most _POOL_MAX handles are parked per option set;
    # ``BasicSolver.release_pool()`` empties the pool, and it is emptied at interpreter exit.
    _POOL = {}
    _POOL_MAX = 2
    _POOL_MAX_BYTES = 112 << 30
    # ``pickle`` of a computed solver carries the factor (reference behaviour, tests/test_pickle.py:21-36)
    # up to this many points (8 GB of packed lower triangle at 46340); beyond it the state drops the
    # factor and the solver comes back un-computed, like the reference's own native solver does
    # (solvers/hodlr.py:69-76).  Set to 0 to always drop, to None to always keep.
    PICKLE_FACTOR_MAX_N = 32768

    def _pool_key(self):
        return tuple(sorted(self._opts.items()))

    def _ensure_handle(self):
        if self._handle is None:
            free = BasicSolver._POOL.get(self._pool_key())
            if free:
                self._handle = free.pop()
                return self._handle
            o = N.gh_chol_opts()
            o.device, o.nb = self._opts["device"], self._opts["nb"]
            o.profile, o.lookahead = int(self._opts["profile"]), int(self._opts["lookahead"])
            h = N._vp()
            N.check(N.lib.gh_chol_create(C.byref(o), C.byref(h)))
            self._handle = h
        return self._handle

    @staticmethod
    def _pooled_bytes():
        return sum(int(N.lib.gh_chol_device_bytes(h)) for free in BasicSolver._POOL.values() for h in free)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                free = BasicSolver._POOL.setdefault(self._pool_key(), [])
                room = BasicSolver._POOL_MAX_BYTES - BasicSolver._pooled_bytes()
                if len(free) < BasicSolver._POOL_MAX and int(N.lib.gh_chol_device_bytes(h)) <= room:
                    free.append(h)                     # as it is: the next iterate re-uses the work arrays too
                elif len(free) < BasicSolver._POOL_MAX:
                    N.lib.gh_chol_trim(h)              # keep the factor-sized buffers, drop the work arrays
                    if int(N.lib.gh_chol_device_bytes(h)) <= room:
                        free.append(h)
                    else:
                        N.lib.gh_chol_destroy(h)
                else:
                    N.lib.gh_chol_destroy(h)
            except Exception:
                pass
            self._handle = None

    def _retry_without_parked_memory(self, call):
        """Run ``call(handle)``; on MemoryError give back what dead solvers left parked on the device -- the handle pool
        (up to _POOL_MAX_BYTES, work arrays included) and the native block cache (up to 48 GB) -- and try ONCE more: a
        new pool key, a multi-GPU handle or the application's own allocations must not fail because of memory nobody uses."""
        try:
            return call(self._ensure_handle())
        except MemoryError:
            type(self).release_pool()
            BasicSolver.release_pool()
            N.lib.gh_release_caches(int(self._opts["device"]))
            return call(self._ensure_handle())

    @classmethod
    def release_pool(cls):
        """Destroy every parked native handle (frees their device memory)."""
        for free in cls._POOL.values():
            while free:
                try:
                    N.lib.gh_chol_destroy(free.pop())
                except Exception:
                    pass
        cls._POOL.clear()

    # Pickling.  The reference's BasicSolver pickles computed (its factor is a NumPy array,
    # basic.py:68; tests/test_pickle.py:21-36); here the factor is downloaded (packed lower triangle +
    # diagonal-block inverses) into the state and uploaded again on first use after unpickling.
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = None
        state["_dk"] = None
        state.pop("_factor_state", None)
        keep = self._computed and self._handle is not None and (
            BasicSolver.PICKLE_FACTOR_MAX_N is None or self._n <= BasicSolver.PICKLE_FACTOR_MAX_N)
        if keep:
            h = self._handle
            L = np.empty(int(N.lib.gh_chol_factor_size(h)))
            dinv = np.empty(int(N.lib.gh_chol_dinv_size(h)))
            N.check(N.lib.gh_chol_export_factor(h, N.ptr(L), N.ptr(dinv)))
            state["_factor_state"] = (L, dinv)
        elif self._computed and getattr(self, "_factor_state", None) is not None:
            state["_factor_state"] = self._factor_state       # unpickled and never used: pass it on
        else:
            state["_computed"] = False
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("_factor_state", None)

    def _restore(self):
        """Upload a pickled factor into a fresh native handle (first use after unpickling)."""
        L, dinv = self._factor_state
        self._dk = DeviceKernel(self.kernel)
        h = self._ensure_handle()
        N.check(N.lib.gh_chol_import_factor(h, self._n, self._x_host.shape[1], N.ptr(self._x_host), N.ptr(L), N.ptr(dinv),
                                            float(self._log_det)))
        self._factor_state = None

    # -- the solver protocol
    def compute(self, x, yerr):
        """basic.py:51-70.  ``yerr`` already contains the white noise (gp.py:330)."""
        x = N.as_f64(x)
        if x.ndim != 2:
            raise ValueError("x must be (nsamples, ndim)")
        yerr = N.as_f64(np.zeros(len(x)) + yerr)
        self._computed = False
        self._dk = DeviceKernel(self.kernel)
        if x.shape[1] != self._dk.ndim:
            raise RuntimeError("dimension mismatch")
        h = self._ensure_handle()
        logdet = C.c_double(0.0)
        self._factor_state = None
        self._retry_without_parked_memory(lambda hh: N.check(N.lib.gh_chol_compute(
            hh, self._dk.handle, N.ptr(x), len(x), x.shape[1], N.ptr(yerr), C.byref(logdet))))
        self._n = len(x)
        self._x_host = x                     # (the inputs travel with a pickled factor: predict / grad need them)
        self.log_determinant = logdet.value
        self.computed = True

    def objective(self, x, yerr, r, which=None, want_grad=True):
        """``compute(x, yerr)`` + ``r^T K^-1 r`` + (optionally) the kernel part of the gradient of the
        log-likelihood in ONE device call (gh_chol_objective; gp.py:470-480 with :303-337, :369-397,
        :429-466).  Returns ``(log_det, quad, grad_all | None, alpha | None, diagA | None)`` and leaves
        the solver computed."""
        x = N.as_f64(x)
        if x.ndim != 2:
            raise ValueError("x must be (nsamples, ndim)")
        n = len(x)
        yerr = N.as_f64(np.zeros(n) + yerr)
        r = N.as_f64(r).reshape(-1)
        if len(r) != n:
            raise ValueError("dimension mismatch")
        self._computed = False
        self._factor_state = None
        self._dk = DeviceKernel(self.kernel)
        if x.shape[1] != self._dk.ndim:
            raise RuntimeError("dimension mismatch")
        h = self._ensure_handle()
        logdet, quad = C.c_double(0.0), C.c_double(0.0)
        g = alpha = diagA = wh = None
        if want_grad:
            wh = np.ascontiguousarray(np.ones(max(self._dk.size, 1)) if which is None else which, dtype=np.uint32)
            g = np.zeros(max(self._dk.size, 1))
            alpha, diagA = np.empty(n), np.empty(n)
        self._retry_without_parked_memory(lambda hh: N.check(N.lib.gh_chol_objective(
            hh, self._dk.handle, N.ptr(x), n, x.shape[1], N.ptr(yerr), N.ptr(r), N.ptr(wh),
            C.byref(logdet), C.byref(quad), N.ptr(g), N.ptr(alpha), N.ptr(diagA))))
        self._n = n
        self._x_host = x
        self.log_determinant = logdet.value
        self.computed = True
        return logdet.value, quad.value, (g[:self._dk.size] if g is not None else None), alpha, diagA

    def _need(self):
        if self._computed and self._handle is None and getattr(self, "_factor_state", None) is not None:
            self._restore()
        if not self._computed or self._handle is None:
            raise RuntimeError("you must call 'compute' first")
        return self._handle

    def apply_inverse(self, y, in_place=False):
        """basic.py:72-87 (``cho_solve``): ``y`` is (n,) or (n, nrhs)."""
        h = self._need()
        yin = y
        y = np.asarray(y, dtype=np.float64)
        if y.shape[0] != self._n or y.ndim > 2:
            raise ValueError("dimension mismatch")
        yc = np.ascontiguousarray(y)
        nrhs = 1 if yc.ndim == 1 else yc.shape[1]
        writable_inplace = (in_place and isinstance(yin, np.ndarray) and yin.dtype == np.float64
                            and yin.flags.c_contiguous and yin.flags.writeable)
        out = yin if writable_inplace else np.empty_like(yc)
        if nrhs > 0:
            N.check(N.lib.gh_chol_solve(h, N.ptr(yc), nrhs, N.ptr(out)))
        if in_place and not writable_inplace and isinstance(yin, np.ndarray):
            try:
                yin[...] = out                     # honour overwrite_b for non-contiguous callers (gp.py:296-301)
                return yin
            except (ValueError, TypeError):
                pass
        return out

    def dot_solve(self, y):
        """basic.py:89-102."""
        h = self._need()
        y = N.as_f64(y).reshape(-1)
        if len(y) != self._n:
            raise ValueError("dimension mismatch")
        out = C.c_double(0.0)
        N.check(N.lib.gh_chol_dot_solve(h, N.ptr(y), C.byref(out)))
        return out.value

    def apply_sqrt(self, r):
        """basic.py:104-114: ``r @ U`` with ``U^T U = K``."""
        h = self._need()
        r = N.as_f64(r)
        one_d = r.ndim == 1
        r2 = r.reshape(1, -1) if one_d else r
        if r2.shape[1] != self._n:
            raise ValueError("dimension mismatch")
        out = np.empty_like(r2)
        N.check(N.lib.gh_chol_apply_sqrt(h, N.ptr(r2), r2.shape[0], N.ptr(out)))
        return out[0] if one_d else out

    def get_inverse(self):
        """basic.py:116-121."""
        h = self._need()
        out = np.empty((self._n, self._n), dtype=np.float64)
        N.check(N.lib.gh_chol_get_inverse(h, N.ptr(out)))
        return out

    # -- fused, device-resident GP glue (optional protocol extensions)
    def predict(self, kernel, r, xs, return_var=False, return_cov=False):
        """mean / variance / covariance terms of gp.py:532-545 for residual ``r = y - mean``:
        returns ``K* K^-1 r`` and (optionally) ``diag`` or full ``K** - K* K^-1 K*^T``."""
        h = self._need()
        dk = DeviceKernel(kernel) if kernel is not self.kernel else self._dk
        r, xs = N.as_f64(r).reshape(-1), N.as_f64(xs)
        m = len(xs)
        mu = np.empty(m)
        var = np.empty(m) if return_var else None
        cov = np.empty((m, m)) if return_cov else None
        N.check(N.lib.gh_chol_predict(h, dk.handle, N.ptr(r), N.ptr(xs), m, N.ptr(mu), N.ptr(var), N.ptr(cov)))
        return mu, var, cov

    def grad(self, r, which):
        """kernel part of gp.py:429-466: returns (grad over ALL kernel params (masked ones 0),
        alpha = K^-1 r, diag(alpha alpha^T - K^-1))."""
        h = self._need()
        r = N.as_f64(r).reshape(-1)
        which = np.ascontiguousarray(which, dtype=np.uint32)
        g = np.zeros(max(self._dk.size, 1))
        alpha, diagA = np.empty(self._n), np.empty(self._n)
        N.check(N.lib.gh_chol_grad(h, self._dk.handle, N.ptr(which), N.ptr(r), N.ptr(g), N.ptr(alpha), N.ptr(diagA)))
        return g[:self._dk.size], alpha, diagA

    def profile(self):
        p = N.gh_chol_profile()
        N.check(N.lib.gh_chol_get_profile(self._need(), C.byref(p)))
        return dict(ms_total=p.ms_total, ms_build=p.ms_build, ms_panel=p.ms_panel, ms_trailing=p.ms_trailing,
                    trailing_flops=p.trailing_flops, n_trailing=p.n_trailing, ms_solve=p.ms_solve)


atexit.register(BasicSolver.release_pool)
