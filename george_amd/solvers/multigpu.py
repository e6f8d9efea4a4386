"""``MultiGPUSolver`` -- the dense solver on several MI355X of ONE process, behind the C ABI.

``GP(kernel, solver=MultiGPUSolver, devices=[0, 1, 2, 3, 4, 5, 6, 7])`` shards the factorisation of
``BasicSolver`` (reference ``src/george/solvers/basic.py:51-121``) block-cyclically over the listed
devices (``gh_mgpu_*``, george_amd/csrc/gh_mgpu.hip: one host thread per device, RCCL over xGMI;
default grid ``len(devices) x 1``: whole tile rows per device in snake order).  The whole solver
protocol: ``compute`` / ``log_determinant`` / ``computed`` / ``dot_solve`` / ``apply_inverse`` /
``apply_sqrt`` / ``get_inverse``, plus the device-resident ``predict`` of ``george_amd.GP`` -- every
O(N^2 R) operation is a tile sweep on the sharded factor with all its right-hand sides together.  The
multi-PROCESS form (one rank per GPU under torch.distributed) is
``george_amd.distributed.DistributedBasicSolver``.
"""
import ctypes as C

import numpy as np

from .. import _native as N
from ..program import DeviceKernel

__all__ = ["MultiGPUSolver", "MultiGPUHODLRSolver"]


class MultiGPUSolver(object):

    def __init__(self, kernel, devices=None, nb=0, grid=None, transport="rccl", plain_cyclic=False, chain_only=False, trace=False,
                 one_comm=False):
        self.kernel = kernel
        if devices is None:
            devices = list(range(max(int(N.lib.gh_device_count()), 1)))
        self.devices = [int(d) for d in devices]
        if not 1 <= len(self.devices) <= 16:
            raise ValueError("devices must list 1..16 device ordinals")
        if transport not in ("rccl", "copy"):
            raise ValueError("transport must be 'rccl' or 'copy'")
        self.nb = int(nb)
        self.grid = tuple(grid) if grid is not None else (0, 0)
        self.transport = transport
        self.flags = ((N.GH_MGPU_PLAIN_CYCLIC if plain_cyclic else 0) | (N.GH_MGPU_CHAIN_ONLY if chain_only else 0) |
                      (N.GH_MGPU_TRACE if trace else 0) | (N.GH_MGPU_ONE_COMM if one_comm else 0))
        self._handle = None
        self._dk = None
        self._computed = False
        self._log_det = None

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

    def _ensure_handle(self):
        if self._handle is None:
            o = N.gh_mgpu_opts()
            o.n_dev = len(self.devices)
            for i, d in enumerate(self.devices):
                o.devices[i] = d
            o.pr, o.pc = int(self.grid[0]), int(self.grid[1])
            o.nb = self.nb
            o.transport = N.GH_MGPU_RCCL if self.transport == "rccl" else N.GH_MGPU_COPY
            o.flags = self.flags
            h = N._vp()
            N.check(N.lib.gh_mgpu_create(C.byref(o), C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                N.lib.gh_mgpu_destroy(h)
            except Exception:
                pass
            self._handle = None

    def grid_shape(self):
        pr, pc, nb = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        N.check(N.lib.gh_mgpu_grid(self._ensure_handle(), C.byref(pr), C.byref(pc), C.byref(nb)))
        return pr.value, pc.value, nb.value

    def comm_mode(self):
        """2: chain and bulk gather on their own RCCL communicators; 1: one communicator (``one_comm=True``, or chosen by
        gh_mgpu_create because two streams of a rank did not dispatch independently); 0: transport "copy" """
        return int(N.lib.gh_mgpu_comm_mode(self._ensure_handle()))

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
        N.check(N.lib.gh_mgpu_compute(h, self._dk.handle, N.ptr(x), len(x), x.shape[1], N.ptr(yerr), C.byref(logdet)))
        self._n = len(x)
        self._x_host = x
        self.log_determinant = logdet.value
        self.computed = not (self.flags & N.GH_MGPU_CHAIN_ONLY)       # (a chain-only run is a timing aid: nothing to solve with)

    def _need(self):
        if not self._computed or self._handle is None:
            raise RuntimeError("you must call 'compute' first")
        return self._handle

    def owner(self, tile_row, tile_col):
        """rank (index into ``devices``) that holds tile (I, J)"""
        return int(N.lib.gh_mgpu_owner(self._ensure_handle(), int(tile_row), int(tile_col)))

    def trace(self):
        """``trace=True``: array of (rank, step, phase, milliseconds, flops or bytes) rows of the last compute()
        (phases: include/george_amd.h, gh_mgpu_get_trace)"""
        h = self._ensure_handle()
        cnt = C.c_int64(0)
        N.check(N.lib.gh_mgpu_get_trace(h, None, 0, C.byref(cnt)))
        out = np.zeros((max(cnt.value, 1), 5))
        N.check(N.lib.gh_mgpu_get_trace(h, N.ptr(out), cnt.value, C.byref(cnt)))
        return out[:cnt.value]

    def apply_inverse(self, y, in_place=False):
        """basic.py:72-87: ``y`` is (n,) or (n, nrhs)."""
        h = self._need()
        yin = y
        y = np.asarray(y, dtype=np.float64)
        if y.ndim < 1 or y.ndim > 2 or y.shape[0] != self._n:
            raise ValueError("dimension mismatch")
        yc = np.ascontiguousarray(y)
        nrhs = 1 if yc.ndim == 1 else yc.shape[1]
        out = np.empty_like(yc)
        if nrhs > 0:
            N.check(N.lib.gh_mgpu_solve(h, N.ptr(yc), nrhs, N.ptr(out)))
        if in_place and isinstance(yin, np.ndarray) and yin.dtype == np.float64:
            try:
                yin[...] = out
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
        N.check(N.lib.gh_mgpu_dot_solve(h, N.ptr(y), C.byref(out)))
        return out.value

    def get_inverse(self):
        """basic.py:116-121: ``cho_solve(factor, I)`` in column chunks through the sharded sweeps."""
        h = self._need()
        out = np.empty((self._n, self._n), dtype=np.float64)
        N.check(N.lib.gh_mgpu_get_inverse(h, N.ptr(out)))
        return out

    def apply_sqrt(self, r):
        """basic.py:104-114: ``r @ U`` with ``U^T U = K``."""
        h = self._need()
        r = N.as_f64(r)
        one_d = r.ndim == 1
        r2 = np.ascontiguousarray(r.reshape(1, -1) if one_d else r)
        if r2.shape[1] != self._n:
            raise ValueError("dimension mismatch")
        out = np.empty_like(r2)
        N.check(N.lib.gh_mgpu_apply_sqrt(h, N.ptr(r2), r2.shape[0], N.ptr(out)))
        return out[0] if one_d else out

    def predict(self, kernel, r, xs, return_var=False, return_cov=False):
        """mean / variance / covariance terms of gp.py:532-545 on the sharded factor (as ``BasicSolver.predict``)."""
        h = self._need()
        dk = DeviceKernel(kernel) if kernel is not self.kernel else self._dk
        r, xs = N.as_f64(r).reshape(-1), N.as_f64(xs)
        m = len(xs)
        mu = np.empty(m)
        var = np.empty(m) if return_var else None
        if return_cov and m > 2048:
            # the device call offers the full covariance up to 2048 test points: beyond it, the mean from the device and
            # gp.py:543-545 as written, on the sharded solves
            N.check(N.lib.gh_mgpu_predict(h, dk.handle, N.ptr(r), N.ptr(xs), m, N.ptr(mu), None, None))
            Kxs = kernel.get_value(xs, self._x_host)
            cov = kernel.get_value(xs) - np.dot(Kxs, self.apply_inverse(np.ascontiguousarray(Kxs.T)))
            if var is not None:
                var[:] = np.diag(cov)
            return mu, var, cov
        cov = np.empty((m, m)) if return_cov else None
        N.check(N.lib.gh_mgpu_predict(h, dk.handle, N.ptr(r), N.ptr(xs), m, N.ptr(mu), N.ptr(var), N.ptr(cov)))
        return mu, var, cov

    # pickling drops the (device-resident, sharded) factor, as the reference's native solver does (hodlr.py:69-76)
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = None
        state["_dk"] = None
        state["_computed"] = False
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)


class MultiGPUHODLRSolver(MultiGPUSolver):
    """The HODLR solver with its tree split over several MI355X of one process (``gh_hodlr_mgpu_*``,
    george_amd/csrc/gh_hodlr.hip): the top log2(len(devices)) levels are shared, each device owns one
    sub-tree.  Same keywords as ``HODLRSolver`` (reference ``src/george/solvers/hodlr.py:13-76``:
    ``min_size=100, tol=0.1, seed=42``) plus ``devices``; same node-by-node random streams as the
    single-GPU solver, so ranks and answers agree with it to rounding.  ``len(devices)`` must be a power
    of two and ``N / len(devices) >= 2 * min_size``.  ``apply_sqrt`` raises ``NotImplementedError``
    (hodlr.py:62-64); pickling drops the factor (:69-76)."""

    def __init__(self, kernel, min_size=100, tol=0.1, seed=42, devices=None, max_rank=0):
        if devices is None:                                  # all visible devices, cut down to a power of two (3, 6, 7 GPUs: 2, 4, 4)
            have = max(int(N.lib.gh_device_count()), 1)
            devices = list(range(1 << (min(have, 16).bit_length() - 1)))
        super(MultiGPUHODLRSolver, self).__init__(kernel, devices=devices)
        n = len(self.devices)
        if n & (n - 1):
            raise ValueError("the HODLR tree is split over 1, 2, 4, 8 or 16 devices (got %d)" % n)
        self.min_size, self.tol, self.seed, self.max_rank = min_size, tol, seed, int(max_rank)

    def _ensure_handle(self):
        if self._handle is None:
            o = N.gh_hodlr_mgpu_opts()
            o.n_dev = len(self.devices)
            for i, d in enumerate(self.devices):
                o.devices[i] = d
            o.min_size, o.seed, o.max_rank, o.tol = int(self.min_size), int(self.seed), self.max_rank, float(self.tol)
            h = N._vp()
            N.check(N.lib.gh_hodlr_mgpu_create(C.byref(o), C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                N.lib.gh_hodlr_mgpu_destroy(h)
            except Exception:
                pass
            self._handle = None

    predict = None           # (GP.predict then takes the reference's generic path on apply_inverse, gp.py:482-545)
    owner = trace = get_inverse_sharded = None

    def grid_shape(self):
        raise NotImplementedError("the HODLR split has no process grid: see rows()")

    def get_inverse(self):
        """hodlr.h / _hodlr.cpp:194-199: solve against the identity."""
        return self.apply_inverse(np.eye(self._n))

    def rows(self):
        """[(first row, number of rows)] of every device's sub-tree (after compute())."""
        n = len(self.devices)
        r0, nr = (C.c_int64 * n)(), (C.c_int64 * n)()
        N.check(N.lib.gh_hodlr_mgpu_rows(self._need(), r0, nr))
        return [(int(r0[i]), int(nr[i])) for i in range(n)]

    def ranks(self):
        """ACA ranks of all internal nodes, level by level, left to right (as ``HODLRSolver.ranks()``)."""
        h = self._need()
        cap = 1 << 16
        buf, cnt = (C.c_int32 * cap)(), C.c_int32(0)
        N.check(N.lib.gh_hodlr_mgpu_ranks(h, buf, cap, C.byref(cnt)))
        return [int(buf[i]) for i in range(cnt.value)]

    def compute(self, x, yerr):
        """hodlr.h:75-103 over the split tree.  ``yerr`` already contains the white noise (gp.py:330)."""
        x = N.as_f64(x)
        if x.ndim != 2:
            raise ValueError("x must be (nsamples, ndim)")
        yerr = N.as_f64(np.zeros(len(x)) + yerr)
        self._computed = False
        self._dk = DeviceKernel(self.kernel)
        if x.shape[1] != self._dk.ndim:
            raise RuntimeError("dimension mismatch")
        key = (tuple(self.devices), int(self.min_size), int(self.seed), self.max_rank, float(self.tol))
        if self._handle is not None and getattr(self, "_handle_key", None) != key:
            N.lib.gh_hodlr_mgpu_destroy(self._handle)          # options were changed on the instance
            self._handle = None
        self._handle_key = key
        h = self._ensure_handle()
        logdet = C.c_double(0.0)
        N.check(N.lib.gh_hodlr_mgpu_compute(h, self._dk.handle, N.ptr(x), len(x), x.shape[1], N.ptr(yerr), C.byref(logdet)))
        self._n = len(x)
        self.log_determinant = logdet.value
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        """hodlr.h:107-114: ``y`` is (n,) or (n, nrhs)."""
        h = self._need()
        yin = y
        y = np.asarray(y, dtype=np.float64)
        if y.ndim < 1 or y.ndim > 2 or y.shape[0] != self._n:
            raise ValueError("dimension mismatch")
        yc = np.ascontiguousarray(y)
        nrhs = 1 if yc.ndim == 1 else yc.shape[1]
        out = np.empty_like(yc)
        if nrhs > 0:
            N.check(N.lib.gh_hodlr_mgpu_solve(h, N.ptr(yc), nrhs, N.ptr(out)))
        if in_place and isinstance(yin, np.ndarray) and yin.dtype == np.float64:
            try:
                yin[...] = out
                return yin
            except (ValueError, TypeError):
                pass
        return out

    def dot_solve(self, y):
        """hodlr.h:116-120."""
        h = self._need()
        y = N.as_f64(y).reshape(-1)
        if len(y) != self._n:
            raise ValueError("dimension mismatch")
        out = C.c_double(0.0)
        N.check(N.lib.gh_hodlr_mgpu_dot_solve(h, N.ptr(y), C.byref(out)))
        return out.value

    def apply_sqrt(self, r):
        raise NotImplementedError("apply_sqrt is not implemented for the HODLRSolver")
