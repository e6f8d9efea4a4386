"""Dense GP solve sharded over several MI355X: 2-D block-cyclic Cholesky, one process per GPU.

The N x N covariance matrix is cut into NB x NB tiles; tile (I, J) lives on the rank at position
(prow(I), J mod Pc) of a Pr x Pc process grid.  DEFAULT GRID: Pr = world, Pc = 1 -- whole tile rows per
rank, dealt in "snake" order (prow(I) = I mod 2Pr folded back: 0 1 .. Pr-1 Pr-1 .. 1 0), so that every rank
holds the same share of the lower triangle.  On the xGMI full mesh every pair of GPUs has a link of its
own, and what bounds a step is the CHAIN potrf(k) -> L_kk to the others -> TRSM -> what block column k+1
needs -> potrf(k+1): with whole tile rows per rank only two NB x NB tiles travel on that chain, the TRSM
and the block-column update are split over ALL ranks, and the bulk of the panel (the all-gather every
rank's trailing update needs) moves beside it on a communicator and a stream of its own.  A 2 x 4 grid
puts (N - k NB) / 2 x NB doubles of row panel on one link inside the chain at every step
(profiles/r04/scale_model.md prices both).  `grid=(Pr, Pc)` selects any other
grid (then prow(I) = I mod Pr: plain 2-D block-cyclic).  Every rank BUILDS its own tiles on its own GPU
from (kernel, x) -- nothing is scattered -- and the factorisation proceeds right-looking, one tile column
per step:

  P(k) "panel"   1. the owner of the diagonal tile factors it (gh_dev_potrf_block) and broadcasts
                    L_kk and its diagonal-block inverses down its process column;
                 2. the ranks of that process column TRSM their panel tiles (gh_dev_trsm_right);
                 3. the panel travels: along every process row (the "row panel": from 4 ranks up as
                    a scatter to ALL ranks + forward, so that every xGMI link carries 1/world of it
                    -- `_row_exchange`; else a broadcast inside the row), then an all-gather inside
                    every process column of the tiles that column needs transposed ("column panel");
  U(k) "update"  4. every rank updates its own trailing tiles with fp64-MFMA GEMMs (gh_dev_gemm).

With look-ahead (default on GPUs) U(k) is split: the tiles of block column k+1 first, then the chain of
P(k+1) is issued on a second (high-priority) HIP stream -- its kernels AND its collectives -- and the
column-panel all-gather on a third stream with a communicator of its own, while the rest of U(k) runs
on the main stream; panel workspaces are double-buffered.  Collectives are torch.distributed
(backend "nccl" == RCCL over xGMI on the GPUs; "gloo" in the CPU tests): a broadcast to the 1-3
peers of a row/column is a direct-link transfer on the xGMI mesh.  log|K| and the failure flag need
one all-reduce of a scalar each; r^T K^-1 r = ||L^-1 r||^2 is a left-looking tile sweep (one small
reduce + broadcast per tile row).

The tile arithmetic is delegated to an `ops` object: `HipTileOps` (the product path: device
pointers into the C ABI of include/george_amd.h, launched on torch's CURRENT stream) -- the CPU
tests substitute a NumPy stand-in to exercise the ownership / communication logic with world_size
2, 4 and 8 under gloo.
"""
import contextlib
import math
import os

import numpy as np

__all__ = ["grid_shape", "nccl_options", "HipTileOps", "BlockCyclicCholesky", "DistributedBasicSolver", "DistributedDenseJob"]


def nccl_options():
    """ProcessGroupNCCL options that put RCCL's kernels on a high-priority stream (they share the
    GPU with trailing-update GEMMs that fill every CU), or None when the backend is not nccl."""
    try:
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_backend() != "nccl":
            return None
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:
        return None


def _new_group(dist, ranks):
    opts = nccl_options()
    if opts is not None:
        try:
            return dist.new_group(ranks, pg_options=opts)
        except (TypeError, RuntimeError):        # older signature / options rejected: plain group
            pass
    return dist.new_group(ranks)


# Sub-communicators and the all-to-all capability are a property of the process group, not of one
# factorisation: `GP.compute` builds a new solver at every optimiser evaluation (gp.py:327), and a
# `new_group` per call would leak RCCL communicators and pay their set-up plus the probe each time.
# Keyed by (world, Pr, Pc); every rank creates every group, in the same order, exactly once.
_GROUP_CACHE = {}


def _grid_groups(dist, world, Pr, Pc):
    key = (world, Pr, Pc, id(dist.group.WORLD))          # (a re-initialised process group gets fresh sub-groups)
    hit = _GROUP_CACHE.get(key)
    if hit is None:
        rows = [_new_group(dist, [r * Pc + c for c in range(Pc)]) for r in range(Pr)]
        cols = [_new_group(dist, [r * Pc + c for r in range(Pr)]) for c in range(Pc)]
        # a second communicator per process column for the bulk all-gather of the column panel: a collective in flight
        # on the first one would hold up the next chain transfer (one communicator = one in-order queue)
        bulk = [_new_group(dist, [r * Pc + c for r in range(Pr)]) for c in range(Pc)] if Pr > 1 else [None] * Pc
        hit = _GROUP_CACHE[key] = {"rows": rows, "cols": cols, "bulk": bulk, "a2a": None}
    return hit


def _a2a_capable(dist, ops, cache, world, rank):
    """Uneven all_to_all_single (with empty slots) on the world group; a backend that cannot do it
    raises here on EVERY rank, and all fall back to the in-row broadcast.  Probed once per group set."""
    if cache["a2a"] is None:
        try:
            probe_in = ops.zeros(max(world - 1, 1))
            probe_out = ops.zeros(max(world - 1, 1))
            splits = [0 if q == rank else 1 for q in range(world)]
            dist.all_to_all_single(probe_out[:world - 1], probe_in[:world - 1], output_split_sizes=splits, input_split_sizes=splits)
            cache["a2a"] = True
        except (RuntimeError, NotImplementedError, TypeError):
            cache["a2a"] = False
    return cache["a2a"]


def clear_caches():
    """Forget cached sub-groups and solver workspaces (call before destroy_process_group)."""
    _GROUP_CACHE.clear()
    _CHOL_CACHE.clear()


_CHOL_CACHE = {}          # (n, nb, world, rank, device, lookahead, group) -> [parked BlockCyclicCholesky workspaces]
_CHOL_CACHE_MAX = 2       # parked workspaces in total


def grid_shape(world, grid=None):
    """(Pr, Pc): `grid` if given, else world x 1 (whole tile rows per rank: the
    module docstring says why); "square" asks for the grid a switched network would want (Pr <= Pc, as square as
    the world size allows: 1x2, 2x2, 2x4)."""
    if grid is None:
        return world, 1
    if isinstance(grid, str):
        if grid == "square":
            pr = int(math.isqrt(world))
            while world % pr:
                pr -= 1
            return pr, world // pr
        grid = tuple(int(v) for v in grid.lower().split("x"))
    pr, pc = int(grid[0]), int(grid[1])
    if pr < 1 or pc < 1 or pr * pc != world:
        raise ValueError("grid %dx%d does not hold %d ranks" % (pr, pc, world))
    return pr, pc


class HipTileOps(object):
    """Tile kernels on the local GPU through the C ABI (device pointers).  Every launch goes to
    torch's CURRENT stream, so `with torch.cuda.stream(s):` routes kernels and collectives alike."""

    has_streams = True

    def __init__(self, device, kernel_spec):
        import torch
        from . import _native as N
        from .program import DeviceKernel
        self.torch, self.N = torch, N
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.dk = DeviceKernel(kernel_spec)
        self.ndim = self.dk.ndim

    def _st(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream or None

    def zeros(self, *shape, dtype=None):
        return self.torch.zeros(*shape, dtype=dtype or self.torch.float64, device=self.device)

    def to_device(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def kmat(self, x, n, yerr, row0, nrows, col0, ncols, out):
        self.N.check(self.N.lib.gh_dev_kmat_block(self.dk.handle, x.data_ptr(), n, self.ndim, yerr.data_ptr(),
                                                  row0, nrows, col0, ncols, out.data_ptr(), out.stride(0), self._st()))

    def potrf(self, a, dinv, info, base):
        self.N.check(self.N.lib.gh_dev_potrf_block(a.data_ptr(), a.stride(0), a.shape[0], dinv.data_ptr(),
                                                   info.data_ptr(), base, self._st()))

    def trsm(self, l11, dinv, a21):
        self.N.check(self.N.lib.gh_dev_trsm_right(l11.data_ptr(), l11.stride(0), dinv.data_ptr(), a21.data_ptr(),
                                                  a21.stride(0), a21.shape[0], a21.shape[1], self._st()))

    def gemm_nt(self, c, a, b):
        """c -= a @ b.T"""
        self.N.check(self.N.lib.gh_dev_gemm(c.data_ptr(), c.stride(0), a.data_ptr(), a.stride(0), b.data_ptr(),
                                            b.stride(0), c.shape[0], c.shape[1], a.shape[1], -1.0, 1.0, 0, self._st()))

    def gemm_nt_stair(self, c, a, b, group_rows, widths):
        """Row group g of c (group_rows rows) -= a[g-th rows] @ b[:widths[g]].T -- one launch for the whole staircase
        (include/george_amd.h gh_dev_gemm_nt_stair); widths non-decreasing, multiples of 128."""
        import ctypes as C
        w = (C.c_int64 * len(widths))(*[int(v) for v in widths])
        self.N.check(self.N.lib.gh_dev_gemm_nt_stair(c.data_ptr(), c.stride(0), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                                     int(group_rows), len(widths), w, a.shape[1], self._st()))

    def gemm(self, c, a, b, alpha=1.0, beta=0.0, a_t=False, b_t=False):
        """c = beta*c + alpha * op(a) @ op(b); all extents multiples of 128 (include/george_amd.h gh_dev_gemm:
        its native operand form is A(m,k) row-major and B(n,k) row-major, i.e. ``a @ b.T``)."""
        m, n = c.shape
        k = a.shape[0] if a_t else a.shape[1]
        flags = (1 if a_t else 0) | (0 if b_t else 2)          # GH_GEMM_A_MMAJOR | GH_GEMM_B_NMAJOR
        self.N.check(self.N.lib.gh_dev_gemm(c.data_ptr(), c.stride(0), a.data_ptr(), a.stride(0), b.data_ptr(),
                                            b.stride(0), m, n, k, float(alpha), float(beta), flags, self._st()))

    def set_kernel(self, kernel_spec):
        """Re-flatten the kernel (hyper-parameters change at every optimiser step; the ops object and
        its streams are kept)."""
        from .program import DeviceKernel
        self.dk = DeviceKernel(kernel_spec)
        if self.dk.ndim != self.ndim:
            raise RuntimeError("dimension mismatch")

    def gemv(self, a, x, y, alpha, beta):
        """y = beta*y + alpha * a @ x"""
        self.N.check(self.N.lib.gh_dev_gemv(a.data_ptr(), a.stride(0), a.shape[0], a.shape[1], 0,
                                            x.data_ptr(), y.data_ptr(), alpha, beta, self._st()))

    def logdet_accum(self, a, out):
        self.N.check(self.N.lib.gh_dev_logdet_accum(a.data_ptr(), a.stride(0), a.shape[0], out.data_ptr(), self._st()))

    def trsv(self, l, dinv, w, z):
        """z = L^-1 w for a factored diagonal tile (one chained launch)."""
        n = l.shape[0]
        if getattr(self, "_trsv_scratch", None) is None or self._trsv_scratch.numel() < n // 128 + 1:
            self._trsv_scratch = self.torch.zeros(n // 128 + 1, dtype=self.torch.int32, device=self.device)
        self.N.check(self.N.lib.gh_dev_trsv_lower(l.data_ptr(), l.stride(0), dinv.data_ptr(), n, w.data_ptr(), z.data_ptr(),
                                                  self._trsv_scratch.data_ptr(), self._st()))
        # the chain's time-out flag sits behind the block flags and is cleared by the next call:
        # fold it into an accumulator on the same stream, read back once per sweep (trsv_failed, all-reduced with the sum)
        if getattr(self, "_trsv_fail", None) is None:
            self._trsv_fail = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        self._trsv_fail += self._trsv_scratch[n // 128:n // 128 + 1]

    def trsv_failed(self):
        """Device counter (1 element, int32) of chained solves that gave up waiting since the last
        clear_trsv_failed() -- their z would be garbage: the kernel sets the flag after a 2 s stall.
        BlockCyclicCholesky.dot_solve all-reduces it together with its sum so that every rank raises."""
        if getattr(self, "_trsv_fail", None) is None:
            self._trsv_fail = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        return self._trsv_fail

    def clear_trsv_failed(self):
        if getattr(self, "_trsv_fail", None) is not None:
            self._trsv_fail.zero_()

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    # stream plumbing for the look-ahead pipeline
    def make_side_stream(self):
        lo, hi = self.torch.cuda.Stream.priority_range()
        return self.torch.cuda.Stream(device=self.device, priority=hi)

    def make_bulk_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def main_stream(self):
        return self.torch.cuda.current_stream(self.device)

    def on(self, stream):
        return self.torch.cuda.stream(stream)

    def event(self, stream, timing=False):
        ev = self.torch.cuda.Event(enable_timing=timing)
        ev.record(stream)
        return ev

    def wait(self, stream, event):
        stream.wait_event(event)

    def pool(self):
        """A few extra streams: independent tile launches (one GEMM per tile column, one build per
        tile) are spread over them so that the tail of one launch overlaps the head of the next."""
        if getattr(self, "_pool", None) is None:
            self._pool = [self.torch.cuda.Stream(device=self.device) for _ in range(3)]
        return self._pool


class BlockCyclicCholesky(object):

    def __init__(self, ops, n, nb=512, rank=None, world=None, lookahead=True, grid=None, snake=True, chain_only=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ops = torch, dist, ops
        self.live = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size() if self.live else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if self.live else 0)
        self.Pr, self.Pc = grid_shape(self.world, grid)
        self.pr, self.pc = divmod(self.rank, self.Pc)
        self.snake = bool(snake) and self.Pc == 1 and self.Pr > 1
        self.chain_only = bool(chain_only)                                               # timing aid: no trailing update but block column k+1
        if nb % 128:
            raise ValueError("nb must be a multiple of 128")
        self.n, self.nb = int(n), int(nb)
        self.nt = -(-self.n // self.nb)
        self.rows = [i for i in range(self.nt) if self.prow(i) == self.pr]
        self.cols = [j for j in range(self.nt) if self.pcol(j) == self.pc]
        self.lrow = {i: li for li, i in enumerate(self.rows)}
        self.lcol = {j: lj for lj, j in enumerate(self.cols)}
        self.lookahead = bool(lookahead) and getattr(ops, "has_streams", False)
        nbk, nlr = self.nb, max(len(self.rows), 1)
        self.A = ops.zeros(nlr * nbk, max(len(self.cols), 1) * nbk)
        self.dinv = ops.zeros(self.nt, nbk // 128, 128, 128)        # inverses of the 128-blocks of every L_kk I see
        self.Lkk = ops.zeros(nbk, nbk)
        # panel workspaces, double-buffered for the look-ahead pipeline
        cmax0 = max([len([j for j in range(1, self.nt) if self.pcol(j) == self.pc and self.prow(j) == mm])
                     for mm in range(self.Pr)] + [1])
        # (padded so that a row panel splits into `world` equal chunks for the all-links exchange)
        self._ws_row_flat = [ops.zeros(-(-(nlr * nbk * nbk) // self.world) * self.world) for _ in range(2)]
        self.ws_row = [f[:nlr * nbk * nbk].view(nlr * nbk, nbk) for f in self._ws_row_flat]
        mode = os.environ.get("GEORGE_AMD_DIST_ROWXCHG", "auto")
        groups = _grid_groups(dist, self.world, self.Pr, self.Pc) if (self.live and self.world > 1) else None
        self.row_a2a = self.live and self.Pc > 1 and (mode == "a2a" or (mode == "auto" and self.Pr > 1))
        if self.row_a2a:
            self.row_a2a = _a2a_capable(dist, ops, groups, self.world, self.rank)
        if self.row_a2a:
            nmax = -(-(self.nt // self.Pr + 1) * nbk * nbk // self.world)       # largest chunk of any process row
            self.ws_relay = [ops.zeros(self.Pr * nmax) for _ in range(2)]
            self.ws_fwd = [ops.zeros(self.world * nmax) for _ in range(2)]
        self.ws_send = [ops.zeros(cmax0, nbk, nbk) for _ in range(2)]
        # gather target: ONE flat buffer per parity, viewed as (Pr, cmax, nb, nb) for the step's cmax -- all_gather_into_tensor then
        # writes the members' tiles in place (the list form of all_gather goes through a temporary and Pr copy kernels per step)
        self.ws_gath = [ops.zeros(self.Pr * cmax0 * nbk * nbk) for _ in range(2)] if self.Pr > 1 else None
        self._gather_into_tensor = hasattr(dist, "all_gather_into_tensor")
        self.ws_next = [ops.zeros(nbk, nbk) for _ in range(2)]      # tile k+1 of the column panel, ahead of the gather
        # the column panel in LOCAL COLUMN ORDER (tile lj of my tile columns at [lj]): the rest of U(k) is one GEMM per local
        # tile row whose B operand is a run of consecutive local columns
        self.ws_colp = [ops.zeros(max(len(self.cols), 1), nbk, nbk) for _ in range(2)]
        self._idx_cache = {}
        self.info = ops.zeros(1, dtype=torch.int64)
        self.logdet_dev = ops.zeros(1)
        self.log_determinant = None
        self.computed = False
        self.profile = False          # record HIP events around every trailing update (bench.py roofline)
        self._upd = []                # (event before, event after, flops launched) per _update call
        self._tl = []                 # (step, [events: start, panel factored, row panel here, column panel here])
        # sub-communicators (created once per process group, _grid_groups)
        self.row_groups, self.col_groups, self.bulk_groups = [None] * self.Pr, [None] * self.Pc, [None] * self.Pc
        if groups is not None:
            self.row_groups, self.col_groups, self.bulk_groups = groups["rows"], groups["cols"], groups["bulk"]

    # -- helpers ---------------------------------------------------------------------------------
    def grank(self, pr, pc):
        return pr * self.Pc + pc

    def prow(self, i):
        """process row of global tile row i (snake order when whole tile rows are dealt: equal lower-triangle shares)"""
        if not self.snake:
            return i % self.Pr
        t = i % (2 * self.Pr)
        return t if t < self.Pr else 2 * self.Pr - 1 - t

    def pcol(self, j):
        return j % self.Pc

    def tile(self, i, j):
        nb = self.nb
        li, lj = self.lrow[i], self.lcol[j]
        return self.A[li * nb:(li + 1) * nb, lj * nb:(lj + 1) * nb]

    def _first_local_row_at_least(self, g):
        """smallest local row index whose global tile row is >= g"""
        for li, i in enumerate(self.rows):
            if i >= g:
                return li
        return len(self.rows)

    def _bcast(self, t, src, group):
        if self.live and group is not None:
            self.dist.broadcast(t, src=src, group=group)

    def _fanout(self, thunks):
        """Run independent tile launches; on the GPU they are spread over a small stream pool,
        fenced against the current stream on both sides."""
        ops = self.ops
        if not getattr(ops, "has_streams", False) or len(thunks) < 2:
            for f in thunks:
                f()
            return
        cur = ops.main_stream()
        pool = ops.pool()[:len(thunks)]
        ev0 = ops.event(cur)
        for st in pool:
            ops.wait(st, ev0)
        for idx, f in enumerate(thunks):
            with ops.on(pool[idx % len(pool)]):
                f()
        for st in pool:
            ops.wait(cur, ops.event(st))

    # -- build: every rank evaluates its own tiles on its own GPU ------------------------------------
    def build(self, x, yerr):
        nb = self.nb
        todo = []
        for i in self.rows:
            for j in self.cols:
                if j <= i:
                    todo.append(lambda i=i, j=j: self.ops.kmat(x, self.n, yerr, i * nb, nb, j * nb, nb, self.tile(i, j)))
        self._fanout(todo)

    # -- row panel over ALL links ---------------------------------------------------------------------
    def _row_exchange(self, k, buf):
        """Every process row r has one rank (in process column k % Pc) holding that row's part of
        panel k, S_r doubles, which the other Pc - 1 ranks of the row need.  A broadcast inside the
        row moves S_r over ONE xGMI link per hop (ring) -- and this transfer sits on the critical
        path of every step.  The node is a full mesh, so use the other rows' GPUs as relays:
          phase 1  each holder scatters its panel in `world` equal chunks, chunk i to rank i;
          phase 2  every rank forwards the chunk it got from row r's holder to the members of row r.
        Both are one all_to_all_single on the world group (uneven splits, zeros where nothing
        moves); every link then carries S_r / world per phase instead of S_r."""
        W, Pr, Pc, nb = self.world, self.Pr, self.Pc, self.nb
        kc = self.pcol(k)
        cnt = [len([i for i in range(k + 1, self.nt) if self.prow(i) == r]) for r in range(Pr)]
        chunk = [-(-(c * nb * nb) // W) for c in cnt]                      # doubles per chunk, per process row
        if max(chunk) == 0:
            return
        holder = [self.grank(r, kc) for r in range(Pr)]
        mine = self._ws_row_flat[buf]
        me, my_r = self.rank, self.pr
        # phase 1: holders scatter
        relay = self.ws_relay[buf][:sum(chunk)]
        in1 = [chunk[my_r]] * W if me == holder[my_r] else [0] * W
        out1 = [chunk[q // Pc] if q == holder[q // Pc] else 0 for q in range(W)]
        src1 = mine[:chunk[my_r] * W] if me == holder[my_r] else mine[:0]
        self.dist.all_to_all_single(relay, src1, output_split_sizes=out1, input_split_sizes=in1)
        # phase 2: forward; the same chunk goes to every member of its row (all_to_all_single sends
        # slices of ONE buffer, so the chunk is replicated per destination: an HBM copy)
        off = [sum(chunk[:r]) for r in range(Pr)]
        in2 = [chunk[d // Pc] if d != holder[d // Pc] else 0 for d in range(W)]
        fwd = self.ws_fwd[buf][:sum(in2)]
        pos = 0
        for d in range(W):
            if in2[d]:
                r = d // Pc
                fwd[pos:pos + in2[d]].copy_(relay[off[r]:off[r] + chunk[r]])
                pos += in2[d]
        if me == holder[my_r]:
            out2, dst2 = [0] * W, mine[:0]
        else:
            out2, dst2 = [chunk[my_r]] * W, mine[:chunk[my_r] * W]
        self.dist.all_to_all_single(dst2, fwd, output_split_sizes=out2, input_split_sizes=in2)

    # -- P(k), chain part: factor panel k, move what block column k+1 needs; returns the panel state ---------------
    def _panel(self, k, buf, mark=None):
        nb, nt, Pr, Pc, pr, pc = self.nb, self.nt, self.Pr, self.Pc, self.pr, self.pc
        ops = self.ops
        nloc_r = len(self.rows)
        kr, kc = self.prow(k), self.pcol(k)
        in_col = (pc == kc)
        timed = self.profile and getattr(ops, "has_streams", False)
        tl = [ops.event(ops.main_stream(), timing=True)] if timed else None
        if pr == kr and in_col:
            akk = self.tile(k, k)
            ops.potrf(akk, self.dinv[k], self.info, k * nb)
            ops.logdet_accum(akk, self.logdet_dev)
            self.Lkk.copy_(akk)
        if k == nt - 1:
            return None
        if in_col and Pr > 1:
            self._bcast(self.Lkk, self.grank(kr, kc), self.col_groups[kc])
            self._bcast(self.dinv[k], self.grank(kr, kc), self.col_groups[kc])
        li0 = self._first_local_row_at_least(k + 1)
        m = (nloc_r - li0) * nb
        wrow = self.ws_row[buf][:m]
        if in_col and m > 0:
            lk = self.lcol[k]
            panel = self.A[li0 * nb:, lk * nb:(lk + 1) * nb]
            ops.trsm(self.Lkk, self.dinv[k], panel)
            wrow.copy_(panel)
        if timed:
            tl.append(ops.event(ops.main_stream(), timing=True))       # potrf + L_kk broadcast + column TRSM
        if self.row_a2a:
            self._row_exchange(k, buf)                 # (world collective: every rank, every step)
        elif Pc > 1 and m > 0:
            self._bcast(wrow, self.grank(pr, kc), self.row_groups[pr])
        # The next panel only waits for block column k+1, whose update needs ONE tile of the column
        # panel, P_{k+1}: send that one ahead (a tile to the Pr - 1 column peers of the process
        # column that owns block column k+1) and let `mark` record "enough for block column k+1";
        # the gather of all the other tiles (_gather) then runs beside that update and the next chain.
        pj_fast = {}
        if pc == self.pcol(k + 1):
            nxt = self.ws_next[buf]
            src_pr = self.prow(k + 1)
            if pr == src_pr:
                s0 = (self.lrow[k + 1] - li0) * nb
                nxt.copy_(wrow[s0:s0 + nb])
            if Pr > 1:
                self._bcast(nxt, self.grank(src_pr, pc), self.col_groups[pc])
            pj_fast[k + 1] = nxt
        if timed:
            tl.append(ops.event(ops.main_stream(), timing=True))       # row panel + tile k+1 have travelled
        if mark is not None:
            mark()
        return [li0, wrow, None, pj_fast, k, buf, tl]

    # -- P(k), bulk part: the column panel -- tiles P_j, j > k + 1, pcol(j) == pc -- gathered inside my process column
    def _gather(self, panel):
        if panel is None:
            return
        li0, wrow, _, _, k, buf, tl = panel
        nb, nt, Pr, pr, pc = self.nb, self.nt, self.Pr, self.pr, self.pc
        ops = self.ops
        pj = {}
        first = k + 2                                   # (block column k+1 is served by the tile that travelled ahead)
        mine = [j for j in range(first, nt) if self.pcol(j) == pc and self.prow(j) == pr]
        cnt = [len([j for j in range(first, nt) if self.pcol(j) == pc and self.prow(j) == mm]) for mm in range(Pr)]
        cmax = max(cnt) if cnt else 0
        if cmax > 0 and not self.chain_only:
            send = self.ws_send[buf][:cmax]
            for t, j in enumerate(mine):
                s = (self.lrow[j] - li0) * nb
                send[t].copy_(wrow[s:s + nb])
            if Pr > 1 and self.live:
                flat = self.ws_gath[buf][:Pr * cmax * nb * nb].view(Pr, cmax, nb, nb)
                gathered = [flat[mm] for mm in range(Pr)]
                if self._gather_into_tensor:
                    try:
                        self.dist.all_gather_into_tensor(flat.view(Pr * cmax, nb, nb), send, group=self.bulk_groups[pc])
                    except (RuntimeError, NotImplementedError, TypeError):     # (a backend without it raises on every rank alike)
                        self._gather_into_tensor = False
                if not self._gather_into_tensor:
                    self.dist.all_gather(gathered, send, group=self.bulk_groups[pc])
            else:
                gathered = [send]
            colp = self.ws_colp[buf]
            for mm in range(Pr):
                js = [j for j in range(first, nt) if self.pcol(j) == pc and self.prow(j) == mm]
                if not js:
                    continue
                key = (k, mm)
                idx = self._idx_cache.get(key)
                if idx is None:
                    idx = self._idx_cache[key] = self.torch.tensor([self.lcol[j] for j in js], dtype=self.torch.int64,
                                                                   device=colp.device)
                colp.index_copy_(0, idx, gathered[mm][:len(js)])
            pj = colp
        panel[2] = pj
        if tl is not None:
            tl.append(ops.event(ops.main_stream(), timing=True))       # column panel gathered
            self._tl.append((k, tl))

    # -- U(k) restricted to the given global tile columns ---------------------------------------------
    def _update(self, panel, cols, fast=False):
        """fast: `cols` is block column k+1 alone, served by the tile that travelled ahead -- one GEMM over all my rows
        below it.  Else every local tile row i takes C[i, first col .. min(i, last col)] -= W_i P^T with P a run of consecutive
        local columns of the re-packed column panel: one staircase launch over all rows where the tile ops offer it (the HIP
        ops), one GEMM per row otherwise (one launch per tile COLUMN -- round 4's first form -- was 64-128 small launches per
        step on a P x 1 grid: 39-59 TFLOP/s per rank, profiles/r04/scale_model.md)."""
        if panel is None or not cols:
            return
        li0, wrow, colp, pj_fast = panel[:4]
        nb, nloc_r = self.nb, len(self.rows)
        todo, flops = [], 0.0
        if fast:
            for j in cols:
                ls = self._first_local_row_at_least(j)
                if ls >= nloc_r:
                    continue
                lj = self.lcol[j]
                flops += 2.0 * (nloc_r - ls) * nb * nb * nb
                todo.append(lambda ls=ls, lj=lj, j=j: self.ops.gemm_nt(
                    self.A[ls * nb:, lj * nb:(lj + 1) * nb], wrow[(ls - li0) * nb:], pj_fast[j]))
        else:
            jlo, jhi = min(cols), max(cols)
            l0 = self.lcol[jlo]
            flat = colp.view(-1, nb)                              # (n_local_cols * nb) x nb
            reach = []                                            # (local row, one past its last local column), rows ascending
            for li in range(self._first_local_row_at_least(jlo), nloc_r):
                i = self.rows[li]
                l1 = len([j for j in self.cols if j <= min(i, jhi)])
                if l1 > l0:
                    reach.append((li, l1))
                    flops += 2.0 * (l1 - l0) * nb * nb * nb
            if reach and getattr(self.ops, "gemm_nt_stair", None) is not None:
                # my tile rows are contiguous in A and in wrow, and each reaches as far as its own diagonal tile: ONE staircase
                # launch (8 launches of ~1.3 chip-fulls each per step at N = 65536 on 8 ranks otherwise: 62 TFLOP/s per rank)
                first = reach[0][0]
                todo.append(lambda first=first, reach=reach: self.ops.gemm_nt_stair(
                    self.A[first * nb:, l0 * nb:], wrow[(first - li0) * nb:], flat[l0 * nb:], nb, [(l1 - l0) * nb for _, l1 in reach]))
            else:
                for li, l1 in reversed(reach):                    # largest first
                    todo.append(lambda li=li, l1=l1: self.ops.gemm_nt(
                        self.A[li * nb:(li + 1) * nb, l0 * nb:l1 * nb], wrow[(li - li0) * nb:(li - li0 + 1) * nb], flat[l0 * nb:l1 * nb]))
        timed = self.profile and todo and getattr(self.ops, "has_streams", False)
        if timed:
            e0 = self.ops.event(self.ops.main_stream(), timing=True)
        self._fanout(todo)
        if timed:
            self._upd.append((e0, self.ops.event(self.ops.main_stream(), timing=True), flops))

    def timeline(self):
        """Per-step milliseconds on THIS rank since the last call (profile = True): the chain
        potrf -> column TRSM ("panel"), row-panel transfer + tile sent ahead ("exchange"), column-panel
        all-gather ("gather"), and the trailing updates ("update", from update_profile's events)."""
        self.ops.sync()
        steps = [{"k": k, "panel_ms": e[0].elapsed_time(e[1]), "exchange_ms": e[1].elapsed_time(e[2]),
                  "gather_ms": e[2].elapsed_time(e[3])} for k, e in self._tl if len(e) == 4]
        self._tl = []
        return {"steps": len(steps), "panel_ms": sum(s["panel_ms"] for s in steps),
                "exchange_ms": sum(s["exchange_ms"] for s in steps), "gather_ms": sum(s["gather_ms"] for s in steps),
                "per_step": steps}

    def update_profile(self):
        """(milliseconds, flops, calls) of the trailing updates recorded since the last call."""
        self.ops.sync()
        ms = sum(a.elapsed_time(b) for a, b, _ in self._upd)
        fl = sum(f for _, _, f in self._upd)
        n = len(self._upd)
        self._upd = []
        return ms, fl, n

    # -- factorisation -----------------------------------------------------------------------------
    def factor(self):
        nt = self.nt
        ops = self.ops
        self.info.zero_()
        self.logdet_dev.zero_()
        if not self.lookahead:
            for k in range(nt):
                panel = self._panel(k, 0)
                self._gather(panel)
                self._update(panel, [j for j in self.cols if j == k + 1], fast=True)
                if not self.chain_only:
                    self._update(panel, [j for j in self.cols if j > k + 1])
        else:
            # three streams: s_side (high priority) the chain of P(k+1), s_bulk the column-panel gather, s_main the updates
            s_main, s_side = ops.main_stream(), ops.make_side_stream()
            s_bulk = ops.make_bulk_stream() if hasattr(ops, "make_bulk_stream") else s_side
            ops.wait(s_side, ops.event(s_main))                    # the build is complete
            ev = {}

            def mark():
                ev["fast"] = ops.event(s_side)

            def chain_and_gather(k, buf):
                with ops.on(s_side):
                    p = self._panel(k, buf, mark)
                fast_k = ev.get("fast")
                if p is None:
                    return p, ops.event(s_side), ops.event(s_side)
                ops.wait(s_bulk, fast_k)
                with ops.on(s_bulk):
                    self._gather(p)
                    gath = ops.event(s_bulk)
                return p, fast_k, gath

            panel, fast, gath = chain_and_gather(0, 0)
            for k in range(nt):
                ops.wait(s_main, fast)                             # row panel + tile k+1 of the column panel are here
                if k == nt - 1:
                    break
                self._update(panel, [j for j in self.cols if j == k + 1], fast=True)      # block column k+1 first
                ops.wait(s_side, ops.event(s_main))
                gath_k = gath
                nxt, fast, gath = chain_and_gather(k + 1, (k + 1) % 2)
                ops.wait(s_main, gath_k)                           # the whole column panel of step k
                if not self.chain_only:
                    self._update(panel, [j for j in self.cols if j > k + 1])       # the rest, under P(k+1)
                panel = nxt
            ops.wait(s_main, ops.event(s_side))
            ops.wait(s_main, ops.event(s_bulk))
        # scalars: log-det and failure flag
        tot = self.logdet_dev.clone()
        info = self.info.clone()
        if self.live and self.world > 1:
            self.dist.all_reduce(tot)
            big = self.torch.where(info > 0, info, self.torch.full_like(info, 2 ** 62))
            self.dist.all_reduce(big, op=self.dist.ReduceOp.MIN)
            info = self.torch.where(big == 2 ** 62, self.torch.zeros_like(big), big)
        bad = int(info.item())
        if self.chain_only:                                        # (a timing run: the numbers mean nothing)
            self.log_determinant = float(tot.item())
            self.computed = False
            return
        if bad != 0:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % bad)
        self.log_determinant = float(tot.item())
        self.computed = True

    # -- r^T K^-1 r = || L^-1 r ||^2 : left-looking forward substitution over tile rows -------------
    def dot_solve(self, y):
        """`y`: full (padded to nt*nb) vector, replicated on every rank."""
        nb, nt, Pr, Pc, pr, pc = self.nb, self.nt, self.Pr, self.Pc, self.pr, self.pc
        ops = self.ops
        zloc = ops.zeros(max(len(self.cols), 1) * nb)         # z tiles for MY tile columns
        part = ops.zeros(nb)
        w = ops.zeros(nb)
        zk = ops.zeros(nb)
        acc = ops.zeros(1)
        for k in range(nt):
            kr, kc = self.prow(k), self.pcol(k)
            if pr == kr:
                cend = len([j for j in self.cols if j < k])
                lk = self.lrow[k]
                if cend > 0:
                    ops.gemv(self.A[lk * nb:(lk + 1) * nb, :cend * nb], zloc[:cend * nb], part, 1.0, 0.0)
                else:
                    part.zero_()
                if Pc > 1 and self.live:
                    self.dist.reduce(part, dst=self.grank(kr, kc), group=self.row_groups[pr])
            if pr == kr and pc == kc:
                w.copy_(y[k * nb:(k + 1) * nb])
                w -= part
                ops.trsv(self.tile(k, k), self.dinv[k], w, zk)    # z_k = L_kk^-1 w, one chained launch
                acc += (zk * zk).sum()
            if pc == kc:
                if Pr > 1 and self.live:
                    self._bcast(zk, self.grank(kr, kc), self.col_groups[kc])
                lk = self.lcol[k]
                zloc[lk * nb:(lk + 1) * nb].copy_(zk)
        # the chained solve's time-out flag travels WITH the sum: a stalled tile on one rank must raise on
        # every rank (a rank that raised alone would leave the others waiting in the next collective)
        fail = ops.trsv_failed() if hasattr(ops, "trsv_failed") else None
        both = self.torch.cat([acc, fail.to(acc.dtype)]) if fail is not None else acc
        if self.live and self.world > 1:
            self.dist.all_reduce(both)
        vals = both.tolist()
        if fail is not None and vals[1] != 0.0:
            ops.clear_trsv_failed()
            raise RuntimeError("george_amd HIP backend failure: forward solve: a workgroup waited more than 2 s for its "
                               "predecessor (on %d tile solves across the group)" % int(vals[1]))
        return float(vals[0])


    # -- K^-1 B, L^-1 B, r @ L^T on the sharded factor ------------------------------------------------
    def _tile_solve(self, k, blk, trans):
        """blk (nb x rp) <- L_kk^-1 blk  (trans: L_kk^-T blk), blocked substitution over the 128-blocks
        of the diagonal tile with their stored inverses -- GEMMs only."""
        ops, nb = self.ops, self.nb
        L, dinv = self.tile(k, k), self.dinv[k]
        nq = nb // 128
        tmp = ops.zeros(128, blk.shape[1])
        order = range(nq) if not trans else range(nq - 1, -1, -1)
        for q in order:
            bq = blk[q * 128:(q + 1) * 128]
            ops.gemm(tmp, dinv[q], bq, 1.0, 0.0, a_t=trans)
            bq.copy_(tmp)
            if not trans and q + 1 < nq:
                ops.gemm(blk[(q + 1) * 128:], L[(q + 1) * 128:, q * 128:(q + 1) * 128], bq, -1.0, 1.0)
            if trans and q > 0:
                ops.gemm(blk[:q * 128], L[q * 128:(q + 1) * 128, :q * 128], bq, -1.0, 1.0, a_t=True)

    def solve(self, B, forward=True, backward=True):
        """B: (nt*nb x rp) right-hand sides, replicated on every rank (rp a multiple of 128); returns
        L^-1 B, L^-T B or K^-1 B = L^-T L^-1 B, replicated.  Tile row k of the forward sweep:
        the ranks of process row k % Pr fold their local tiles L[k, j<k] into a partial sum, reduced
        onto the diagonal owner, which solves with L_kk and broadcasts Z_k to everybody; the backward
        sweep mirrors it with the tile column (partials from process column k % Pc, L[i>k, k]^T X_i).
        (basic.py:72-87: cho_solve = both sweeps.)"""
        nb, nt, Pr, Pc, pr, pc = self.nb, self.nt, self.Pr, self.Pc, self.pr, self.pc
        ops, live = self.ops, self.live and self.world > 1
        X = B.clone()
        rp = X.shape[1]
        part = ops.zeros(nb, rp)
        if forward:
            for k in range(nt):
                kr, kc = self.prow(k), self.pcol(k)
                owner = self.grank(kr, kc)
                xk = X[k * nb:(k + 1) * nb]
                if pr == kr:
                    part.zero_()
                    lk = self.lrow[k]
                    for j in self.cols:
                        if j < k:
                            lj = self.lcol[j]
                            ops.gemm(part, self.A[lk * nb:(lk + 1) * nb, lj * nb:(lj + 1) * nb], X[j * nb:(j + 1) * nb], 1.0, 1.0)
                    if Pc > 1 and live:
                        self.dist.reduce(part, dst=owner, group=self.row_groups[pr])
                    if pc == kc:
                        xk -= part
                        self._tile_solve(k, xk, False)
                if live:
                    self.dist.broadcast(xk, src=owner)
        if backward:
            for k in range(nt - 1, -1, -1):
                kr, kc = self.prow(k), self.pcol(k)
                owner = self.grank(kr, kc)
                xk = X[k * nb:(k + 1) * nb]
                if pc == kc:
                    part.zero_()
                    lk = self.lcol[k]
                    for i in self.rows:
                        if i > k:
                            li = self.lrow[i]
                            ops.gemm(part, self.A[li * nb:(li + 1) * nb, lk * nb:(lk + 1) * nb], X[i * nb:(i + 1) * nb], 1.0, 1.0, a_t=True)
                    if Pr > 1 and live:
                        self.dist.reduce(part, dst=owner, group=self.col_groups[pc])
                    if pr == kr:
                        xk -= part
                        self._tile_solve(k, xk, True)
                if live:
                    self.dist.broadcast(xk, src=owner)
        return X

    def apply_sqrt_t(self, Rt):
        """Rt: (nt*nb x rp) = r^T, replicated; returns (r @ L^T)^T = L @ r^T, replicated
        (basic.py:104-114 with U = L^T).  Tile row k: partial sums of L[k, j<=k] Rt_j over process row
        k % Pr, reduced onto the diagonal owner and broadcast."""
        nb, nt, Pr, Pc, pr, pc = self.nb, self.nt, self.Pr, self.Pc, self.pr, self.pc
        ops, live = self.ops, self.live and self.world > 1
        out = ops.zeros(*Rt.shape)
        part = ops.zeros(nb, Rt.shape[1])
        for k in range(nt):
            kr, kc = self.prow(k), self.pcol(k)
            owner = self.grank(kr, kc)
            if pr == kr:
                part.zero_()
                lk = self.lrow[k]
                for j in self.cols:
                    if j <= k:
                        lj = self.lcol[j]
                        t = self.A[lk * nb:(lk + 1) * nb, lj * nb:(lj + 1) * nb]
                        if j == k:
                            t = self.torch.tril(t)        # (the factorisation leaves the tile's upper 128-blocks untouched)
                        ops.gemm(part, t, Rt[j * nb:(j + 1) * nb], 1.0, 1.0)
                if Pc > 1 and live:
                    self.dist.reduce(part, dst=owner, group=self.row_groups[pr])
                if pc == kc:
                    out[k * nb:(k + 1) * nb].copy_(part)
            if live:
                self.dist.broadcast(out[k * nb:(k + 1) * nb], src=owner)
        return out


class DistributedBasicSolver(object):
    """Solver plugin with the BasicSolver protocol (reference src/george/solvers/basic.py:51-121:
    ``compute`` / ``log_determinant`` / ``computed`` / ``dot_solve`` / ``apply_inverse`` /
    ``get_inverse`` / ``apply_sqrt``), sharded over the process group.
    ``GP(kernel, solver=DistributedBasicSolver, nb=512)`` works unchanged on every rank; inputs and
    results are replicated (every rank passes the same arrays and gets the same answers).

    The block-cyclic workspace (local tiles, panel buffers, streams) is parked when a solver is
    dropped and picked up by the next solver of the same (n, nb, world, device) -- `GP.compute`
    makes a new solver at every optimiser evaluation -- and sub-communicators are created once per
    process group."""

    def __init__(self, kernel, nb=512, device=None, ops=None, lookahead=True, grid=None):
        self.kernel, self.nb = kernel, nb
        self._grid = grid
        self._ops = ops
        self._device = device
        self._lookahead = lookahead
        self.computed = False
        self.log_determinant = None

    def _get_chol(self, n):
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if live else 1
        rank = dist.get_rank() if live else 0
        if self._ops is not None:                       # caller-supplied tile ops (CPU tests): no caching
            return BlockCyclicCholesky(self._ops, n, self.nb, lookahead=self._lookahead, grid=self._grid)
        import torch
        dev = self._device if self._device is not None else torch.cuda.current_device()
        ndim = getattr(self.kernel, "ndim", None)           # (a parked workspace's tile ops are bound to an input dimension)
        key = (n, self.nb, world, rank, dev, self._lookahead, id(dist.group.WORLD) if live else 0, ndim, str(self._grid))
        if getattr(self, "_chol", None) is not None and self._key == key:
            chol = self._chol                           # recompute on the same solver object
        else:
            self._release()
            free = _CHOL_CACHE.get(key)
            chol = free.pop() if free else None
        if chol is None:
            chol = BlockCyclicCholesky(HipTileOps(dev, self.kernel), n, self.nb, lookahead=self._lookahead, grid=self._grid)
        else:
            chol.ops.set_kernel(self.kernel)
            chol.computed = False
        self._key = key
        return chol

    def _release(self):
        """Park the workspace for the next solver of the same shape (at most _CHOL_CACHE_MAX parked
        workspaces in total; the rest is freed)."""
        chol, key = getattr(self, "_chol", None), getattr(self, "_key", None)
        self._chol = None
        if chol is None or key is None or self._ops is not None:
            return
        try:
            if sum(len(v) for v in _CHOL_CACHE.values()) < _CHOL_CACHE_MAX:
                _CHOL_CACHE.setdefault(key, []).append(chol)
        except Exception:
            pass

    def __del__(self):
        self._release()

    def compute(self, x, yerr):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2:
            raise ValueError("x must be (nsamples, ndim)")
        yerr = np.ascontiguousarray(np.zeros(len(x)) + yerr, dtype=np.float64)
        self.computed = False
        self._chol = self._get_chol(len(x))
        ops = self._chol.ops
        if x.shape[1] != ops.ndim:
            raise RuntimeError("dimension mismatch")
        self._n = len(x)
        xd, ed = ops.to_device(x), ops.to_device(yerr)
        self._chol.build(xd, ed)
        self._chol.factor()
        self.log_determinant = self._chol.log_determinant
        self.computed = True

    def _need(self):
        if not self.computed or getattr(self, "_chol", None) is None:
            raise RuntimeError("you must call 'compute' first")
        return self._chol

    def dot_solve(self, y):
        chol = self._need()
        ypad = np.zeros(chol.nt * chol.nb)
        ypad[:self._n] = np.asarray(y, dtype=np.float64).reshape(-1)
        return chol.dot_solve(chol.ops.to_device(ypad))

    def _padded(self, y2):
        """(n, r) host array -> zero-padded (nt*nb, rp) device matrix, rp a multiple of 128"""
        chol = self._chol
        r = y2.shape[1]
        buf = np.zeros((chol.nt * chol.nb, -(-r // 128) * 128))
        buf[:self._n, :r] = y2
        return chol.ops.to_device(buf)

    def apply_inverse(self, y, in_place=False):
        """basic.py:72-87: ``y`` is (n,) or (n, nrhs)."""
        chol = self._need()
        ya = np.asarray(y, dtype=np.float64)
        if ya.ndim < 1 or ya.ndim > 2 or ya.shape[0] != self._n:
            raise ValueError("dimension mismatch")
        y2 = ya.reshape(self._n, -1)
        if y2.shape[1] == 0:
            return ya.copy()
        X = chol.solve(self._padded(y2))
        out = X[:self._n, :y2.shape[1]].cpu().numpy().reshape(ya.shape)
        if in_place and isinstance(y, np.ndarray) and y.dtype == np.float64:
            y[...] = out
            return y
        return np.ascontiguousarray(out)

    def get_inverse(self):
        """basic.py:116-121."""
        return self.apply_inverse(np.eye(self._n), in_place=True)

    def apply_sqrt(self, r):
        """basic.py:104-114: ``r @ U`` with ``U^T U = K`` (U = L^T)."""
        chol = self._need()
        r = np.asarray(r, dtype=np.float64)
        one_d = r.ndim == 1
        r2 = r.reshape(1, -1) if one_d else r
        if r2.shape[1] != self._n:
            raise ValueError("dimension mismatch")
        out = chol.apply_sqrt_t(self._padded(np.ascontiguousarray(r2.T)))
        res = np.ascontiguousarray(out[:self._n, :r2.shape[0]].cpu().numpy().T)
        return res[0] if one_d else res


class DistributedDenseJob(object):
    """bench.py helper: compute() + log_likelihood() of the headline config with device-resident
    inputs, sharded over the launched ranks.  ``kernel(name, amplitude)`` builds the kernel
    (bench.make_kernel); ``ops`` replaces the HIP tile kernels (launcher self-test only)."""

    def __init__(self, n, nb, local_rank, make_inputs, kernel=None, kernel_name="expsquared", ops=None, grid=None):
        import george_amd.kernels as K
        x, yerr, y = make_inputs(n)
        self._amp = float(np.var(y))
        self._mk = kernel if kernel is not None else (lambda name, amp: float(amp) * K.ExpSquaredKernel(1.0))
        spec = self._mk(kernel_name, self._amp)
        self.ops = ops if ops is not None else HipTileOps(local_rank, spec)
        self.n, self.nb = n, (nb or (1024 if n >= 24576 else 512))
        self.chol = BlockCyclicCholesky(self.ops, n, self.nb, grid=grid)
        self.chol.profile = True
        self.x = self.ops.to_device(x[:, None])
        self.yerr = self.ops.to_device(np.sqrt(yerr ** 2 + 1.25e-12))
        ypad = np.zeros(self.chol.nt * self.chol.nb)
        ypad[:n] = y
        self.y = self.ops.to_device(ypad)
        self.ops.sync()

    def set_kernel(self, kernel_name):
        """Another kernel on the same block-cyclic workspace (bench.py: configs[2] after the headline)."""
        spec = self._mk(kernel_name, self._amp)
        if hasattr(self.ops, "set_kernel"):
            self.ops.set_kernel(spec)
        else:
            self.ops.kernel = spec

    def chol_grid(self):
        return self.chol.Pr, self.chol.Pc

    def step(self):
        self.chol.build(self.x, self.yerr)
        self.chol.factor()
        q = self.chol.dot_solve(self.y)
        return -0.5 * (self.n * np.log(2 * np.pi) + self.chol.log_determinant) - 0.5 * q

    def reset_profile(self):
        self.chol.update_profile()
        self.chol.timeline()

    def close(self):
        pass
