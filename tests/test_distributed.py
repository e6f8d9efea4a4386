"""N > 1 path on CPU: the 2-D block-cyclic driver (george_amd/distributed.py) run with world_size
2 and 4 under the gloo backend.  The tile kernels are replaced by a NumPy stand-in with the same
interface (test infrastructure; the product path uses HipTileOps), so what is exercised here is
tile ownership, the broadcast / all-gather / reduce pattern and the loop order -- against a dense
NumPy Cholesky of the same matrix."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyTileOps(object):
    """CPU stand-in for HipTileOps: same methods, torch CPU tensors, NumPy/oracle arithmetic."""

    def __init__(self, kernel_spec):
        self.kernel = kernel_spec
        self.ndim = kernel_spec.ndim

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or torch.float64)

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def kmat(self, x, n, yerr, row0, nrows, col0, ncols, out):
        from oracle import kernels_np
        X, e = x.numpy(), yerr.numpy()
        blk = np.zeros((nrows, ncols))
        vr, vc = max(0, min(nrows, n - row0)), max(0, min(ncols, n - col0))
        if vr and vc:
            blk[:vr, :vc] = kernels_np.value_general(self.kernel, X[row0:row0 + vr], X[col0:col0 + vc])
        for r in range(nrows):
            c = row0 + r - col0
            if 0 <= c < ncols:
                blk[r, c] = blk[r, c] + e[row0 + r] ** 2 if row0 + r < n else 1.0
        out.copy_(torch.from_numpy(blk))

    def potrf(self, a, dinv, info, base):
        A = np.tril(a.numpy()) + np.tril(a.numpy(), -1).T
        try:
            L = np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            if int(info.item()) == 0:
                info.fill_(base + 1)
            return
        a.copy_(torch.from_numpy(L))
        for b in range(a.shape[0] // 128):
            dinv[b].copy_(torch.from_numpy(np.linalg.inv(L[128 * b:128 * b + 128, 128 * b:128 * b + 128])))

    def trsm(self, l11, dinv, a21):
        a21.copy_(torch.from_numpy(np.linalg.solve(np.tril(l11.numpy()), a21.numpy().T).T))

    def gemm_nt(self, c, a, b):
        c -= a @ b.T

    def gemm(self, c, a, b, alpha=1.0, beta=0.0, a_t=False, b_t=False):
        A = a.T if a_t else a
        B = b.T if b_t else b
        c.copy_(beta * c + alpha * (A @ B) if beta != 0.0 else alpha * (A @ B))

    def gemv(self, a, x, y, alpha, beta):
        y.copy_(beta * y + alpha * (a @ x) if beta != 0.0 else alpha * (a @ x))

    def logdet_accum(self, a, out):
        out += 2.0 * torch.log(torch.diagonal(a)).sum()

    def trsv(self, l, dinv, w, z):
        import scipy.linalg
        z.copy_(torch.from_numpy(scipy.linalg.solve_triangular(np.tril(l.numpy()), w.numpy(), lower=True)))

    def sync(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, nb, q, xchg="auto"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GEORGE_AMD_DIST_ROWXCHG"] = xchg          # row-panel transport: broadcast in the row / all-links exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import george_amd.kernels as K
        from george_amd.distributed import BlockCyclicCholesky, DistributedBasicSolver, grid_shape
        rng = np.random.RandomState(7)
        x = np.sort(rng.uniform(0, 10, n))
        y = np.sin(x)
        kernel = 0.5 * K.Matern32Kernel(1.3)
        solver = DistributedBasicSolver(kernel, nb=nb, ops=NumpyTileOps(kernel))
        solver.compute(x[:, None], 0.1)
        quad = solver.dot_solve(y)
        # the rest of the solver protocol on the sharded factor (basic.py:72-121)
        Y3 = np.stack([y, np.cos(3 * x), x ** 2], axis=1)
        alpha = solver.apply_inverse(y)
        alpha3 = solver.apply_inverse(Y3)
        sq = solver.apply_sqrt(Y3.T.copy())
        inv = solver.get_inverse() if n <= 900 else None
        # a second solver of the same shape picks up the parked workspace / cached sub-groups
        from george_amd import distributed as D
        ngroups = len(D._GROUP_CACHE)
        again = DistributedBasicSolver(kernel, nb=nb, ops=NumpyTileOps(kernel))
        again.compute(x[:, None], 0.1)
        assert len(D._GROUP_CACHE) == ngroups and again.log_determinant == solver.log_determinant
        # every rank must hold the same scalars
        t = torch.tensor([solver.log_determinant, quad], dtype=torch.float64)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
        # not-positive-definite must surface as LinAlgError on EVERY rank
        bad = DistributedBasicSolver(K.CosineKernel(log_period=0.0), nb=nb, ops=NumpyTileOps(K.CosineKernel(log_period=0.0)))
        raised = False
        try:
            bad.compute(x[:, None], 0.0)
        except np.linalg.LinAlgError:
            raised = True
        assert raised
        if rank == 0:
            q.put((solver.log_determinant, quad, grid_shape(world), alpha, alpha3, sq, inv))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,xchg", [(2, 700, 128, "auto"), (4, 1100, 128, "auto"), (2, 900, 256, "a2a"),
                                             (4, 513, 128, "bcast"), (8, 1300, 128, "auto"), (8, 900, 128, "bcast")])
def test_block_cyclic_cholesky_gloo(world, n, nb, xchg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, q, xchg)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        # (read BEFORE joining: a child that has put large arrays cannot exit until they are consumed)
        logdet, quad, grid, alpha, alpha3, sq, inv = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert grid == {2: (1, 2), 4: (2, 2), 8: (2, 4)}[world]
    # dense reference
    sys.path.insert(0, ROOT)
    import george_amd.kernels as K
    from oracle import kernels_np
    rng = np.random.RandomState(7)
    x = np.sort(rng.uniform(0, 10, n))
    y = np.sin(x)
    kernel = 0.5 * K.Matern32Kernel(1.3)
    Kd = kernels_np.value_symmetric(kernel, x[:, None]) + 0.01 * np.eye(n)
    assert abs(logdet - np.linalg.slogdet(Kd)[1]) < 1e-8 * n
    assert abs(quad - y @ np.linalg.solve(Kd, y)) < 1e-8 * abs(quad)
    Y3 = np.stack([y, np.cos(3 * x), x ** 2], axis=1)
    assert alpha.shape == (n,) and np.allclose(alpha, np.linalg.solve(Kd, y), rtol=1e-8, atol=1e-10)
    assert alpha3.shape == (n, 3) and np.allclose(alpha3, np.linalg.solve(Kd, Y3), rtol=1e-8, atol=1e-9)
    U = np.linalg.cholesky(Kd).T                                     # basic.py:114: r @ U, U^T U = K
    assert sq.shape == (3, n) and np.allclose(sq, Y3.T @ U, rtol=1e-9, atol=1e-10)
    if inv is not None:
        assert np.allclose(inv, np.linalg.inv(Kd), rtol=1e-7, atol=1e-9)


def test_single_process_degenerate_grid():
    """world == 1, no process group: the same driver must run without any collective."""
    sys.path.insert(0, ROOT)
    import george_amd.kernels as K
    from george_amd.distributed import DistributedBasicSolver, grid_shape
    from oracle import kernels_np
    assert grid_shape(1) == (1, 1) and grid_shape(8) == (2, 4) and grid_shape(2) == (1, 2)
    n = 300
    x = np.linspace(0, 5, n)
    kernel = 1.0 * K.ExpSquaredKernel(0.7)
    s = DistributedBasicSolver(kernel, nb=128, ops=NumpyTileOps(kernel))
    s.compute(x[:, None], 0.2)
    Kd = kernels_np.value_symmetric(kernel, x[:, None]) + 0.04 * np.eye(n)
    assert abs(s.log_determinant - np.linalg.slogdet(Kd)[1]) < 1e-8 * n
    y = np.cos(x)
    assert abs(s.dot_solve(y) - y @ np.linalg.solve(Kd, y)) < 1e-8
    assert np.allclose(s.apply_inverse(y), np.linalg.solve(Kd, y), rtol=1e-9, atol=1e-11)
    assert np.allclose(s.apply_sqrt(y), y @ np.linalg.cholesky(Kd).T, rtol=1e-10, atol=1e-12)
    with pytest.raises(RuntimeError):
        DistributedBasicSolver(kernel, nb=128, ops=NumpyTileOps(kernel)).apply_inverse(y)
