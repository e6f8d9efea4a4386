"""N > 1 path on CPU: the block-cyclic driver (george_amd/distributed.py) run with world_size
2, 3, 4 and 8 under the gloo backend, on its default grid (world x 1, snake order) and on 2-D grids.  The tile kernels are replaced by a NumPy stand-in with the same
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


sys.path.insert(0, os.path.join(ROOT, "tests"))
from np_tile_ops import NumpyTileOps  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, nb, q, xchg="auto", grid=None):
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
        solver = DistributedBasicSolver(kernel, nb=nb, ops=NumpyTileOps(kernel), grid=grid)
        solver.compute(x[:, None], 0.1)
        quad = solver.dot_solve(y)
        owners = [solver._chol.prow(i) for i in range(4 * world)]
        # the rest of the solver protocol on the sharded factor (basic.py:72-121)
        Y3 = np.stack([y, np.cos(3 * x), x ** 2], axis=1)
        alpha = solver.apply_inverse(y)
        alpha3 = solver.apply_inverse(Y3)
        sq = solver.apply_sqrt(Y3.T.copy())
        inv = solver.get_inverse() if n <= 900 else None
        # a second solver of the same shape picks up the parked workspace / cached sub-groups
        from george_amd import distributed as D
        ngroups = len(D._GROUP_CACHE)
        again = DistributedBasicSolver(kernel, nb=nb, ops=NumpyTileOps(kernel), grid=grid)
        again.compute(x[:, None], 0.1)
        assert len(D._GROUP_CACHE) == ngroups and again.log_determinant == solver.log_determinant
        # every rank must hold the same scalars
        t = torch.tensor([solver.log_determinant, quad], dtype=torch.float64)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
        # not-positive-definite must surface as LinAlgError on EVERY rank
        bad = DistributedBasicSolver(K.CosineKernel(log_period=0.0), nb=nb, ops=NumpyTileOps(K.CosineKernel(log_period=0.0)), grid=grid)
        raised = False
        try:
            bad.compute(x[:, None], 0.0)
        except np.linalg.LinAlgError:
            raised = True
        assert raised
        if rank == 0:
            q.put((solver.log_determinant, quad, (solver._chol.Pr, solver._chol.Pc, owners), alpha, alpha3, sq, inv))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,xchg,grid", [
    (2, 700, 128, "auto", None), (4, 1100, 128, "auto", None), (8, 1300, 128, "auto", None), (3, 800, 128, "auto", None),     # world x 1, snake
    (2, 900, 256, "a2a", (1, 2)), (4, 513, 128, "bcast", "square"), (4, 1100, 128, "auto", (2, 2)),
    (8, 1300, 128, "auto", (2, 4)), (8, 900, 128, "bcast", (2, 4)), (8, 900, 128, "auto", (4, 2))])
def test_block_cyclic_cholesky_gloo(world, n, nb, xchg, grid):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, q, xchg, grid)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        # (read BEFORE joining: a child that has put large arrays cannot exit until they are consumed)
        logdet, quad, grid_got, alpha, alpha3, sq, inv = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    Pr, Pc, owners = grid_got
    if grid is None:
        assert (Pr, Pc) == (world, 1)                    # whole tile rows per rank, in snake order
        snake = [t if t < world else 2 * world - 1 - t for t in range(2 * world)]
        assert owners == [snake[i % (2 * world)] for i in range(4 * world)]
    else:
        assert (Pr, Pc) == ({4: (2, 2)}[world] if grid == "square" else tuple(grid))
        assert owners == [i % Pr for i in range(4 * world)]
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
    assert grid_shape(1) == (1, 1) and grid_shape(8) == (8, 1) and grid_shape(2) == (2, 1)
    assert grid_shape(8, "square") == (2, 4) and grid_shape(8, "4x2") == (4, 2) and grid_shape(6, (2, 3)) == (2, 3)
    with pytest.raises(ValueError):
        grid_shape(8, (3, 2))
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


def test_staircase_update_and_per_row_update_agree():
    """The trailing update of a rank: one staircase launch over all its tile rows where the tile ops offer gemm_nt_stair,
    one gemm_nt per tile row otherwise -- the same tiles either way, so the same factor bit for bit."""
    sys.path.insert(0, ROOT)
    import george_amd.kernels as K
    from george_amd.distributed import DistributedBasicSolver
    from oracle import kernels_np
    from np_tile_ops import NumpyTileOpsPerRow
    n = 128 * 5 + 37
    x = np.sort(np.random.RandomState(3).uniform(0, 8, n))
    kernel = 1.3 * K.Matern32Kernel(0.9)
    y = np.sin(x)
    got = []
    for ops_cls in (NumpyTileOps, NumpyTileOpsPerRow):
        s = DistributedBasicSolver(kernel, nb=128, ops=ops_cls(kernel))
        s.compute(x[:, None], 0.1)
        got.append((s.log_determinant, s.dot_solve(y), s.apply_inverse(y)))
    assert got[0][0] == got[1][0] and got[0][1] == got[1][1] and np.array_equal(got[0][2], got[1][2])
    Kd = kernels_np.value_symmetric(kernel, x[:, None]) + 0.01 * np.eye(n)
    assert abs(got[0][0] - np.linalg.slogdet(Kd)[1]) < 1e-8 * n
