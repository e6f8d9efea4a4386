"""The block-cyclic driver with the REAL tile kernels (HipTileOps -> C ABI) on one MI355X
(degenerate 1x1 grid; the multi-rank communication pattern is covered under gloo in
tests/test_distributed.py).  Checks the device tile path against the single-GPU solver."""
import numpy as np
import pytest

import zoo
from george_amd import kernels, GP, BasicSolver
from george_amd.distributed import DistributedBasicSolver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,nb", [(1500, 256), (2048, 512), (700, 128), (3000, 1024)])
def test_tile_driver_matches_single_gpu_solver(n, nb):
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.Matern32Kernel(1.0)
    a = GP(kernel, solver=BasicSolver)
    a.compute(x, yerr)
    b = GP(kernel, solver=DistributedBasicSolver, nb=nb)
    b.compute(x, yerr)
    assert abs(a.solver.log_determinant - b.solver.log_determinant) <= 1e-10 * abs(a.solver.log_determinant)
    la, lb = a.log_likelihood(y), b.log_likelihood(y)
    assert abs(la - lb) <= 1e-10 * abs(la), (la, lb)


def test_tile_driver_not_positive_definite():
    k = kernels.CosineKernel(log_period=0.0)
    s = DistributedBasicSolver(k, nb=128)
    with pytest.raises(np.linalg.LinAlgError):
        s.compute(np.linspace(0, 3, 400)[:, None], 0.0)


def test_bench_job_single_rank():
    import bench
    from george_amd.distributed import DistributedDenseJob
    job = DistributedDenseJob(4096, 512, 0, bench.make_inputs)
    ll = job.step()
    ref = bench.DenseJob(4096, 512, 0, profile=False)
    assert abs(ll - ref.step()) <= 1e-10 * abs(ll)
    ref.close()


# ---- several ranks sharing the ONE GPU of the test box, gloo as the transport: the real tile kernels,
# ---- streams and look-ahead pipeline on block-cyclic local storage with Pr x Pc > 1 (RCCL refuses two
# ---- ranks on one device, so the collectives themselves are gloo's; the call pattern is the same).
def _shared_gpu_worker(rank, world, port, n, nb, lookahead, q):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import george_amd.kernels as K
        from george_amd.distributed import DistributedBasicSolver
        rng = np.random.RandomState(11)
        x = np.sort(rng.uniform(0, 10, n))
        y = np.sin(x)
        kernel = float(np.var(y)) * K.ExpSquaredKernel(1.0)
        s = DistributedBasicSolver(kernel, nb=nb, device=0, lookahead=lookahead)
        s.compute(x[:, None], 0.1)
        quad = s.dot_solve(y)
        if rank == 0:
            q.put((s.log_determinant, quad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,lookahead", [(2, 1500, 256, True), (2, 1500, 256, False),
                                                 (4, 2500, 256, True), (4, 1100, 128, False),
                                                 (8, 3000, 128, True)])
def test_ranks_sharing_one_gpu_gloo(world, n, nb, lookahead):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, n, nb, lookahead, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    logdet, quad = q.get(timeout=10)
    rng = np.random.RandomState(11)
    x = np.sort(rng.uniform(0, 10, n))
    y = np.sin(x)
    ref = BasicSolver(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
    ref.compute(x[:, None], 0.1 * np.ones(n))
    assert abs(logdet - ref.log_determinant) <= 1e-10 * abs(ref.log_determinant)
    assert abs(quad - ref.dot_solve(y)) <= 1e-9 * abs(quad)
