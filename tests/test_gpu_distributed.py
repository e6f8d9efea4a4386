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
# ---- streams and look-ahead pipeline (chain / bulk gather / updates) on block-cyclic local storage, default world x 1 snake grid
# ---- and 2-D grids (RCCL refuses two
# ---- ranks on one device, so the collectives themselves are gloo's; the call pattern is the same).
def _shared_gpu_worker(rank, world, port, n, nb, lookahead, q, grid=None):
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
        s = DistributedBasicSolver(kernel, nb=nb, device=0, lookahead=lookahead, grid=grid)
        s.compute(x[:, None], 0.1)
        quad = s.dot_solve(y)
        if rank == 0:
            q.put((s.log_determinant, quad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,lookahead,grid", [(2, 1500, 256, True, None), (2, 1500, 256, False, (1, 2)),
                                                      (4, 2500, 256, True, None), (4, 1100, 128, False, (2, 2)),
                                                      (8, 3000, 128, True, None), (8, 3000, 128, True, (2, 4)),
                                                      (4, 2500, 256, True, (2, 2))])
def test_ranks_sharing_one_gpu_gloo(world, n, nb, lookahead, grid):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, n, nb, lookahead, q, grid)) for r in range(world)]
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


# ---- RCCL itself, on the one GPU of the test box: a world of ONE rank with backend "nccl".  The grid
# ---- degenerates to 1x1 (no data moves), but every RCCL entry point the multi-GPU driver uses is
# ---- called for real -- communicator creation with the high-priority-stream options, sub-groups,
# ---- broadcast / all_gather / reduce / all_reduce and the uneven all_to_all_single of the row-panel
# ---- exchange -- on the side stream the look-ahead pipeline issues them from, and the tile driver
# ---- runs end to end under that process group.
_NCCL_SCRIPT = r"""
import os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = sys.argv[2]
import numpy as np, torch, torch.distributed as dist
import zoo
from george_amd import distributed as D, kernels, BasicSolver
torch.cuda.set_device(0)
opts = D.nccl_options()
assert opts is not None and opts.is_high_priority_stream
try:
    dist.init_process_group("nccl", rank=0, world_size=1, pg_options=opts)
except TypeError:
    dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
g = D._new_group(dist, [0])                                   # ProcessGroupNCCL.Options path
groups = D._grid_groups(dist, 1, 1, 1)
assert D._grid_groups(dist, 1, 1, 1) is groups                # cached: no second communicator set
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev, priority=torch.cuda.Stream.priority_range()[1])
t = torch.arange(8, dtype=torch.float64, device=dev)
with torch.cuda.stream(side):
    dist.broadcast(t, src=0, group=g)
    outs = [torch.zeros_like(t)]
    dist.all_gather(outs, t, group=groups["rows"][0])
    dist.reduce(t, dst=0, group=groups["cols"][0])
    dist.all_reduce(t)
    a2a_out = torch.zeros(8, dtype=torch.float64, device=dev)
    dist.all_to_all_single(a2a_out, t, output_split_sizes=[8], input_split_sizes=[8])
    empty = torch.zeros(0, dtype=torch.float64, device=dev)
    dist.all_to_all_single(empty, empty, output_split_sizes=[0], input_split_sizes=[0])   # the probe's empty slot
torch.cuda.synchronize()
assert torch.equal(outs[0], t) and torch.equal(a2a_out, t) and float(t.sum()) == 28.0
n, nb = 3000, 512
x, yerr, y = zoo.bench_data(n)
kernel = np.var(y) * kernels.Matern32Kernel(1.0)
s = D.DistributedBasicSolver(kernel, nb=nb)
s.compute(x[:, None], yerr)
assert s._chol.live and s._chol.world == 1
ref = BasicSolver(kernel); ref.compute(x[:, None], yerr)
assert abs(s.log_determinant - ref.log_determinant) <= 1e-10 * abs(ref.log_determinant)
assert abs(s.dot_solve(y) - ref.dot_solve(y)) <= 1e-9 * abs(ref.dot_solve(y))
assert np.allclose(s.apply_inverse(y), ref.apply_inverse(y), rtol=1e-8, atol=1e-10)
Y = np.stack([y, np.cos(x)], axis=1)
assert np.allclose(s.apply_inverse(Y), ref.apply_inverse(Y), rtol=1e-8, atol=1e-10)
assert np.allclose(s.apply_sqrt(Y.T.copy()), ref.apply_sqrt(Y.T.copy()), rtol=1e-10, atol=1e-12)
s._chol.profile = True
s.compute(x[:, None], yerr)
tl = s._chol.timeline()
assert tl["steps"] == s._chol.nt - 1 and tl["panel_ms"] > 0.0
ms, fl, calls = s._chol.update_profile()
assert calls > 0 and ms > 0.0
D.clear_caches()
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_rccl_world_of_one():
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", _NCCL_SCRIPT, root, str(port)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("n,nb", [(1500, 256), (1100, 128)])
def test_tile_driver_full_protocol(n, nb):
    """apply_inverse / get_inverse / apply_sqrt of the sharded solver with the real tile kernels (1x1 grid)."""
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.Matern32Kernel(1.0)
    ref = BasicSolver(kernel)
    ref.compute(x[:, None], yerr)
    s = DistributedBasicSolver(kernel, nb=nb)
    s.compute(x[:, None], yerr)
    Y = np.stack([y, np.cos(x), x], axis=1)
    assert np.allclose(s.apply_inverse(y), ref.apply_inverse(y), rtol=1e-8, atol=1e-10)
    assert np.allclose(s.apply_inverse(Y), ref.apply_inverse(Y), rtol=1e-8, atol=1e-9)
    assert np.allclose(s.apply_sqrt(Y.T.copy()), ref.apply_sqrt(Y.T.copy()), rtol=1e-10, atol=1e-12)
    if n <= 1200:
        assert np.allclose(s.get_inverse(), ref.get_inverse(), rtol=1e-7, atol=1e-9)
    gp = GP(kernel, solver=DistributedBasicSolver, nb=nb)                 # predict() on the sharded factor (generic path)
    gp.compute(x, yerr)
    gq = GP(kernel)
    gq.compute(x, yerr)
    t = np.linspace(0, 10, 33)
    mu, var = gp.predict(y, t, return_var=True)
    mu0, var0 = gq.predict(y, t, return_var=True)
    assert np.allclose(mu, mu0, rtol=1e-8, atol=1e-10) and np.allclose(var, var0, rtol=1e-6, atol=1e-10)
