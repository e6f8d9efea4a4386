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
