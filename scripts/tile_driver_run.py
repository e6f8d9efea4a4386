import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from george_amd.distributed import DistributedDenseJob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
job = DistributedDenseJob(n, 0, 0, bench.make_inputs)
job.step(); torch.cuda.synchronize()
t = time.perf_counter(); ll = job.step(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("tile-driver 1 rank n=%d: %.1f ms  %.2f TF  ll=%.6f" % (n, dt * 1e3, bench.flops_alg(n) / dt * 1e-12, ll))
