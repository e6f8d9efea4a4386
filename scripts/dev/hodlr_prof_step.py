"""five steps of bench.HodlrJob(N) for rocprofv3 --kernel-trace --stats (mask: gh_debug_set_hodlr_passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
job = bench.HodlrJob(n, 0)
for _ in range(8):
    v = job.step()
print(v)
job.close()
