import sys
sys.path.insert(0, '.')
import bench
def run(n, steps=5, warm=2):
    j = bench.DenseJob(n, 0, 0, profile=False)
    e, ll = bench.run_timed(j, steps, warm, lambda: None)
    j.close()
    return e / steps * 1e3
import os
print("pad", os.environ.get("GEORGE_AMD_STREAM_PAD"), " N=4096 %.2f  N=8192 %.2f  N=16384 %.2f  N=20480 %.2f" % (run(4096), run(8192), run(16384), run(20480)))
