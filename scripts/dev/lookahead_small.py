"""Does the look-ahead pay below N = 4096?  compute()+log_likelihood() with and without it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
for n in [int(a) for a in sys.argv[1:]] or [1536, 2048, 3072, 4096, 6144]:
    r = []
    for la in (True, False):
        job = bench.DenseJob(n, 0, 0, profile=False, lookahead=la)
        el, ll = bench.run_timed(job, 20, 5, lambda: None)
        job.close()
        r.append(el / 20 * 1e3)
    print("N=%5d  look-ahead %.3f ms   single stream %.3f ms" % (n, r[0], r[1]), flush=True)
