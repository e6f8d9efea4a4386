"""HODLR compute()+log_likelihood() (bench.HodlrJob, inputs resident) with the round-5 shared passes switched on and off in ONE
process (gh_debug_set_hodlr_passes: bit 0 the narrow solve, bit 1 the sweep's update+reduce pass): per size and mask the best
and median step, the log-likelihood (relative difference to mask 0) and a hash of the ranks.  python scripts/dev/hodlr_passes_ab.py [sizes]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402


def main():
    import torch
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [262144, 32768, 50000, 1048576]
    reps = int(os.environ.get("AB_REPS", "15"))
    print("| N | mask | ms min / median | log-likelihood | rel. to mask 0 | ranks |\n|---|---|---|---|---|---|")
    for n in sizes:
        job = bench.HodlrJob(n, 0)
        ts = {m: [] for m in (0, 1, 2, 3)}
        ll, rk = {}, {}
        for rnd in range(3):
            for m in (0, 1, 2, 3):
                N.lib.gh_debug_set_hodlr_passes(m)
                for rep in range(2 + reps // 3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    v = job.step()
                    torch.cuda.synchronize()
                    if rep >= 2:
                        ts[m].append((time.perf_counter() - t0) * 1e3)
                ll[m] = float(v)
                rk[m] = hashlib.md5(str(job.ranks()).encode()).hexdigest()[:8]
        for m in (0, 1, 2, 3):
            print("| %d | %d | %.3f / %.3f | %.12g | %.2e | %s |" % (n, m, min(ts[m]), float(np.median(ts[m])), ll[m], abs(ll[m] - ll[0]) / abs(ll[0]), rk[m]), flush=True)
        job.close()
    N.lib.gh_debug_set_hodlr_passes(-1)


if __name__ == "__main__":
    main()
