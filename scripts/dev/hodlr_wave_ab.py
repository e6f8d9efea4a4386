"""HODLR compute()+log_likelihood() (bench.HodlrJob, inputs resident) with the wavefront-per-node ACA of the deep levels on and off
in ONE process (gh_debug_set_hodlr_wave_aca): per size the best and median step, the log-likelihood (must be IDENTICAL) and a hash
of the ranks.   python scripts/dev/hodlr_wave_ab.py [sizes]"""
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
    print("| N | wave ACA | ms min / median | log-likelihood | same bits | ranks |\n|---|---|---|---|---|---|")
    for n in sizes:
        job = bench.HodlrJob(n, 0)
        ts = {m: [] for m in (0, 1)}
        ll, rk = {}, {}
        for rnd in range(3):
            for m in (0, 1):
                N.lib.gh_debug_set_hodlr_wave_aca(m)
                for rep in range(2 + reps // 3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    v = job.step()
                    torch.cuda.synchronize()
                    if rep >= 2:
                        ts[m].append((time.perf_counter() - t0) * 1e3)
                ll[m] = float(v)
                rk[m] = hashlib.md5(str(job.ranks()).encode()).hexdigest()[:8]
        for m in (0, 1):
            print("| %d | %d | %.3f / %.3f | %.15g | %s | %s |" % (n, m, min(ts[m]), float(np.median(ts[m])), ll[m], ll[m] == ll[0], rk[m]), flush=True)
        job.close()
    N.lib.gh_debug_set_hodlr_wave_aca(1)


if __name__ == "__main__":
    main()
