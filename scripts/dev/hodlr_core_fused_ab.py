"""HODLR compute()+log_likelihood() with the one-workgroup ACA launch in tree order (0) / fused core launch by the previous compute()'s
durations (1): gh_debug_set_hodlr_core_fused, one process.   python scripts/dev/hodlr_lpt_ab.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:]] or [262144]
print("| N | fused core launch | ms min / median | log-likelihood |\n|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(2):
        for w in (0, 1):
            N.lib.gh_debug_set_hodlr_core_fused(w)
            job = bench.HodlrJob(n, 0)
            ts = []
            for rep in range(13):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 4: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(w, []).extend(ts); res[(w, "ll")] = float(v)
            job.close()
    for w in (0, 1):
        print("| %d | %d | %.3f / %.3f | %.15g |" % (n, w, min(res[w]), float(np.median(res[w])), res[(w, "ll")]), flush=True)
N.lib.gh_debug_set_hodlr_core_fused(1)
