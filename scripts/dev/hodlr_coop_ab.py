"""HODLR compute()+log_likelihood() at C4 with the cooperative ACA launch given 256 / 192 / 128 / 96 / 64 workgroups
(gh_debug_set_hodlr_coop_wgs), one process.   python scripts/dev/hodlr_coop_ab.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
print("| N | coop workgroups | ms min / median | log-likelihood |\n|---|---|---|---|")
res = {}
for rnd in range(2):
    for w in (256, 192, 128, 96, 64):
        N.lib.gh_debug_set_hodlr_coop_wgs(w)
        job = bench.HodlrJob(n, 0)
        ts = []
        for rep in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
            if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(w, []).extend(ts); res[(w, "ll")] = float(v)
        job.close()
for w in (256, 192, 128, 96, 64):
    print("| %d | %d | %.3f / %.3f | %.15g |" % (n, w, min(res[w]), float(np.median(res[w])), res[(w, "ll")]), flush=True)
N.lib.gh_debug_set_hodlr_coop_wgs(256)
