"""HODLR C4 with the clusters below the root given all / half / a quarter of their workgroups (gh_debug_set_hodlr_coop_lower)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:]] or [262144]
f = N.lib.gh_debug_set_hodlr_coop_lower
print("| N | lower clusters: 1 / this of the even-load width | ms min / median | log-likelihood |\n|---|---|---|---|")
for n in sizes:
  res = {}
  for rnd in range(2):
    for w in (1, 2, 4):
        f(w)
        job = bench.HodlrJob(n, 0)
        ts = []
        for rep in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
            if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(w, []).extend(ts); res[(w, "ll")] = float(v)
        job.close()
  for w in (1, 2, 4):
    print("| %d | %d | %.3f / %.3f | %.15g |" % (n, w, min(res[w]), float(np.median(res[w])), res[(w, "ll")]), flush=True)
f(2)
