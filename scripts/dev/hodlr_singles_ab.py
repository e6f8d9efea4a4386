"""HODLR compute()+log_likelihood() at C4 with level 5 inside the cooperative ACA launch (gh_debug_set_hodlr_coop_singles) on / off,
one process.   python scripts/dev/hodlr_singles_ab.py [sizes]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [262144, 1048576, 65536]
print("| N | level 5 in the cooperative launch | ms min / median | log-likelihood | ranks |\n|---|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            N.lib.gh_debug_set_hodlr_coop_singles(mode)
            job = bench.HodlrJob(n, 0)
            ts = []
            for rep in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).extend(ts); res[(mode, "ll")] = float(v); res[(mode, "rk")] = hashlib.md5(str(job.ranks()).encode()).hexdigest()[:8]
            job.close()
    for mode in (0, 1):
        print("| %d | %d | %.3f / %.3f | %.15g | %s |" % (n, mode, min(res[mode]), float(np.median(res[mode])), res[(mode, "ll")], res[(mode, "rk")]), flush=True)
N.lib.gh_debug_set_hodlr_coop_singles(1)
