"""compute()+log_likelihood() (bench.DenseJob, inputs resident) with the one-launch chain link on and off in ONE process
(gh_debug_set_panel_link): per size the best and median step and whether the log-likelihoods agree.  python scripts/dev/panel_link_ab.py [sizes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 2048, 4096, 8192, 12288, 16384, 24064]
print("| N | link launch | ms min / median | log-likelihood | same bits |\n|---|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(2):
        for mode in (0, 1):
            N.lib.gh_debug_set_panel_link(mode)
            job = bench.DenseJob(n, 0, 0, profile=False)
            ts = []
            for rep in range(3 + (12 if n <= 16384 else 5)):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).extend(ts); res[(mode, "ll")] = float(v)
            job.close()
    for mode in (0, 1):
        print("| %d | %d | %.3f / %.3f | %.15g | %s |" % (n, mode, min(res[mode]), float(np.median(res[mode])), res[(mode, "ll")],
                                                       res[(mode, "ll")] == res[(0, "ll")]), flush=True)
N.lib.gh_debug_set_panel_link(1)
