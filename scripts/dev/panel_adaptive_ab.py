"""compute()+log_likelihood() (bench.DenseJob, inputs resident) with the adaptive panel width on and off in ONE process
(gh_debug_set_adaptive_panels): per size the best and median step and the log-likelihood.  python scripts/dev/panel_adaptive_ab.py [sizes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
MODES = [int(m) for m in os.environ.get('MODES', '0,1').split(',')]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536, 49152, 32768, 28672]
print("| N | adaptive panels | ms min / median | log-likelihood | rel. to 1024 throughout |\n|---|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(2):
        for mode in MODES:
            N.lib.gh_debug_set_adaptive_panels(mode)
            job = bench.DenseJob(n, 0, 0, profile=False)
            ts = []
            for rep in range(2 + (5 if n <= 32768 else 3)):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 2: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).extend(ts); res[(mode, "ll")] = float(v)
            job.close()
    for mode in MODES:
        print("| %d | %d | %.2f / %.2f | %.15g | %.2e |" % (n, mode, min(res[mode]), float(np.median(res[mode])), res[(mode, "ll")],
                                                        abs(res[(mode, "ll")] - res[(0, "ll")]) / abs(res[(0, "ll")])), flush=True)
N.lib.gh_debug_set_adaptive_panels(1)
