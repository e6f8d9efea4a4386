"""compute()+log_likelihood() (bench.DenseJob) with the stream-ordering events released at device scope (1) or system scope (0):
python scripts/dev/event_scope_ab.py [sizes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from george_amd import _native as N  # noqa: E402
import torch
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 16384, 32768]
LINK = int(os.environ.get("LINK", "0"))
N.lib.gh_debug_set_panel_link(LINK)
print("| N | device-scope events | ms min / median | log-likelihood |\n|---|---|---|---|")
for n in sizes:
    res = {}
    for rnd in range(2):
        for mode in (0, 1):
            N.lib.gh_debug_set_event_scope(mode)
            job = bench.DenseJob(n, 0, 0, profile=False)
            ts = []
            for rep in range(3 + (12 if n <= 16384 else 5)):
                torch.cuda.synchronize(); t0 = time.perf_counter(); v = job.step(); torch.cuda.synchronize()
                if rep >= 3: ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault(mode, []).extend(ts); res[(mode, "ll")] = float(v)
            job.close()
    for mode in (0, 1):
        print("| %d | %d | %.3f / %.3f | %.15g |" % (n, mode, min(res[mode]), float(np.median(res[mode])), res[(mode, "ll")]), flush=True)
