"""compute()+log_likelihood() with the launch chain (arm 0) and the dataflow factorisation (arm 1), same process, same
handle type, device-resident inputs (bench.DenseJob): per size the best and median of `reps` steps of each arm, arms
interleaved, and whether the two log-likelihoods are the same bits.  python scripts/dev/dataflow_ab.py [sizes ...]"""
import json
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
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 2048, 4096, 8192, 12288, 16384, 20480]
    reps = int(os.environ.get("AB_REPS", "9"))
    rows = []
    for n in sizes:
        jobs = {}
        for arm in (0, 1):
            N.lib.gh_debug_set_dataflow(arm)
            jobs[arm] = bench.DenseJob(n, 0, 0, profile=False)
        ts, vals = {0: [], 1: []}, {}
        for rep in range(reps + 2):
            for arm in (0, 1):
                N.lib.gh_debug_set_dataflow(arm)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                v = jobs[arm].step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                if rep >= 2:
                    ts[arm].append(dt)
                vals.setdefault(arm, set()).add(float(v))
        for j in jobs.values():
            j.close()
        fl = n ** 3 / 3.0 + 2.0 * n * n
        row = {"n": n, "chain_ms_min": min(ts[0]), "chain_ms_med": float(np.median(ts[0])),
               "dataflow_ms_min": min(ts[1]), "dataflow_ms_med": float(np.median(ts[1])),
               "dataflow_tflops": fl / (min(ts[1]) * 1e-3) * 1e-12, "chain_tflops": fl / (min(ts[0]) * 1e-3) * 1e-12,
               "same_bits": vals[0] == vals[1] and len(vals[0]) == 1, "values": sorted(vals[0] | vals[1])}
        rows.append(row)
        print(json.dumps(row), flush=True)
    N.lib.gh_debug_set_dataflow(-1)
    print("| N | launch chain ms (min / median) | dataflow ms (min / median) | ratio | same bits |")
    print("|---|---|---|---|---|")
    for r in rows:
        print("| %d | %.3f / %.3f | %.3f / %.3f | %.3f | %s |" % (r["n"], r["chain_ms_min"], r["chain_ms_med"], r["dataflow_ms_min"],
                                                               r["dataflow_ms_med"], r["dataflow_ms_min"] / r["chain_ms_min"], r["same_bits"]))


if __name__ == "__main__":
    main()
