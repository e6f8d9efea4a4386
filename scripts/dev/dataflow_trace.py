"""Per-task trace of the dataflow factorisation (gh_debug_dflow_trace): one compute() per size with the tracing kernels,
records saved as gpurun_out/dflow_trace_N<n>.npy (uint64 [records, 4], see include/george_amd_debug.h) and a summary
printed: the diagonal worker's phases per link, per-queue task counts and durations, how busy the workers were.
python scripts/dev/dataflow_trace.py [sizes ...];  analysis alone: python scripts/dev/dataflow_trace.py --load file.npy"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def summarise(rec, n=None):
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    kind = (rec[:, 3] & 0xff).astype(int)
    half = ((rec[:, 3] >> 8) & 0xff).astype(int)
    fin = ((rec[:, 3] >> 16) & 0xff).astype(int)
    blk = (rec[:, 3] >> 32).astype(int)
    k0 = ((rec[:, 2] >> 32) & 0xffff).astype(int)
    k1 = ((rec[:, 2] >> 48) & 0xffff).astype(int)
    base = t0.min()
    span = (t1.max() - base) / 100.0
    out = ["records %d, span %.1f us" % (len(rec), span)]
    for k, name in ((8, "diag wait"), (9, "diag multiply+publish"), (10, "diag update"), (11, "diag potf2+publish")):
        m = kind == k
        if m.any():
            d = (t1[m] - t0[m]) / 100.0
            out.append("%-24s n %5d  mean %7.2f  median %7.2f  p90 %7.2f  max %8.2f  sum %9.1f us" %
                       (name, m.sum(), d.mean(), np.median(d), np.percentile(d, 90), d.max(), d.sum()))
    m11 = kind == 11
    if m11.sum() > 2:
        ends = np.sort(t1[m11])
        link = np.diff(ends) / 100.0
        out.append("link (potf2 end to potf2 end): mean %.2f median %.2f p90 %.2f max %.2f us" %
                   (link.mean(), np.median(link), np.percentile(link, 90), link.max()))
    work = kind < 8
    for q, name in ((0, "crit"), (1, "next-block steps"), (2, "hi"), (3, "lo near"), (4, "lo far")):
        m = kind == q
        if not m.any():
            continue
        d = (t1[m] - t0[m]) / 100.0
        steps = np.where(half[m] == 2, 2, 1) * (k1[m] - k0[m] + fin[m])       # half-tile products of 64 x 128 x 128
        out.append("%-19s tasks %7d  mean %7.2f  median %7.2f  p90 %7.2f us   us per 64x128x128 product %.2f   busy %.1f ms" %
                   (name, m.sum(), d.mean(), np.median(d), np.percentile(d, 90), d.sum() / max(1, steps.sum()), d.sum() / 1e3))
    if work.any():
        nb = len(np.unique(blk[work]))
        busy = (t1[work] - t0[work]).sum() / 100.0
        out.append("workers seen %d, busy %.1f %% of workers x span" % (nb, 100.0 * busy / (nb * span)))
        # occupancy over time: busy workers in 20 slices
        edges = np.linspace(base, t1.max(), 21)
        occ = []
        for a, b in zip(edges[:-1], edges[1:]):
            ov = np.clip(np.minimum(t1[work], b) - np.maximum(t0[work], a), 0, None).sum() / (b - a)
            occ.append(ov)
        out.append("busy workers per 5 % slice: " + " ".join("%d" % round(v) for v in occ))
    return "\n".join(out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--load":
        print(summarise(np.load(sys.argv[2])))
        return
    import bench
    from george_amd import _native as N
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 16384]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    N.lib.gh_debug_set_dataflow(1)
    for n in sizes:
        job = bench.DenseJob(n, 0, 0, profile=False)
        job.step()
        cap = 600000
        cnt = C.c_int64(0)
        N.check(N.lib.gh_debug_dflow_trace(cap, None, 0, C.byref(cnt)))
        job.step()
        buf = np.zeros((cap, 4), dtype=np.uint64)
        N.check(N.lib.gh_debug_dflow_trace(0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(cnt)))
        rec = buf[:cnt.value]
        np.save(os.path.join(ROOT, "gpurun_out", "dflow_trace_N%d.npy" % n), rec)
        print("== N = %d" % n)
        print(summarise(rec, n), flush=True)
        job.close()
    N.lib.gh_debug_set_dataflow(-1)


if __name__ == "__main__":
    main()
