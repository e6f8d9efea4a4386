"""Does a config measured AFTER the N=65536 job in the same process run as fast as before it?"""
import sys, time
sys.path.insert(0, '.')
import bench
import ctypes as C
def overlap(j):
    out = (C.c_double * 36)()
    j.N.check(j.N.lib.gh_debug_stream_overlap(j.h, out, 36))
    names = ["null", "main", "chain", "rows", "near", "masked"]
    return " ".join("%s|%s=%.2f" % (names[a], names[b], out[a * 6 + b]) for a in range(6) for b in range(a + 1, 6))
def run(n, steps=5, warm=2, prof=True):
    j = bench.DenseJob(n, 0, 0, profile=prof)
    e, ll = bench.run_timed(j, steps, warm, lambda: None)
    print("   overlap:", overlap(j))
    p = j.profile() if prof else None
    j.close()
    s = "%.2f ms" % (e / steps * 1e3)
    if p is not None:
        s += "  (last step: total %.2f build %.2f panel %.2f trailing %.2f [%d launches] solve %.2f union %.2f)" % (
            p.ms_total, p.ms_build, p.ms_panel, p.ms_trailing, p.n_trailing, p.ms_solve, p.ms_update_union)
    return s
import os
mode = sys.argv[1]
print("mode", mode)
print("16384 first:", run(16384, prof=False))
if mode == "a":
    print("C4: %.2f ms" % (bench.hodlr_report(262144, 0, cpu_n=0)["seconds_per_step"] * 1e3))
elif mode == "b":
    print("65536 prof off:", run(65536, 2, 1, prof=False))
elif mode == "c":
    print("32768 prof off:", run(32768, 2, 1, prof=False))
elif mode == "d":
    print("4096 prof off:", run(4096, 2, 1, prof=False))
elif mode[0] == "s":
    import torch
    k = int(mode[1:])
    ss = [torch.cuda.Stream() for _ in range(k)]
    for q in ss:
        with torch.cuda.stream(q):
            torch.zeros(16, device="cuda").sum().item()
    print("created", k, "torch streams (kept alive)")
elif mode == "h":
    print("small HODLR: %.2f ms" % (bench.hodlr_report(8192, 0, steps=2, warmup=1, cpu_n=0)["seconds_per_step"] * 1e3))
print("16384 after:", run(16384, prof=False))
