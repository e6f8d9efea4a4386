"""A/B of the GEMM kernels in ONE process: the tree's libgeorge_amd.so (column "b64") against a variant built as
george_amd/csrc/libgeorge_amd_c.so (column "b128"; the names are those of the first use, scripts/dev/arms/gemm_pair_ab_r04.py).
SYRK-shaped and rectangular launches of the factorisation's sizes, alternating the two libraries, plus a numerical check of the
variant against NumPy.  Used for profiles/r04/gemm_dma_addr_ab.md (s_setprio) and gemm_tile_order_ab.md."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = {"b64": os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd.so"), "b128": os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd_c.so")}
libs = {}
for k, p in LIBS.items():
    l = C.CDLL(p)
    l.gh_dev_gemm.restype = C.c_int
    l.gh_dev_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                              C.c_double, C.c_double, C.c_int32, C.c_void_p]
    libs[k] = l


def gemm(lib, c, a, b, m, n, k, flags):
    rc = lib.gh_dev_gemm(c.data_ptr(), c.stride(0), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), m, n, k, -1.0, 1.0, flags, None)
    assert rc == 0


# numerics of the new variant
rng = np.random.RandomState(0)
for (m, n, k, fl) in [(256, 384, 272, 0), (640, 640, 1024, 4), (128, 128, 128, 0), (1024, 128, 128, 0)]:
    A = rng.randn(m, k); B = A if fl & 4 else rng.randn(n, k); C0 = rng.randn(m, n)
    a, b, c = [torch.from_numpy(v).cuda() for v in (A, B, C0)]
    gemm(libs["b128"], c, a, a if fl & 4 else b, m, n, k, fl)
    torch.cuda.synchronize()
    got, want = c.cpu().numpy(), C0 - A @ B.T
    mask = np.kron(np.tril(np.ones((m // 128, n // 128))), np.ones((128, 128))).astype(bool) if fl & 4 else np.ones((m, n), bool)
    print("check m=%d n=%d k=%d lower=%d: max err %.2e" % (m, n, k, bool(fl & 4), np.abs(got - want)[mask].max()))
    assert np.abs(got - want)[mask].max() < 1e-10

shapes = [(32768, 32768, 1024, 4), (65536, 65536, 1024, 4), (16384, 16384, 1024, 4), (16384, 16384, 4096, 0), (8192, 8192, 1024, 0),
          (14336, 1024, 1024, 0), (2048, 1024, 128, 0)]
if len(sys.argv) > 1:                                       # shapes as MxNxKxFLAGS (FLAGS: 4 = lower)
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
print("| shape | b64 ms (TFLOP/s) | b128 ms (TFLOP/s) | b128 / b64 |\n|---|---|---|---|")
for (m, n, k, fl) in shapes:
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    b = a if fl & 4 else torch.randn(n, k, dtype=torch.float64, device="cuda")
    c = torch.randn(m, n, dtype=torch.float64, device="cuda")
    tiles = (m // 128) * (m // 128 + 1) / 2 if fl & 4 else (m // 128) * (n // 128)
    best = {"b64": 1e30, "b128": 1e30}
    for rnd in range(3):
        for name in ("b64", "b128"):
            lib = libs[name]
            gemm(lib, c, a, b, m, n, k, fl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 if m >= 32768 else 8
            e0.record()
            for _ in range(reps):
                gemm(lib, c, a, b, m, n, k, fl)
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / reps)
    tf = {q: tiles * 2 * 128 * 128 * k / best[q] * 1e-9 for q in best}
    print("| M=%d N=%d K=%d %s | %.3f (%.2f) | %.3f (%.2f) | %.4f |" % (m, n, k, "lower" if fl & 4 else "full", best["b64"], tf["b64"], best["b128"], tf["b128"],
                                                                       best["b128"] / best["b64"]), flush=True)
