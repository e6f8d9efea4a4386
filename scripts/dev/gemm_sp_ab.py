"""A/B of the k-major x k-major GEMM kernel in ONE process and ONE library: gemm_f64_mfma_dma (end-of-slab barrier, then the
slab's 16 ds_read_b128, then its 64 matrix instructions) against gemm_f64_mfma_dma_sp (barrier in the middle of a slab, each
half of the fragment reads issued under the other half's matrix instructions), switched with gh_debug_set_gemm_sp.  First the
bits: every shape's result must be IDENTICAL in the two modes.  Then SYRK-shaped and rectangular launches of the
factorisation's sizes, modes alternated, best of 3 rounds.  python scripts/dev/gemm_sp_ab.py [quick]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from george_amd import _native as N  # noqa: E402

lib = N.lib


def gemm(c, a, b, m, n, k, flags, alpha=-1.0, beta=1.0):
    dp = C.POINTER(C.c_double)
    rc = lib.gh_dev_gemm(C.cast(c.data_ptr(), dp), c.stride(0), C.cast(a.data_ptr(), dp), a.stride(0), C.cast(b.data_ptr(), dp), b.stride(0),
                         m, n, k, alpha, beta, flags, None)
    assert rc == 0, N.last_error()


def main():
    quick = "quick" in sys.argv
    rng = np.random.RandomState(0)
    ok = True
    for (m, n, k, fl, alpha, beta) in [(256, 384, 32, 0, -1.0, 1.0), (640, 640, 1024, 4, -1.0, 1.0), (128, 128, 64, 0, 1.0, 0.0),
                                       (1024, 128, 128, 0, -1.0, 1.0), (2048, 2048, 96, 4, -1.0, 1.0), (4096, 1024, 1024, 0, 1.0, 0.0),
                                       (8192, 8192, 1024, 4, -1.0, 1.0), (512, 640, 4096, 0, 2.5, -0.5)]:
        A = rng.randn(m, k)
        B = A if fl & 4 else rng.randn(n, k)
        C0 = rng.randn(m, n)
        outs = []
        for mode in (0, 1):
            lib.gh_debug_set_gemm_sp(mode)
            a, b, c = [torch.from_numpy(v).cuda() for v in (A, B, C0)]
            gemm(c, a, a if fl & 4 else b, m, n, k, fl, alpha, beta)
            torch.cuda.synchronize()
            outs.append(c.cpu().numpy())
        mask = np.kron(np.tril(np.ones((m // 128, n // 128))), np.ones((128, 128))).astype(bool) if fl & 4 else np.ones((m, n), bool)
        same = np.array_equal(outs[0][mask], outs[1][mask])
        err = np.abs(outs[1] - (beta * C0 + alpha * (A @ B.T)))[mask].max() / max(1.0, np.abs(A @ B.T).max())
        print("bits m=%d n=%d k=%d lower=%d alpha=%g beta=%g: identical %s, error vs NumPy %.1e" % (m, n, k, bool(fl & 4), alpha, beta, same, err), flush=True)
        ok = ok and same and err < 1e-13
    assert ok
    shapes = [(32768, 32768, 1024, 4), (65536, 65536, 1024, 4), (16384, 16384, 1024, 4), (16384, 16384, 4096, 0), (8192, 8192, 1024, 0),
              (14336, 1024, 1024, 0), (8192, 1024, 1024, 0)]
    if quick:
        shapes = shapes[:1] + shapes[2:4]
    print("| shape | end-of-slab barrier ms (TFLOP/s) | half-slab pipelined ms (TFLOP/s) | ratio |\n|---|---|---|---|")
    for (m, n, k, fl) in shapes:
        torch.manual_seed(0)
        a = torch.randn(m, k, dtype=torch.float64, device="cuda")
        b = a if fl & 4 else torch.randn(n, k, dtype=torch.float64, device="cuda")
        c = torch.randn(m, n, dtype=torch.float64, device="cuda")
        tiles = (m // 128) * (m // 128 + 1) / 2 if fl & 4 else (m // 128) * (n // 128)
        best = {0: 1e30, 1: 1e30}
        for rnd in range(3):
            for mode in (0, 1):
                lib.gh_debug_set_gemm_sp(mode)
                gemm(c, a, b, m, n, k, fl)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3 if m >= 32768 else 8
                e0.record()
                for _ in range(reps):
                    gemm(c, a, b, m, n, k, fl)
                e1.record()
                torch.cuda.synchronize()
                best[mode] = min(best[mode], e0.elapsed_time(e1) / reps)
        tf = {q: tiles * 2 * 128 * 128 * k / best[q] * 1e-9 for q in best}
        print("| M=%d N=%d K=%d %s | %.3f (%.2f) | %.3f (%.2f) | %.4f |" % (m, n, k, "lower" if fl & 4 else "full", best[0], tf[0], best[1], tf[1],
                                                                           best[1] / best[0]), flush=True)
    lib.gh_debug_set_gemm_sp(-1)


if __name__ == "__main__":
    main()
