"""A/B of the 256 x 128-tile GEMM kernel (gemm_f64_mfma_dma_tall) against the 128 x 128 one on the
shapes of the factorisation's trailing updates: bit-identical results (same K order per element),
time per launch, TFLOP/s.   python scripts/gemm_tall_ab.py [quick]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from george_amd import _native as N


def run(m, n, k, lower, tall, reps=3, keep=None):
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    b = a if lower else torch.randn(n, k, dtype=torch.float64, device="cuda")
    c = torch.randn(m, n, dtype=torch.float64, device="cuda")
    prev = N.lib.gh_debug_set_gemm_tall(tall)
    try:
        def go():
            N.check(N.lib.gh_dev_gemm(c.data_ptr(), n, a.data_ptr(), k, b.data_ptr(), k, m, n, k, -1.0, 1.0, 4 if lower else 0, None))
        go(); torch.cuda.synchronize()
        first = c.clone() if keep else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): go()
        e1.record(); torch.cuda.synchronize()
    finally:
        N.lib.gh_debug_set_gemm_tall(prev)
    ms = e0.elapsed_time(e1) / reps
    tiles = (m // 128) * (m // 128 + 1) / 2 if lower else (m // 128) * (n // 128)
    return ms, tiles * 2 * 128 * 128 * k / ms * 1e-9, first


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(8192, 8192, 1024, True), (8320, 8320, 1024, True), (16384, 1024, 1024, False), (16256, 1024, 1024, False),
          (32768, 32768, 1024, True)]
if not quick:
    shapes += [(49152, 49152, 1024, True), (64512, 64512, 1024, True), (64512, 1024, 1024, False), (32768, 32768, 2048, True)]
for (m, n, k, lower) in shapes:
    ms0, tf0, c0 = run(m, n, k, lower, 0, keep=m <= 16384)
    ms1, tf1, c1 = run(m, n, k, lower, 2, keep=m <= 16384)
    same = ""
    if c0 is not None:
        d0 = torch.tril(c0) if lower else c0
        d1 = torch.tril(c1) if lower else c1
        same = "  identical" if torch.equal(d0, d1) else "  MAX DIFF %.3e" % float((d0 - d1).abs().max())
    print("M=%6d N=%6d K=%5d %s | 128x128: %8.3f ms %6.2f TF | 256x128: %8.3f ms %6.2f TF | %+5.1f %%%s"
          % (m, n, k, "lower" if lower else "full ", ms0, tf0, ms1, tf1, (tf1 / tf0 - 1) * 100, same), flush=True)
