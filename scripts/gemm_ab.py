"""A/B of the GEMM instruction modes on the trailing-update shape (SYRK, lower, K = nb)."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from george_amd import _native as N

def run(n, k, mode, reps=5):
    torch.manual_seed(0)
    p = torch.randn(n, k, dtype=torch.float64, device="cuda")
    c = torch.randn(n, n, dtype=torch.float64, device="cuda")
    N.lib.gh_debug_set_mfma(mode)
    def go():
        N.check(N.lib.gh_dev_gemm(c.data_ptr(), n, p.data_ptr(), k, p.data_ptr(), k, n, n, k, -1.0, 1.0, 4, None))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tiles = (n // 128) * (n // 128 + 1) / 2
    fl = tiles * 2 * 128 * 128 * k
    return ms, fl / ms * 1e-9

def check(n=1024, k=256):
    """DMA kernel against the register-staged one, bit for bit (same MFMA order)."""
    torch.manual_seed(1)
    p = torch.randn(n, k, dtype=torch.float64, device="cuda")
    q = torch.randn(n, k, dtype=torch.float64, device="cuda")
    outs = []
    for mode in (1, 3):
        N.lib.gh_debug_set_mfma(mode)
        c = torch.ones(n, n, dtype=torch.float64, device="cuda")
        N.check(N.lib.gh_dev_gemm(c.data_ptr(), n, p.data_ptr(), k, q.data_ptr(), k, n, n, k, -1.0, 1.0, 0, None))
        torch.cuda.synchronize()
        outs.append(c)
    ref = 1.0 - p @ q.T
    print("dma vs staged max diff", float((outs[0] - outs[1]).abs().max()), " vs torch", float((outs[0] - ref).abs().max()))

check()

for n, k in [(16384, 512), (32768, 512), (32768, 1024), (49152, 512)]:
    for mode in (1, 3):      # 1 = LDS-DMA kernel, 3 = register-staged, (2 = 4x4x4 MFMA arm)
        ms, tf = run(n, k, mode)
        print("n=%6d k=%5d mode=%d  %8.3f ms  %6.2f TFLOP/s" % (n, k, mode, ms, tf))
