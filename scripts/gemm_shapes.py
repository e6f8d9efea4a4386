"""Where does the fp64 GEMM kernel stand against its ceilings?  Instruction micro-benchmarks
(clock under MFMA load) next to the kernel on shapes with different operand footprints:
small enough to live in the Infinity Cache / L2 versus the streaming SYRK of the factorisation."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from george_amd import _native as N

out = (C.c_double * 16)()
N.check(N.lib.gh_microbench_suite(out, 16))
o = list(out)
print("mfma16x16x4 1w/SIMD: %.1f TF  %.1f cyc/instr  %.2f GHz | 2w/SIMD: %.1f TF %.1f cyc %.2f GHz | 4w: %.1f TF"
      % (o[0], o[1], o[2], o[3], o[4], o[5], o[6]))
print("v_fma_f64 4w: %.1f TF %.2f cyc %.2f GHz | 8w: %.1f TF | mfma4x4x4: %.1f TF %.1f cyc" % (o[7], o[8], o[9], o[10], o[11], o[12]))


def run(m, n, k, flags, reps=4):
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    b = a if (flags & 4) else torch.randn(n, k, dtype=torch.float64, device="cuda")
    c = torch.randn(m, n, dtype=torch.float64, device="cuda")
    def go():
        N.check(N.lib.gh_dev_gemm(c.data_ptr(), n, a.data_ptr(), k, b.data_ptr(), k, m, n, k, -1.0, 1.0, flags, None))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tiles = (m // 128) * (m // 128 + 1) / 2 if (flags & 4) else (m // 128) * (n // 128)
    return ms, tiles * 2 * 128 * 128 * k / ms * 1e-9, tiles


for (m, n, k, fl) in [(4096, 4096, 1024, 0), (4096, 4096, 4096, 0), (8192, 8192, 1024, 0), (8192, 8192, 4096, 0),
                      (16384, 16384, 1024, 0), (32768, 32768, 1024, 4), (32768, 32768, 2048, 4),
                      (65536, 65536, 1024, 4), (262144, 128, 1024, 0), (128, 262144, 1024, 0)]:
    ms, tf, tiles = run(m, n, k, fl)
    print("M=%6d N=%6d K=%5d %s  tiles=%7d  %8.3f ms  %6.2f TFLOP/s" % (m, n, k, "lower" if fl & 4 else "full ", tiles, ms, tf))


# ---- what does the per-tile overhead consist of?  C read (beta), C row stride (TLB reach), K
def run_c(m, n, k, ldc, beta, reps=4):
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    b = torch.randn(n, k, dtype=torch.float64, device="cuda")
    c = torch.zeros(m, ldc, dtype=torch.float64, device="cuda")
    def go():
        N.check(N.lib.gh_dev_gemm(c.data_ptr(), ldc, a.data_ptr(), k, b.data_ptr(), k, m, n, k, -1.0, beta, 0, None))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, (m // 128) * (n // 128) * 2 * 128 * 128 * k / ms * 1e-9

for (m, n, k, ldc, beta) in [(8192, 8192, 1024, 8192, 1.0), (8192, 8192, 1024, 8192, 0.0), (8192, 8192, 1024, 65536, 1.0),
                             (8192, 8192, 1024, 65536, 0.0), (8192, 8192, 256, 8192, 1.0), (8192, 8192, 256, 65536, 1.0),
                             (8192, 8192, 512, 8192, 1.0), (8192, 8192, 2048, 8192, 1.0)]:
    ms, tf = run_c(m, n, k, ldc, beta)
    print("M=%5d N=%5d K=%5d ldc=%6d beta=%.0f  %8.3f ms  %6.2f TFLOP/s" % (m, n, k, ldc, beta, ms, tf))
