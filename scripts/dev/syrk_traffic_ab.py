"""One SYRK-shaped launch per arm (M = N = 32768 / 65536 lower, K = 1024 and 2048) with the half-slab pipelined loop on / off
(gh_debug_set_gemm_sp) x the grouped tile order on / off (gh_debug_set_gemm_grouped), every arm a dispatch of its own in ONE
process, in a fixed order that scripts/dev/syrk_traffic_table.py decodes from a rocprofv3 --pmc FETCH_SIZE (or WRITE_SIZE)
--kernel-trace run.  Without a profiler it prints the arms' times (HIP events, 3 launches each)."""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from george_amd import _native as N  # noqa: E402
lib = N.lib
dp = C.POINTER(C.c_double)

def gemm(c, a, m, k):
    rc = lib.gh_dev_gemm(C.cast(c.data_ptr(), dp), c.stride(0), C.cast(a.data_ptr(), dp), a.stride(0), C.cast(a.data_ptr(), dp), a.stride(0),
                         m, m, k, -1.0, 1.0, 4, None)
    assert rc == 0, N.last_error()

SHAPES = [(32768, 1024), (32768, 2048), (57344, 1024)]
ARMS = [(0, 0), (0, 1), (1, 0), (1, 1)]            # (sp, grouped)
reps = int(os.environ.get("REPS", "1"))
print("| M | K | sp | grouped | ms | TFLOP/s |\n|---|---|---|---|---|---|")
for (m, k) in SHAPES:
    torch.manual_seed(0)
    a = torch.randn(m, k, dtype=torch.float64, device="cuda")
    c = torch.zeros(m, m, dtype=torch.float64, device="cuda")
    tiles = (m // 128) * (m // 128 + 1) / 2
    for (sp, gr) in ARMS:
        lib.gh_debug_set_gemm_sp(sp); lib.gh_debug_set_gemm_grouped(gr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): gemm(c, a, m, k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("| %d | %d | %d | %d | %.3f | %.2f |" % (m, k, sp, gr, ms, tiles * 2 * 128 * 128 * k / ms * 1e-9), flush=True)
    del a, c
lib.gh_debug_set_gemm_sp(-1); lib.gh_debug_set_gemm_grouped(-1)
