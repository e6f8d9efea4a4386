"""One SYRK-shaped launch pair for the PMC passes of scripts/profile_tall.sh:
    python scripts/gemm_tall_one.py <tall 0|2> [M] [K]      (lower-triangular C -= A A^T, twice)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from george_amd import _native as N

tall = int(sys.argv[1])
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
torch.manual_seed(0)
a = torch.randn(m, k, dtype=torch.float64, device="cuda")
c = torch.randn(m, m, dtype=torch.float64, device="cuda")
N.lib.gh_debug_set_gemm_tall(tall)
for _ in range(2):
    N.check(N.lib.gh_dev_gemm(c.data_ptr(), m, a.data_ptr(), k, a.data_ptr(), k, m, m, k, -1.0, 1.0, 4, None))
torch.cuda.synchronize()
t = m // 128
print("tiles", t * (t + 1) // 2, "algorithmic bytes per launch", t * (t + 1) // 2 * 128 * 128 * 8 * 2 + m * k * 8,
      "flops per launch", t * (t + 1) // 2 * 2 * 128 * 128 * k)
