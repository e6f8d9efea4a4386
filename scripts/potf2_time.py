"""Isolated timing and accuracy of the 128x128 potf2+inverse kernel (through gh_dev_potrf_block)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from george_amd import _native as N
n = 128
rng = np.random.default_rng(0)
a = rng.standard_normal((n, n)); a = a @ a.T + n * np.eye(n)
A = torch.from_numpy(a).cuda(); dinv = torch.zeros(n, n, dtype=torch.float64, device="cuda"); info = torch.zeros(1, dtype=torch.int64, device="cuda")
N.check(N.lib.gh_dev_potrf_block(A.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 0, None)); torch.cuda.synchronize()
L = np.tril(A.cpu().numpy()); Lr = np.linalg.cholesky(a)
print("max |L - L_ref| / |L_ref|:", np.abs(L - Lr).max() / np.abs(Lr).max(), " |dinv L - I|:", np.abs(dinv.cpu().numpy() @ L - np.eye(n)).max())
reps = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
A0 = torch.from_numpy(a).cuda(); B = A0.clone()
e0.record()
for _ in range(reps):
    B.copy_(A0)
e1.record(); torch.cuda.synchronize()
t_copy = e0.elapsed_time(e1) / reps * 1e3
e0.record()
for _ in range(reps):
    B.copy_(A0)
    N.lib.gh_dev_potrf_block(B.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 0, None)
e1.record(); torch.cuda.synchronize()
print("us per potf2 (back-to-back, copy of the input subtracted):", e0.elapsed_time(e1) / reps * 1e3 - t_copy, "info", int(info.item()))
