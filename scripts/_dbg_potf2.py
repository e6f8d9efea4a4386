import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
from george_amd import _native as N
n = 128
rng = np.random.default_rng(0)
a = rng.standard_normal((n, n)); a = a @ a.T + n * np.eye(n)
A = torch.from_numpy(a).cuda(); dinv = torch.zeros(n, n, dtype=torch.float64, device="cuda"); info = torch.zeros(1, dtype=torch.int64, device="cuda")
for rep in range(3):
    A.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
    N.check(N.lib.gh_dev_potrf_block(A.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 0, None)); torch.cuda.synchronize()
    buf = (C.c_longlong * 16)(); N.lib.gh_debug_read(buf, 16)
    t = np.array(buf[:9]); print("phase1 a,b,c:", list(buf[9:12]));  print("cycles since start:", (t - t[0]).tolist(), " deltas:", np.diff(t).tolist())
