"""TEST INFRASTRUCTURE ONLY.  Probe: is SciPy's whole-matrix ``cholesky`` (LAPACK dpotrf, the call
of reference ``basic.py:68``) usable at size n in this image?  A = I + 1e-3 * ones has
log|A| = log(1 + 1e-3 n) exactly.

    python oracle/potrf_probe.py 65536
    -> n 65536 ok logdet 4.175879788749634 exact 4.197743154566386   (build container, 7 threads, 237 s)

i.e. at n = 65536 (n * n = 2^32) the factorisation returns without error and is WRONG in the third
digit (it is right at n = 16384 and 20480, where gen_golden_large.py's results agree with the HIP path
and with the block-wise factorisation).  oracle/gen_golden_large.py therefore factors the
configurations with N >= 32768 block-wise."""
import numpy as np, time, sys
from scipy.linalg import cholesky
n = int(sys.argv[1])
A = np.full((n, n), 1e-3)
A[np.diag_indices(n)] += 1.0
t = time.time()
try:
    U = cholesky(A.T, overwrite_a=True, lower=False, check_finite=False)
    ld = 2*np.sum(np.log(np.diag(U)))
    print("n", n, "ok logdet", ld, "exact", np.log1p(1e-3*n), "t", time.time()-t, flush=True)
except Exception as e:
    print("n", n, "FAILED", e, "t", time.time()-t, flush=True)
