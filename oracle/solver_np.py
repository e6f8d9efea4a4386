"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's dense solver
path and of the GP-level formulas built on it.  Never imported by the product
path (only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

The arithmetic of the factorisation itself lives in a third-party dependency
of the reference: SciPy/LAPACK ``dpotrf``/``dpotrs`` (reference
``src/george/solvers/basic.py:8,68,87``; ``pyproject.toml:9`` lists unpinned
``numpy``/``scipy`` -- this image pins scipy 1.15.3 / OpenBLAS 0.3.28).  The
restatement therefore calls the *same* SciPy routines at the same call sites.

``kernel_matrix`` chooses the kernel evaluator: the reference's own compiled
C++ ``KernelInterface`` (``oracle/_ref``; ``kind="reference"``) when present,
else the NumPy restatement ``oracle/kernels_np.py`` (``kind="port"``).
"""
import numpy as np
from scipy.linalg import cholesky, cho_solve

from . import kernels_np
from . import ref_loader


def evaluator_kind():
    return "reference" if ref_loader.load_kernel_interface() is not None else "port"


def kernel_matrix(spec, x1, x2=None, diag=False, force_port=False):
    """Kernel.get_value (reference src/george/kernels.py:102-113)."""
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    KI = None if force_port else ref_loader.load_kernel_interface()
    if KI is not None:
        ki = KI(spec)
        if x2 is None:
            return ki.value_diagonal(x1, x1) if diag else ki.value_symmetric(x1)
        x2 = np.ascontiguousarray(x2, dtype=np.float64)
        return ki.value_diagonal(x1, x2) if diag else ki.value_general(x1, x2)
    if x2 is None:
        return kernels_np.value_diagonal(spec, x1, x1) if diag else kernels_np.value_symmetric(spec, x1)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    return kernels_np.value_diagonal(spec, x1, x2) if diag else kernels_np.value_general(spec, x1, x2)


def kernel_gradient(spec, x, force_port=False):
    """Kernel.get_gradient with include_frozen=True: (n, n, full_size)
    (reference src/george/kernels.py:115-127)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    KI = None if force_port else ref_loader.load_kernel_interface()
    if KI is not None:
        which = np.ones(kernels_np.full_size(spec), dtype=np.uint32)
        return KI(spec).gradient_symmetric(which, x)
    return kernels_np.gradient_symmetric(spec, x)


class DenseOracle(object):
    """BasicSolver restated (reference src/george/solvers/basic.py:20-121)."""

    def __init__(self, spec, force_port=False):
        self.kernel = spec
        self.force_port = force_port
        self.computed = False
        self.log_determinant = None

    def compute(self, x, yerr):
        # basic.py:64-70
        K = kernel_matrix(self.kernel, x, force_port=self.force_port)
        K[np.diag_indices_from(K)] += yerr ** 2
        self._factor = (cholesky(K, overwrite_a=True, lower=False), False)
        self.log_determinant = 2 * np.sum(np.log(np.diag(self._factor[0])))
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        return cho_solve(self._factor, y, overwrite_b=in_place)       # basic.py:87

    def dot_solve(self, y):
        return np.dot(y.T, cho_solve(self._factor, y))                # basic.py:102

    def apply_sqrt(self, r):
        return np.dot(r, self._factor[0])                             # basic.py:114

    def get_inverse(self):
        return self.apply_inverse(np.eye(len(self._factor[0])), in_place=True)   # basic.py:121


def blocked_cholesky_lower(K, nb=8192):
    """In-place lower Cholesky of the C-contiguous symmetric K by LAPACK/BLAS calls on blocks of at most n x nb
    elements: dpotrf on the diagonal block, dtrsm on the rows below, dgemm on the trailing block columns -- LAPACK's
    own right-looking blocked algorithm written out, for sizes at which this image's whole-matrix ``dpotrf`` (the call
    of basic.py:68) is unreliable (oracle/potrf_probe.py, oracle/gen_golden_large.py)."""
    from scipy.linalg import solve_triangular
    n = len(K)
    for k in range(0, n, nb):
        e = min(k + nb, n)
        Lkk = cholesky(K[k:e, k:e], lower=True, check_finite=False)
        K[k:e, k:e] = Lkk
        if e < n:
            K[e:, k:e] = solve_triangular(Lkk, K[e:, k:e].T, lower=True, check_finite=False).T
            for j in range(e, n, nb):
                je = min(j + nb, n)
                K[j:, j:je] -= K[j:, k:e] @ K[j:je, k:e].T
    return K


def blocked_solve_lower(L, b, nb=8192, trans=False):
    from scipy.linalg import solve_triangular
    n = len(L)
    x = np.array(b, dtype=np.float64, copy=True)
    starts = list(range(0, n, nb))
    if not trans:
        for k in starts:
            e = min(k + nb, n)
            x[k:e] = solve_triangular(L[k:e, k:e], x[k:e], lower=True, check_finite=False)
            if e < n:
                x[e:] -= L[e:, k:e] @ x[k:e]
    else:
        for k in reversed(starts):
            e = min(k + nb, n)
            x[k:e] = solve_triangular(L[k:e, k:e], x[k:e], lower=True, trans=1, check_finite=False)
            if k > 0:
                x[:k] -= L[k:e, :k].T @ x[k:e]
    return x


class BlockedDenseOracle(DenseOracle):
    """DenseOracle with the factorisation and the solves on ``nb``-column blocks (same LAPACK/BLAS routines)."""

    def __init__(self, spec, force_port=False, nb=8192):
        DenseOracle.__init__(self, spec, force_port)
        self.nb = nb

    def compute(self, x, yerr):
        K = kernel_matrix(self.kernel, x, force_port=self.force_port)
        K[np.diag_indices_from(K)] += yerr ** 2
        self._L = blocked_cholesky_lower(K, self.nb)
        self.log_determinant = 2 * np.sum(np.log(np.diag(self._L)))
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        return blocked_solve_lower(self._L, blocked_solve_lower(self._L, y, self.nb), self.nb, trans=True)

    def dot_solve(self, y):
        z = blocked_solve_lower(self._L, y, self.nb)
        return float(np.dot(z.T, z))

    def apply_sqrt(self, r):
        return np.dot(r, np.triu(self._L.T))

    def get_inverse(self):
        return self.apply_inverse(np.eye(len(self._L)))


TINY = 1.25e-12      # reference src/george/gp.py:19


def gp_log_likelihood(solver, x, yerr, y, mean=0.0, white_noise=np.log(TINY)):
    """GP.compute + GP.log_likelihood (reference src/george/gp.py:303-337, 369-397)
    for a constant mean and constant log-white-noise."""
    x = np.ascontiguousarray(np.atleast_2d(x.T).T if np.ndim(x) == 1 else x, dtype=np.float64)
    yerr2 = (np.zeros(len(x)) + yerr) ** 2
    solver.compute(x, np.sqrt(yerr2 + np.exp(white_noise)))            # gp.py:330-331
    const = -0.5 * (len(x) * np.log(2 * np.pi) + solver.log_determinant)   # gp.py:333-335
    r = np.ascontiguousarray(y - mean, dtype=np.float64)
    ll = const - 0.5 * solver.dot_solve(r)                            # gp.py:396
    return ll if np.isfinite(ll) else -np.inf


def gp_predict(solver, spec, x, y, t, mean=0.0, return_var=True, return_cov=False, force_port=False):
    """GP.predict (reference src/george/gp.py:482-545); solver already computed."""
    alpha = solver.apply_inverse(np.ascontiguousarray(y - mean, dtype=np.float64))
    Kxs = kernel_matrix(spec, t, x, force_port=force_port)            # gp.py:532
    mu = np.dot(Kxs, alpha) + mean                                    # gp.py:533
    if not (return_var or return_cov):
        return mu
    KinvKxs = solver.apply_inverse(Kxs.T)                             # gp.py:537
    if return_var:
        var = kernel_matrix(spec, t, diag=True, force_port=force_port)
        var -= np.sum(Kxs.T * KinvKxs, axis=0)                        # gp.py:539-541
        return mu, var
    cov = kernel_matrix(spec, t, force_port=force_port)
    cov -= np.dot(Kxs, KinvKxs)                                       # gp.py:543-545
    return mu, cov


def gp_grad_log_likelihood(solver, spec, x, y, mean=0.0, force_port=False):
    """Kernel part of GP.grad_log_likelihood (reference src/george/gp.py:429-466),
    for ALL kernel parameters (include_frozen=True order)."""
    alpha = solver.apply_inverse(np.ascontiguousarray(y - mean, dtype=np.float64))
    K_inv = solver.get_inverse()                                      # gp.py:436
    A = np.einsum("i,j", alpha, alpha) - K_inv                        # gp.py:437
    Kg = kernel_gradient(spec, x, force_port=force_port)              # gp.py:465
    return 0.5 * np.einsum("ijk,ij", Kg, A), A                        # gp.py:466
