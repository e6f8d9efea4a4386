"""TEST INFRASTRUCTURE ONLY: NumPy stand-in for george_amd.distributed.HipTileOps (same methods, torch
CPU tensors, NumPy / oracle arithmetic).  Used by tests/test_distributed.py and by the launcher
self-test of bench.py (``--tile-ops numpy``: exercises ``--gpus N`` -> N ranks -> one JSON line on a
box without a GPU; such a line says so in ``data`` and is not a measurement)."""
import numpy as np
import torch


class NumpyTileOps(object):
    """CPU stand-in for HipTileOps: same methods, torch CPU tensors, NumPy/oracle arithmetic."""

    def __init__(self, kernel_spec):
        self.kernel = kernel_spec
        self.ndim = kernel_spec.ndim

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or torch.float64)

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def kmat(self, x, n, yerr, row0, nrows, col0, ncols, out):
        from oracle import kernels_np
        X, e = x.numpy(), yerr.numpy()
        blk = np.zeros((nrows, ncols))
        vr, vc = max(0, min(nrows, n - row0)), max(0, min(ncols, n - col0))
        if vr and vc:
            blk[:vr, :vc] = kernels_np.value_general(self.kernel, X[row0:row0 + vr], X[col0:col0 + vc])
        for r in range(nrows):
            c = row0 + r - col0
            if 0 <= c < ncols:
                blk[r, c] = blk[r, c] + e[row0 + r] ** 2 if row0 + r < n else 1.0
        out.copy_(torch.from_numpy(blk))

    def potrf(self, a, dinv, info, base):
        A = np.tril(a.numpy()) + np.tril(a.numpy(), -1).T
        try:
            L = np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            if int(info.item()) == 0:
                info.fill_(base + 1)
            return
        a.copy_(torch.from_numpy(L))
        for b in range(a.shape[0] // 128):
            dinv[b].copy_(torch.from_numpy(np.linalg.inv(L[128 * b:128 * b + 128, 128 * b:128 * b + 128])))

    def trsm(self, l11, dinv, a21):
        a21.copy_(torch.from_numpy(np.linalg.solve(np.tril(l11.numpy()), a21.numpy().T).T))

    def gemm_nt(self, c, a, b):
        c -= a @ b.T

    def gemm_nt_stair(self, c, a, b, group_rows, widths):
        """row group g of c -= a[g-th rows] @ b[:widths[g]].T (HipTileOps.gemm_nt_stair / gh_dev_gemm_nt_stair)"""
        assert all(w0 <= w1 for w0, w1 in zip(widths, widths[1:])) and all(w > 0 and w % 128 == 0 for w in widths)
        assert c.shape[0] >= group_rows * len(widths) and a.shape[0] >= group_rows * len(widths) and b.shape[0] >= widths[-1]
        for g, w in enumerate(widths):
            r = slice(g * group_rows, (g + 1) * group_rows)
            c[r, :w] -= a[r] @ b[:w].T

    def gemm(self, c, a, b, alpha=1.0, beta=0.0, a_t=False, b_t=False):
        A = a.T if a_t else a
        B = b.T if b_t else b
        c.copy_(beta * c + alpha * (A @ B) if beta != 0.0 else alpha * (A @ B))

    def gemv(self, a, x, y, alpha, beta):
        y.copy_(beta * y + alpha * (a @ x) if beta != 0.0 else alpha * (a @ x))

    def logdet_accum(self, a, out):
        out += 2.0 * torch.log(torch.diagonal(a)).sum()

    def trsv(self, l, dinv, w, z):
        import scipy.linalg
        z.copy_(torch.from_numpy(scipy.linalg.solve_triangular(np.tril(l.numpy()), w.numpy(), lower=True)))

    def sync(self):
        pass


class NumpyTileOpsPerRow(NumpyTileOps):
    """The same without the staircase launch: the driver then issues one gemm_nt per local tile row."""
    gemm_nt_stair = None
