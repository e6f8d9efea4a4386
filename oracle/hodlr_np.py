"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's HODLR solver.

Follows ``src/george/include/george/hodlr.h`` (``Node`` ctor :29-66, ``compute``
:75-103, ``solve`` :107-114, ``get_exact_matrix`` :122-133, ``low_rank_approx``
:136-221, ``factorize`` :223-235, ``apply_inverse`` :237-254) and the driver
``src/george/solvers/_hodlr.cpp:55-94`` (diag = yerr**2, mt19937 seeded with
``seed``, one generator threaded through the pre-order construction).

PARITY PARTLY UNPINNED: the reference extension itself cannot be built in this
image (it needs Eigen -- an un-vendored, un-pinned git submodule,
``.gitmodules:1-3``).  Deviations from a bit-level restatement, all inside the
tolerance the reference's own tests use (``tests/test_solvers.py:61-62``,
``tests/test_gp.py`` HODLR parametrisations -- ``np.allclose`` vs dense):

* leaves: Eigen ``LDLT`` (hodlr.h:227) -> LAPACK LU (``scipy.linalg.lu_factor``);
  ``log|det|`` = sum log|pivot| either way (hodlr.h:87-89).
* internal nodes: Eigen ``FullPivLU`` (hodlr.h:233) -> LAPACK partial-pivot LU.
* random row choice: ``std::mt19937`` raw stream is reproduced exactly
  (NumPy's MT19937 uses the same ``init_genrand``); ``uniform_int_distribution``
  is restated from libstdc++ >= 10 (``bits/uniform_int_dist.h``, Lemire's
  nearly-divisionless method on the 32-bit range) -- the reference build's
  libstdc++ version is not pinned, so the pivot order may differ from a given
  reference binary.
"""
import numpy as np
from scipy.linalg import lu_factor, lu_solve


class MT19937(object):
    """std::mt19937 seeded like ``random.seed(seed)`` (_hodlr.cpp:66-68)."""

    def __init__(self, seed):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(int(seed))
        self._buf = np.empty(0, dtype=np.uint64)
        self._pos = 0

    def __call__(self):
        if self._pos >= len(self._buf):
            self._buf = self._bg.random_raw(4096)
            self._pos = 0
        v = int(self._buf[self._pos])
        self._pos += 1
        return v


def uniform_int(rng, n):
    """uniform_int_distribution<int>(0, n-1)(rng): libstdc++ _S_nd<uint64_t>, 32-bit range."""
    rnge = n & 0xFFFFFFFF
    product = rng() * rnge
    low = product & 0xFFFFFFFF
    if low < rnge:
        threshold = ((1 << 32) - rnge) % rnge
        while low < threshold:
            product = rng() * rnge
            low = product & 0xFFFFFFFF
    return product >> 32


class _Node(object):

    def __init__(self, diag, rowfun, start, size, min_size, tol, rng, direction=0, parent=None):
        self.diag, self.rowfun = diag, rowfun
        self.start, self.size, self.direction, self.parent = start, size, direction, parent
        half = size // 2
        if half >= min_size:                                           # hodlr.h:48-49
            self.is_leaf = False
            self.rank, U1, V0 = self._low_rank_approx(start + half, size - half, start, half, tol, rng)
            self.U = [V0.copy(), U1]                                   # hodlr.h:53-55
            self.V = [V0, U1.copy()]
            self.children = [
                _Node(diag, rowfun, start, half, min_size, tol, rng, 0, self),
                _Node(diag, rowfun, start + half, size - half, min_size, tol, rng, 1, self),
            ]
        else:
            self.is_leaf = True
            self.rank = 0

    # hodlr.h:136-221
    def _low_rank_approx(self, start_row, n_rows, start_col, n_cols, tol, rng):
        max_rank = min(n_rows, n_cols)
        cap = 64
        U = np.zeros((n_rows, min(cap, max_rank)))
        V = np.zeros((n_cols, min(cap, max_rank)))
        rank = 0
        norm = 0.0
        tol2 = tol * tol
        index = list(range(n_rows))
        rows = np.arange(start_row, start_row + n_rows)
        cols = np.arange(start_col, start_col + n_cols)
        while True:
            while True:
                if not index:                                          # hodlr.h:160-176
                    B = self.rowfun(rows, cols)
                    if n_cols <= n_rows:
                        return max_rank, B.copy(), np.eye(n_cols, max_rank)
                    return max_rank, np.eye(n_rows, max_rank), B.T.copy()
                k = uniform_int(rng, len(index))                       # hodlr.h:179-183
                i = index[k]
                index[k] = index[-1]
                index.pop()
                v = self.rowfun(rows[i:i + 1], cols)[0]                # hodlr.h:186-188
                v = v - V[:, :rank] @ U[i, :rank]
                j = int(np.argmax(np.abs(v)))
                if abs(v[j]) >= 1e-14:
                    break
            v = v / v[j]                                               # hodlr.h:194
            u = self.rowfun(rows, cols[j:j + 1])[:, 0]                 # hodlr.h:197-199
            u = u - U[:, :rank] @ V[j, :rank]
            if rank >= U.shape[1]:
                grow = min(max_rank, 2 * U.shape[1])
                U = np.concatenate([U, np.zeros((n_rows, grow - U.shape[1]))], axis=1)
                V = np.concatenate([V, np.zeros((n_cols, grow - V.shape[1]))], axis=1)
            U[:, rank] = u
            V[:, rank] = v
            rank += 1
            if rank >= max_rank:
                break
            rowcol_norm = float(u @ u) * float(v @ v)                  # hodlr.h:206
            if rowcol_norm < tol2 * norm:
                break
            norm += rowcol_norm                                        # hodlr.h:210-214
            if rank > 1:
                norm += 2.0 * np.max(np.abs(U[:, :rank - 1].T @ u))
                norm += 2.0 * np.max(np.abs(V[:, :rank - 1].T @ v))
        return rank, U[:, :rank].copy(), V[:, :rank].copy()

    def _exact(self):                                                  # hodlr.h:122-133
        idx = np.arange(self.start, self.start + self.size)
        K = self.rowfun(idx, idx)
        K[np.diag_indices_from(K)] += self.diag[idx]
        return K

    def _factorize(self):                                              # hodlr.h:223-235
        if self.is_leaf:
            self.lu = lu_factor(self._exact())
        else:
            r = self.rank
            S = np.eye(2 * r)
            S[:r, r:] = self.V[1].T @ self.U[1]
            S[r:, :r] = self.V[0].T @ self.U[0]
            self.lu = lu_factor(S)

    def _apply_inverse(self, x, start):                                # hodlr.h:237-254
        s = self.start - start
        if self.is_leaf:
            x[s:s + self.size] = lu_solve(self.lu, x[s:s + self.size])
            return
        s1 = self.size // 2
        s2 = self.size - s1
        r = self.rank
        temp = np.empty((2 * r, x.shape[1]))
        temp[:r] = self.V[1].T @ x[s + s1:s + s1 + s2]
        temp[r:] = self.V[0].T @ x[s:s + s1]
        temp = lu_solve(self.lu, temp)
        x[s:s + s1] -= self.U[0] @ temp[:r]
        x[s + s1:s + s1 + s2] -= self.U[1] @ temp[r:]

    def compute(self):                                                 # hodlr.h:75-103
        self.log_det = 0.0
        if not self.is_leaf:
            self.children[0].compute()
            self.children[1].compute()
            self.log_det = self.children[0].log_det + self.children[1].log_det
        self._factorize()
        self.log_det += float(np.sum(np.log(np.abs(np.diag(self.lu[0])))))
        node, start, ind = self.parent, self.start, self.direction
        while node is not None:
            self._apply_inverse(node.U[ind], start)
            start, ind, node = node.start, node.direction, node.parent

    def solve(self, x):                                                # hodlr.h:107-114
        if not self.is_leaf:
            self.children[0].solve(x)
            self.children[1].solve(x)
        self._apply_inverse(x, 0)

    def nodes(self, out=None, level=0):
        """(level, start, size, rank) of every internal node in construction (pre-)order --
        the same walk as oracle/hodlr_ref_driver.cpp ``nodes()``."""
        out = [] if out is None else out
        if not self.is_leaf:
            out.append([level, self.start, self.size, self.rank])
            for c in self.children:
                c.nodes(out, level + 1)
        return out

    def ranks(self, out=None, level=0):
        out = {} if out is None else out
        if not self.is_leaf:
            out.setdefault(level, []).append(self.rank)
            for c in self.children:
                c.ranks(out, level + 1)
        return out


class HODLROracle(object):
    """HODLRSolver restated (reference src/george/solvers/hodlr.py:13-76 +
    _hodlr.cpp ``Solver`` :38-110)."""

    def __init__(self, spec, min_size=100, tol=0.1, seed=42, force_port=False):
        self.kernel, self.min_size, self.tol, self.seed = spec, min_size, tol, seed
        self.force_port = force_port
        self.computed = False
        self.log_determinant = None

    def compute(self, x, yerr):
        from . import ref_loader, kernels_np
        x = np.ascontiguousarray(x, dtype=np.float64)
        KI = None if self.force_port else ref_loader.load_kernel_interface()
        if KI is not None:
            ki = KI(self.kernel)
            rowfun = lambda r, c: ki.value_general(x[r], x[c])
        else:
            rowfun = lambda r, c: kernels_np.value_general(self.kernel, x[r], x[c])
        diag = (np.zeros(len(x)) + yerr) ** 2                          # _hodlr.cpp:76
        rng = MT19937(self.seed)
        self.root = _Node(diag, rowfun, 0, len(x), self.min_size, self.tol, rng)
        self.root.compute()
        self.log_determinant = self.root.log_det
        self.n = len(x)
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        b = np.array(y, dtype=np.float64, copy=True)
        shp = b.shape
        b2 = b.reshape(self.n, -1)
        self.root.solve(b2)
        return b2.reshape(shp)

    def dot_solve(self, y):
        return float(np.dot(y, self.apply_inverse(y)))

    def get_inverse(self):
        return self.apply_inverse(np.eye(self.n))

    def apply_sqrt(self, r):
        raise NotImplementedError("apply_sqrt is not implemented for the HODLRSolver")
