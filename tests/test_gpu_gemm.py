"""Device unit tests of the building blocks behind the C ABI (need an MI355X): fp64 MFMA GEMM in
every operand layout vs NumPy and vs the plain-VALU arm, diagonal-block Cholesky + inverse,
TRSM, and the roofline micro-benchmarks.  Operands are ASYMMETRIC random matrices so a
transposed fragment map cannot pass."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLAGS = dict(A_MMAJOR=1, B_NMAJOR=2, LOWER=4, KLO_MAX=8, KHI_COL=16, KHI_ROW=32)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _gemm(c, a, b, m, n, k, alpha, beta, flags, lda, ldb, ldc):
    from george_amd import _native as N
    import torch
    N.check(N.lib.gh_dev_gemm(c.data_ptr(), ldc, a.data_ptr(), lda, b.data_ptr(), ldb, m, n, k,
                              alpha, beta, flags, None))
    torch.cuda.synchronize()


def test_microbench_peaks():
    from george_amd import _native as N
    tf, gb = C.c_double(0), C.c_double(0)
    N.check(N.lib.gh_microbench_mfma_f64(C.byref(tf)))
    N.check(N.lib.gh_microbench_hbm_copy(C.byref(gb)))
    print("\n[microbench] v_mfma_f64_16x16x4_f64 peak: %.2f TFLOP/s ; 16B copy: %.0f GB/s" % (tf.value, gb.value))
    assert 20.0 < tf.value < 200.0
    assert 1000.0 < gb.value < 9000.0
    out = (C.c_double * 16)()
    N.check(N.lib.gh_microbench_suite(out, 16))
    o = list(out)
    print("[suite] mfma_f64_16x16x4: 1w/SIMD %.1f TF (%.1f cyc/instr @ %.2f GHz), 2w/SIMD %.1f TF (%.1f cyc @ %.2f GHz), "
          "4w/SIMD %.1f TF | v_fma_f64: 4w/SIMD %.1f TF (%.2f cyc/instr @ %.2f GHz), 8w/SIMD %.1f TF | "
          "mfma_f64_4x4x4: %.1f TF (%.1f cyc/instr)" % (o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12]))


def test_mfma_f64_ceiling_is_a_ceiling():
    """SURVEY 8(d)'s second denominator: the bare issue loop must reach at least what the GEMM kernel sustains (round 3's
    loop read 34-47 TFLOP/s against 68-70 in the GEMM: hipcc had wrapped its eight matrix instructions in 128 VGPR <-> AGPR
    accumulator moves per iteration, so it measured the vector ALU's move rate, not the pipe) and cannot exceed the datasheet's 78.6 by more than clock tolerance."""
    import torch
    from george_amd import _native as N
    out = (C.c_double * 8)()
    N.check(N.lib.gh_microbench_mfma_f64_ceiling(out, 8))
    o = list(out)
    print("\n[ceiling] v_mfma_f64_16x16x4 bare issue: %.1f / %.1f / %.1f TFLOP/s at 1 / 2 / 4 wavefronts per SIMD (%.2f / %.2f / %.2f ms)"
          % (o[0], o[1], o[2], o[4], o[5], o[6]))
    M, K = 16384, 4096
    a = torch.randn(M, K, dtype=torch.float64, device="cuda")
    c = torch.zeros(M, M, dtype=torch.float64, device="cuda")
    _gemm(c, a, a, M, M, K, -1.0, 1.0, 0, K, K, M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        N.check(N.lib.gh_dev_gemm(c.data_ptr(), M, a.data_ptr(), K, a.data_ptr(), K, M, M, K, -1.0, 1.0, 0, None))
    e1.record()
    torch.cuda.synchronize()
    gemm_tf = 3 * 2.0 * M * M * K / (e0.elapsed_time(e1) * 1e-3) * 1e-12
    print("[ceiling] gemm_f64_mfma_dma at M = N = %d, K = %d: %.1f TFLOP/s = %.3f of the measured ceiling" % (M, K, gemm_tf, gemm_tf / o[3]))
    assert o[3] >= 0.98 * gemm_tf, (o, gemm_tf)
    assert o[3] <= 78.6 * 1.05


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("a_mm,b_nm", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_layouts(mfma, a_mm, b_nm):
    from george_amd import _native as N
    rng = np.random.RandomState(7 + a_mm * 2 + b_nm)
    M, Nn, K = 256, 384, 272
    A = rng.randn(M, K)
    B = rng.randn(Nn, K)
    C0 = rng.randn(M, Nn + 64)                       # ldc > n: strided C
    a = _dev(A.T if a_mm else A)
    b = _dev(B.T if b_nm else B)
    c = _dev(C0)
    prev = N.lib.gh_debug_set_mfma(mfma)
    try:
        flags = (FLAGS["A_MMAJOR"] if a_mm else 0) | (FLAGS["B_NMAJOR"] if b_nm else 0)
        _gemm(c, a, b, M, Nn, K, -1.5, 0.75, flags, M if a_mm else K, Nn if b_nm else K, Nn + 64)
    finally:
        N.lib.gh_debug_set_mfma(prev)
    got = c.cpu().numpy()
    want = C0.copy()
    want[:, :Nn] = 0.75 * C0[:, :Nn] - 1.5 * A @ B.T
    err = np.abs(got - want).max()
    assert err < 1e-11, "max err %g (mfma=%d a_mm=%d b_nm=%d); first bad %s" % (
        err, mfma, a_mm, b_nm, np.argwhere(np.abs(got - want) > 1e-11)[:4].tolist())
    # beta == 0 must not read C (NaN garbage allowed)
    c2 = _dev(np.full((M, Nn), np.nan))
    prev = N.lib.gh_debug_set_mfma(mfma)
    try:
        _gemm(c2, a, b, M, Nn, K, 1.0, 0.0, flags, M if a_mm else K, Nn if b_nm else K, Nn)
    finally:
        N.lib.gh_debug_set_mfma(prev)
    assert np.abs(c2.cpu().numpy() - A @ B.T).max() < 1e-11


def test_gemm_lower_and_k_clipping():
    rng = np.random.RandomState(3)
    n = 512
    P = rng.randn(n, 256)
    C0 = rng.randn(n, n)
    c = _dev(C0)
    p = _dev(P)
    _gemm(c, p, p, n, n, 256, -1.0, 1.0, FLAGS["LOWER"], 256, 256, n)        # SYRK, lower tiles only
    got = c.cpu().numpy()
    want = C0 - P @ P.T
    for ti in range(4):
        for tj in range(4):
            blk = (slice(128 * ti, 128 * ti + 128), slice(128 * tj, 128 * tj + 128))
            if tj <= ti:
                assert np.abs(got[blk] - want[blk]).max() < 1e-11
            else:
                assert np.array_equal(got[blk], C0[blk])                      # upper tiles untouched
    # triangular operands: W = Linv^T Linv with k >= max(row0, col0); out = r L^T with k <= col0+128
    L = np.tril(rng.randn(n, n))
    l = _dev(L)
    w = _dev(np.zeros((n, n)))
    _gemm(w, l, l, n, n, n, 1.0, 0.0, FLAGS["A_MMAJOR"] | FLAGS["B_NMAJOR"] | FLAGS["LOWER"] | FLAGS["KLO_MAX"], n, n, n)
    assert np.abs(np.tril(w.cpu().numpy()) - np.tril(L.T @ L)).max() < 1e-10
    R = rng.randn(128, n)
    o = _dev(np.zeros((128, n)))
    _gemm(o, _dev(R), l, 128, n, n, 1.0, 0.0, FLAGS["KHI_COL"], n, n, n)
    assert np.abs(o.cpu().numpy() - R @ L.T).max() < 1e-10


@pytest.mark.parametrize("mfma", [1, 0])
def test_potrf_block_and_trsm(mfma):
    import torch
    from george_amd import _native as N
    rng = np.random.RandomState(11)
    n, m = 384, 256
    G = rng.randn(n, n)
    A = G @ G.T + n * np.eye(n)
    a = _dev(np.tril(A) + np.triu(rng.randn(n, n), 1) * 0)      # only the lower triangle matters
    dinv = _dev(np.zeros((n // 128, 128, 128)))
    info = torch.zeros(1, dtype=torch.int64, device="cuda")
    prev = N.lib.gh_debug_set_mfma(mfma)
    try:
        N.check(N.lib.gh_dev_potrf_block(a.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 0, None))
        torch.cuda.synchronize()
        L = np.linalg.cholesky(A)
        got = np.tril(a.cpu().numpy())
        assert int(info.item()) == 0
        assert np.abs(got - L).max() < 1e-10 * np.abs(L).max()
        d = dinv.cpu().numpy()
        for j in range(n // 128):
            Ljj = L[128 * j:128 * j + 128, 128 * j:128 * j + 128]
            assert np.abs(d[j] @ Ljj - np.eye(128)).max() < 1e-10
        B = rng.randn(m, n)
        b = _dev(B)
        N.check(N.lib.gh_dev_trsm_right(a.data_ptr(), n, dinv.data_ptr(), b.data_ptr(), n, m, n, None))
        torch.cuda.synchronize()
        want = np.linalg.solve(L, B.T).T                                      # B L^-T
        assert np.abs(b.cpu().numpy() - want).max() < 1e-10 * max(1.0, np.abs(want).max())
    finally:
        N.lib.gh_debug_set_mfma(prev)


def test_potrf_reports_failing_pivot():
    import torch
    from george_amd import _native as N
    n = 256
    A = np.eye(n)
    A[200, 200] = -1.0
    a = _dev(A)
    dinv = _dev(np.zeros((2, 128, 128)))
    info = torch.zeros(1, dtype=torch.int64, device="cuda")
    N.check(N.lib.gh_dev_potrf_block(a.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 1000, None))
    torch.cuda.synchronize()
    assert int(info.item()) == 1000 + 201          # LAPACK dpotrf-style 1-based index


@pytest.mark.parametrize("p", [0, 5, 15, 16, 31, 100, 127, 128, 143, 255])
@pytest.mark.parametrize("bad", ["negative", "nan"])
def test_potrf_failing_pivot_index_matches_lapack(p, bad):
    """The second 128x128 kernel has no pivot test inside its column steps: a bad pivot poisons what
    follows and the first non-positive diagonal entry of the stored 16x16 block is looked for afterwards
    (gh_potf2_body.h).  The index it reports must still be LAPACK's, wherever the pivot sits in a
    16-column step, a 128-block or the panel."""
    import torch
    from scipy.linalg import lapack
    from george_amd import _native as N
    rng = np.random.RandomState(3)
    n = 256
    G = rng.randn(n, n)
    A = G @ G.T + n * np.eye(n)
    A[p, p] = -1.0e6 if bad == "negative" else np.nan
    _, info_ref = lapack.dpotrf(A, lower=1)
    if bad == "negative":
        assert info_ref == p + 1
    a = _dev(A)
    dinv = _dev(np.zeros((2, 128, 128)))
    info = torch.zeros(1, dtype=torch.int64, device="cuda")
    N.check(N.lib.gh_dev_potrf_block(a.data_ptr(), n, n, dinv.data_ptr(), info.data_ptr(), 0, None))
    torch.cuda.synchronize()
    assert int(info.item()) == p + 1


@pytest.mark.parametrize("rows,cols,K", [(5, 2, 80), (9, 9, 48), (12, 1, 1024), (40, 7, 256), (3, 3, 16)])
def test_lower_trapezoid(rows, cols, K):
    """GH_GEMM_LOWER with m > n: the leading n columns of a lower-triangular C (tiles with tile row >= tile column); tiles above the
    diagonal stay untouched, and
    every computed tile is bit-identical to the same tile of the full lower-triangular launch."""
    rng = np.random.RandomState(rows * 7 + cols)
    M, Nn = 128 * rows, 128 * cols
    A = rng.randn(M, K)
    C0 = rng.randn(M, M)
    a, c = _dev(A), _dev(C0)
    _gemm(c, a, a, M, Nn, K, -1.0, 1.0, FLAGS["LOWER"], K, K, M)
    got = c.cpu().numpy()
    cf = _dev(C0)
    _gemm(cf, a, a, M, M, K, -1.0, 1.0, FLAGS["LOWER"], K, K, M)
    full = cf.cpu().numpy()
    blk = np.kron(np.tril(np.ones((rows, rows))), np.ones((128, 128))).astype(bool)
    blk[:, Nn:] = False
    assert np.array_equal(got[blk], full[blk])
    assert np.array_equal(got[~blk], C0[~blk])
    assert np.abs(got - (C0 - A @ A.T))[blk].max() < 1e-10 * max(1.0, K / 16)


@pytest.mark.parametrize("rows,cols,K,lower", [(130, 130, 1024, True), (137, 129, 1024, True), (9, 129, 1024, False),
                                               (33, 17, 8192, False)])
def test_grouped_tile_order(rows, cols, K, lower):
    """Launches whose column operand exceeds 128 MiB walk their tiles in row groups of 8, column-major inside a group
    (gh_gemm.hip, tile_of / grouped_tri / grouped_rect): a partial last group, a trapezoid (triangle part + rectangular part) and
    plain rectangles.  Every tile must be produced exactly once -- a tile visited twice would subtract A B^T twice, a tile
    missed would keep C0 -- and bit-identically to the same tile computed by a launch small enough for the row-major order."""
    import torch
    M, Nn = 128 * rows, 128 * cols
    assert Nn * K * 8 > (128 << 20)
    g = torch.Generator(device="cuda").manual_seed(rows * 131 + cols)
    a = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    b = a if lower else torch.randn(Nn, K, dtype=torch.float64, device="cuda", generator=g)
    c0 = torch.randn(M, Nn, dtype=torch.float64, device="cuda", generator=g)
    c = c0.clone()
    _gemm(c, a, b, M, Nn, K, -1.0, 1.0, FLAGS["LOWER"] if lower else 0, K, K, Nn)
    want = c0 - a @ b[:Nn].T
    tr = torch.arange(M, device="cuda")[:, None] // 128
    tc = torch.arange(Nn, device="cuda")[None, :] // 128
    blk = (tr >= tc) if lower else torch.ones(M, Nn, dtype=torch.bool, device="cuda")
    assert torch.equal(c[~blk], c0[~blk])
    err = (c - want)[blk].abs().max().item()
    assert err < 1e-10 * K / 16, err
    # the same tiles from row-major launches of the same kernel: all rows, half the columns (<= 128 MiB of column operand)
    half = (cols // 2) * 128
    for c_lo, c_hi in ((0, half), (half, Nn)):
        assert (c_hi - c_lo) * K * 8 <= (128 << 20) and rows * (c_hi - c_lo) // 128 > 128
        ref = c0[:, c_lo:c_hi].contiguous()
        _gemm(ref, a, b[c_lo:c_hi], M, c_hi - c_lo, K, -1.0, 1.0, 0, K, K, c_hi - c_lo)
        m = blk[:, c_lo:c_hi]
        assert torch.equal(ref[m], c[:, c_lo:c_hi][m])


@pytest.mark.parametrize("group_rows,widths,K", [(1024, [1024, 3072, 3072, 9216, 16384, 17408], 1024),     # the multi-GPU update's shape
                                                  (128, [128 * w for w in range(1, 41)], 256),                 # 40 one-tile-high groups
                                                  (256, [128] * 3, 128),                                       # sub-chip: per-group fallback
                                                  (128, [128 * (1 + w // 2) for w in range(70)], 128),         # 70 groups: still one launch
                                                  (128, [128 * (1 + w // 2) for w in range(140)], 128)])      # > 128 groups: per-group fallback
def test_staircase_gemm(group_rows, widths, K):
    """gh_dev_gemm_nt_stair: row group g of c is updated over its first widths[g] columns, all groups in one launch (widest group
    first, 8 x 8 tile blocks inside a group).  Against torch per group, bit-identical to one gh_dev_gemm_nt per group, and nothing
    written to the right of a group's last column."""
    import torch
    from george_amd import _native as N
    ng, M, Nn = len(widths), group_rows * len(widths), max(widths)
    g = torch.Generator(device="cuda").manual_seed(ng * 7 + K)
    a = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    b = torch.randn(Nn, K, dtype=torch.float64, device="cuda", generator=g)
    c0 = torch.randn(M, Nn + 128, dtype=torch.float64, device="cuda", generator=g)     # (ldc > the widest group)
    c = c0.clone()
    w = (C.c_int64 * ng)(*widths)
    N.check(N.lib.gh_dev_gemm_nt_stair(c.data_ptr(), c.stride(0), a.data_ptr(), K, b.data_ptr(), K, group_rows, ng, w, K, None))
    ref = c0.clone()
    for q, wq in enumerate(widths):
        r = slice(q * group_rows, (q + 1) * group_rows)
        N.check(N.lib.gh_dev_gemm_nt(ref[r].data_ptr(), ref.stride(0), a[r].data_ptr(), K, b.data_ptr(), K, group_rows, wq, K, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(c, ref)
    for q, wq in enumerate(widths):
        r = slice(q * group_rows, (q + 1) * group_rows)
        assert torch.equal(c[r, wq:], c0[r, wq:])
        err = (c[r, :wq] - (c0[r, :wq] - a[r] @ b[:wq].T)).abs().max().item()
        assert err < 1e-10 * max(1.0, K / 16), (q, err)
    # arguments the launch refuses
    bad = (C.c_int64 * 2)(256, 128)
    assert N.lib.gh_dev_gemm_nt_stair(c.data_ptr(), c.stride(0), a.data_ptr(), K, b.data_ptr(), K, 128, 2, bad, K, None) != 0
    assert N.lib.gh_dev_gemm_nt_stair(c.data_ptr(), c.stride(0), a.data_ptr(), K, b.data_ptr(), K, 100, 1, w, K, None) != 0


@pytest.mark.parametrize("m,n,k,lower,alpha,beta", [(256, 384, 32, False, -1.0, 1.0), (640, 640, 1024, True, -1.0, 1.0), (128, 128, 64, False, 1.0, 0.0),
                                                    (1024, 128, 128, False, -1.0, 1.0), (2048, 2048, 96, True, -1.0, 1.0),
                                                    (4224, 4224, 1024, True, -1.0, 1.0), (512, 640, 4096, False, 2.5, -0.5),
                                                    (384, 256, 48, False, -1.0, 1.0)])
def test_half_slab_pipelined_kernel_gives_the_same_bits(m, n, k, lower, alpha, beta):
    """gemm_f64_mfma_dma_sp (the k-major x k-major kernel with the barrier in the middle of a slab, the default) against
    gemm_f64_mfma_dma (gh_debug_set_gemm_sp(0)): every accumulator adds the same k-steps in the same order -- identical
    results, bit for bit; k = 48: a shape the pipelined form does not take (odd number of slabs) runs the old kernel in
    both modes"""
    from george_amd import _native as N
    rng = np.random.RandomState(m + n + k)
    A = rng.randn(m, k)
    B = A if lower else rng.randn(n, k)
    C0 = rng.randn(m, n)
    outs = []
    try:
        for mode in (0, 1):
            N.lib.gh_debug_set_gemm_sp(mode)
            a, c = _dev(A), _dev(C0)
            b = a if lower else _dev(B)
            _gemm(c, a, b, m, n, k, alpha, beta, FLAGS["LOWER"] if lower else 0, k, k, n)
            outs.append(c.cpu().numpy())
    finally:
        N.lib.gh_debug_set_gemm_sp(-1)
    mask = np.kron(np.tril(np.ones((m // 128, n // 128))), np.ones((128, 128))).astype(bool) if lower else np.ones((m, n), bool)
    assert np.array_equal(outs[0][mask], outs[1][mask])
    want = beta * C0 + alpha * (A @ B.T)
    assert np.abs(outs[1] - want)[mask].max() <= 1e-13 * max(1.0, np.abs(A @ B.T).max())
