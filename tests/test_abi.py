"""The C-ABI library loads and exports every symbol include/george_amd.h declares (CPU only:
no compute call is made)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="george_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from george_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libgeorge_amd.so does not export %s" % n
    # the validation switches, probes and micro-benchmarks live in a header of their own: none of them is in the boundary
    dbg = _declared("george_amd_debug.h")
    # (gh_mgpu_get_trace: the read-out of the sharded solver's timing aid GH_MGPU_TRACE, moved there with the flag in round 5)
    assert dbg and all(n.startswith(("gh_debug_", "gh_microbench_")) or n == "gh_mgpu_get_trace" for n in dbg)
    assert not [n for n in names if n.startswith(("gh_debug_", "gh_microbench_"))]
    for n in dbg:
        assert hasattr(lib, n), "libgeorge_amd.so does not export %s" % n
    # and the ctypes signature table covers exactly the two headers
    assert sorted(_native.SIGNATURES) == sorted(names + dbg)


def test_struct_layout_matches_header():
    from george_amd import _native as N
    # gh_knode: 8 int32 + 8 int32 axes + (4 + 1 + 36 + 8 + 8) doubles
    assert ctypes.sizeof(N.gh_knode) == 16 * 4 + 57 * 8
    assert ctypes.sizeof(N.gh_chol_opts) == 8 * 4
    assert ctypes.sizeof(N.gh_chol_profile) == 11 * 8
    assert ctypes.sizeof(N.gh_hodlr_opts) == 4 * 4 + 8 + 4 * 4
    assert ctypes.sizeof(N.gh_mgpu_opts) == (1 + 16 + 2 + 1 + 1 + 4) * 4
    assert ctypes.sizeof(N.gh_hodlr_mgpu_opts) == (1 + 16 + 3) * 4 + 8 + 4 * 4
    assert N.gh_hodlr_mgpu_opts.tol.offset == 80


def test_program_validation_runs_without_gpu():
    """gh_kernel_create is host code: flattening + validation + error mapping work on CPU."""
    import george_amd.kernels as K
    from george_amd import program
    k = 12. * K.ExpSquaredKernel(0.4, ndim=3) + 0.1
    dk = program.DeviceKernel(k)
    assert (dk.ndim, dk.size) == (3, 3)
    arr = program.flatten(k)
    assert [n.op for n in arr] == [0, 0, 0, 2, 1]          # postfix: c, c, expsq, *, +
    with pytest.raises(ValueError):
        program.flatten(object())                          # "invalid kernel", parser.h:16
    with pytest.raises(ValueError):
        K.ExpSquaredKernel(1.0, ndim=2) + K.ExpSquaredKernel(1.0, ndim=3)   # dimension mismatch at build time


def test_multi_gpu_entry_points_declared_and_validated():
    """the sharded dense solver sits behind the C ABI (gh_mgpu_*).  Argument validation is
    host code and runs without a GPU: bad device counts / grids are GH_ERR_BAD_ARG (ValueError), and with
    no device visible creation fails loudly (RuntimeError), never a silent single-GPU or CPU path."""
    from george_amd import _native as N, MultiGPUSolver
    for name in ("gh_mgpu_create", "gh_mgpu_destroy", "gh_mgpu_compute", "gh_mgpu_info", "gh_mgpu_grid",
                 "gh_mgpu_dot_solve", "gh_mgpu_solve", "gh_mgpu_apply_sqrt", "gh_mgpu_get_inverse", "gh_mgpu_predict",
                 "gh_mgpu_owner", "gh_mgpu_get_trace", "gh_dev_trsv_lower_t"):
        assert name in N.SIGNATURES and hasattr(N.lib, name)
    o = N.gh_mgpu_opts()
    h = N._vp()
    o.n_dev = 0
    with pytest.raises(ValueError):
        N.check(N.lib.gh_mgpu_create(ctypes.byref(o), ctypes.byref(h)))
    o.n_dev, o.nb = 2, 100
    with pytest.raises(ValueError):
        N.check(N.lib.gh_mgpu_create(ctypes.byref(o), ctypes.byref(h)))
    with pytest.raises(ValueError):
        MultiGPUSolver(None, devices=[], transport="rccl")
    with pytest.raises(ValueError):
        MultiGPUSolver(None, devices=[0], transport="smoke signals")
    import george_amd
    if george_amd.device_count() == 0:
        o.n_dev, o.nb = 1, 0
        with pytest.raises(RuntimeError):
            N.check(N.lib.gh_mgpu_create(ctypes.byref(o), ctypes.byref(h)))


def test_hodlr_split_entry_points_declared_and_validated():
    """the HODLR tree split over several devices sits behind the C ABI (gh_hodlr_mgpu_*).  Validation
    is host code: device counts that are not a power of two are GH_ERR_BAD_ARG, and with no device visible
    creation fails loudly."""
    from george_amd import _native as N, MultiGPUHODLRSolver
    for name in ("gh_hodlr_mgpu_create", "gh_hodlr_mgpu_destroy", "gh_hodlr_mgpu_compute", "gh_hodlr_mgpu_solve",
                 "gh_hodlr_mgpu_dot_solve", "gh_hodlr_mgpu_ranks", "gh_hodlr_mgpu_rows"):
        assert name in N.SIGNATURES and hasattr(N.lib, name)
    o = N.gh_hodlr_mgpu_opts()
    h = N._vp()
    for bad in (0, 3, 6, 32):
        o.n_dev = bad
        with pytest.raises(ValueError):
            N.check(N.lib.gh_hodlr_mgpu_create(ctypes.byref(o), ctypes.byref(h)))
    with pytest.raises(ValueError):
        MultiGPUHODLRSolver(None, devices=[0, 0, 0])
    s = MultiGPUHODLRSolver(None, devices=[0, 0])
    with pytest.raises(RuntimeError):
        s.dot_solve(np.zeros(3))                           # "you must call 'compute' first"
    with pytest.raises(NotImplementedError):
        s.apply_sqrt(np.zeros(3))
    import george_amd
    if george_amd.device_count() == 0:
        o.n_dev = 2
        with pytest.raises(RuntimeError):
            N.check(N.lib.gh_hodlr_mgpu_create(ctypes.byref(o), ctypes.byref(h)))


def test_fails_loudly_without_gpu():
    import george_amd
    if george_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    k = george_amd.kernels.ExpSquaredKernel(1.0)
    with pytest.raises(RuntimeError):
        k.get_value(np.zeros((4, 1)))
    gp = george_amd.GP(k)
    with pytest.raises(RuntimeError):
        gp.compute(np.arange(4.0), 0.1)


def _reference_tree_levels(n, min_size):
    """hodlr.h:47-64: a node of `size` points is internal iff size // 2 >= min_size; children (start, half) and
    (start + half, size - half).  Returns, per level, the (start, size, internal) triples left to right."""
    levels, cur = [], [(0, n)]
    while cur:
        lvl, nxt = [], []
        for start, size in cur:
            half = size // 2
            internal = half >= min_size
            lvl.append((start, size, internal))
            if internal:
                nxt += [(start, half), (start + half, size - half)]
        levels.append(lvl)
        cur = nxt
    return levels


@pytest.mark.parametrize("n,n_dev,min_size", [(262144, 8, 100), (3000, 4, 64), (1000, 2, 100), (8191, 8, 100), (100000, 16, 37),
                                              (4096, 1, 100), (801, 4, 100)])
def test_split_layout_follows_the_reference_tree(n, n_dev, min_size):
    """gh_hodlr_mgpu_layout is host code: the sub-trees of the split are the nodes of level log2(n_dev) of the reference's
    tree, and the seed offsets are the positions of their internal nodes in the global levels (what makes the split draw
    the same pivot rows as the single-GPU solver)."""
    from george_amd import _native as N
    levels = _reference_tree_levels(n, min_size)
    depth = n_dev.bit_length() - 1
    row0, nrows = (ctypes.c_int64 * 16)(), (ctypes.c_int64 * 16)()
    nl = ctypes.c_int32(0)
    N.check(N.lib.gh_hodlr_mgpu_layout(n, n_dev, min_size, row0, nrows, None, 0, ctypes.byref(nl)))
    subs = levels[depth]
    assert len(subs) == n_dev
    assert [(row0[p], nrows[p]) for p in range(n_dev)] == [(s, z) for s, z, _ in subs]
    assert sum(nrows[p] for p in range(n_dev)) == n
    L = max(nl.value, 1)
    so = (ctypes.c_int32 * (n_dev * L))()
    N.check(N.lib.gh_hodlr_mgpu_layout(n, n_dev, min_size, row0, nrows, so, L, ctypes.byref(nl)))
    for lv in range(nl.value):
        glob = [(s, z) for s, z, internal in levels[depth + lv] if internal]          # internal nodes of the global level, in order
        for p in range(n_dev):
            lo, hi = row0[p], row0[p] + nrows[p]
            mine = [i for i, (s, z) in enumerate(glob) if lo <= s < hi]
            if mine:
                assert so[p * L + lv] == mine[0], (p, lv)
            else:                                                                      # no internal node of its own there: the count before it
                assert so[p * L + lv] == sum(1 for s, z in glob if s < lo)
    assert nl.value == max(0, len([lvl for lvl in levels[depth:] if any(i for _, _, i in lvl)]))


def test_split_layout_rejects_a_leaf_above_the_split():
    from george_amd import _native as N
    row0, nrows = (ctypes.c_int64 * 16)(), (ctypes.c_int64 * 16)()
    nl = ctypes.c_int32(0)
    with pytest.raises(ValueError):
        N.check(N.lib.gh_hodlr_mgpu_layout(700, 8, 100, row0, nrows, None, 0, ctypes.byref(nl)))     # level 2: 175 points, half 87
    with pytest.raises(ValueError):
        N.check(N.lib.gh_hodlr_mgpu_layout(4096, 3, 100, row0, nrows, None, 0, ctypes.byref(nl)))


def test_stand_in_rccl_exports_what_gh_mgpu_binds():
    """tests/mock_rccl/libmock_rccl.so (test infrastructure of tests/test_gpu_mgpu_mock_rccl.py, built by __graft_entry__.build()):
    the eight entry points george_amd/csrc/gh_mgpu.hip resolves with dlsym, the marker that lifts "one rank per device", and
    the two read-outs; and the product library exposes which communicator mode a sharded solver runs in."""
    path = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    src = open(os.path.join(ROOT, "george_amd", "csrc", "gh_mgpu.hip")).read()
    bound = re.findall(r'MG_SYM\([A-Za-z]+, "(nccl[A-Za-z]+)"\)', src)
    assert sorted(bound) == sorted(["ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                                    "ncclAllReduce", "ncclGetErrorString"])
    for name in bound + ["ncclMockSharedDeviceOk", "mock_rccl_stats", "mock_rccl_last_error"]:
        assert hasattr(lib, name), name
    st = (ctypes.c_longlong * 8)()
    lib.mock_rccl_stats(st)
    assert list(st) == [0] * 8                                   # nothing has run: no HIP call is made by loading it
    from george_amd import _native
    assert "gh_mgpu_comm_mode" in _native.SIGNATURES and _native.GH_MGPU_ONE_COMM == 8
