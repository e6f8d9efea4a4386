"""The workgroup -> tile maps of george_amd/csrc/gh_gemm.hip (tile_of: XCD remap, row groups walked column-major over a triangle /
a rectangle, the staircase of gh_dev_gemm_nt_stair), restated in Python with the same integer arithmetic: every map must visit every
tile of its shape exactly once, for any shape -- no padded grids.  (The kernels themselves are checked on the GPU:
tests/test_gpu_gemm.py::test_grouped_tile_order, test_staircase_gemm.)"""
import math

import pytest

G = 8            # GH_TILE_GROUP


def xcd_remap(bid, nblk):
    q, r = divmod(nblk, 8)
    xcd, idx = bid % 8, bid // 8
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx


def grouped_rect(l, rows, cols):
    g = l // (G * cols)
    gg = min(G, rows - g * G)
    r = l - g * G * cols
    return g * G + r % gg, r // gg


def grouped_tri(l, T):
    H = G * (G + 1) // 2
    start = lambda q: G * G * q * (q - 1) // 2 + q * H
    c = H - 0.5 * G * G
    g = int((math.sqrt(c * c + 2.0 * G * G * l) - c) / (G * G))
    while start(g) > l:
        g -= 1
    while start(g + 1) <= l:
        g += 1
    r0 = g * G
    gg = min(G, T - r0)
    r = l - start(g)
    rect = gg * (r0 + 1)
    if r < rect:
        return r0 + r % gg, r // gg
    r -= rect
    j = 1
    while r >= gg - j:
        r -= gg - j
        j += 1
    return r0 + j + r, r0 + j


def trapezoid(l, tiles_m, tiles_n):
    tri = tiles_n * (tiles_n + 1) // 2
    if l < tri:
        return grouped_tri(l, tiles_n)
    tm, tn = grouped_rect(l - tri, tiles_m - tiles_n, tiles_n)
    return tm + tiles_n, tn


def stair_table(h, widths):                      # gh_launch_gemm: widest group first
    pre, at = [], 0
    for s in range(len(widths)):
        pre.append(at)
        at += h * widths[len(widths) - 1 - s]
    return pre + [at]


def stair(l, h, pre):
    n = len(pre) - 1
    s = 0
    while s + 1 < n and l >= pre[s + 1]:
        s += 1
    r = l - pre[s]
    return (n - 1 - s) * h + r % h, r // h


@pytest.mark.parametrize("nblk", [1, 7, 8, 9, 63, 64, 65, 1000, 4097])
def test_xcd_remap_is_a_permutation(nblk):
    assert sorted(xcd_remap(b, nblk) for b in range(nblk)) == list(range(nblk))
    # each XCD (bid % 8) gets one contiguous range of logical tiles
    for x in range(min(8, nblk)):
        mine = sorted(xcd_remap(b, nblk) for b in range(x, nblk, 8))
        assert mine == list(range(mine[0], mine[0] + len(mine)))


@pytest.mark.parametrize("T", [1, 2, 7, 8, 9, 15, 16, 17, 100, 257, 512])
def test_grouped_triangle(T):
    n = T * (T + 1) // 2
    got = [grouped_tri(l, T) for l in range(n)]
    assert set(got) == {(m, k) for m in range(T) for k in range(m + 1)} and len(got) == n
    # 64 consecutive tiles away from the diagonal touch at most 8 tile rows and 9 tile columns
    if T >= 64:
        l0 = (G * G * 5 * 4 // 2 + 5 * G * (G + 1) // 2)          # start of row group 5
        blk = got[l0:l0 + 64]
        assert len({m for m, _ in blk}) <= 8 and len({k for _, k in blk}) <= 9


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (8, 8), (9, 3), (112, 8), (17, 33), (8, 512), (1, 100)])
def test_grouped_rectangle(rows, cols):
    got = [grouped_rect(l, rows, cols) for l in range(rows * cols)]
    assert set(got) == {(m, k) for m in range(rows) for k in range(cols)} and len(got) == rows * cols


@pytest.mark.parametrize("tiles_m,tiles_n", [(5, 2), (9, 9), (12, 1), (40, 7), (137, 129), (64, 8)])
def test_lower_trapezoid(tiles_m, tiles_n):
    n = tiles_n * (tiles_n + 1) // 2 + (tiles_m - tiles_n) * tiles_n
    got = [trapezoid(l, tiles_m, tiles_n) for l in range(n)]
    assert set(got) == {(m, k) for m in range(tiles_m) for k in range(min(m + 1, tiles_n))} and len(got) == n


@pytest.mark.parametrize("h,widths", [(8, [8, 24, 24, 72, 128, 136]), (1, list(range(1, 41))), (2, [1, 1, 1]), (8, [512] * 8),
                                       (8, [8 * (1 + i) for i in range(64)])])
def test_staircase(h, widths):
    pre = stair_table(h, widths)
    got = [stair(l, h, pre) for l in range(pre[-1])]
    want = {(g * h + r, k) for g, w in enumerate(widths) for r in range(h) for k in range(w)}
    assert set(got) == want and len(got) == len(want)
    assert got[0][0] // h == len(widths) - 1                       # the widest group is walked first
