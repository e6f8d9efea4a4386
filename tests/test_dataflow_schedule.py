"""The task queues of the dataflow factorisation (george_amd/csrc/gh_dflow.hip) replayed on the host.

The device kernel stores no dependencies: a task's inputs follow from its fields and three families of counters
(D, rowh, kd).  This file restates that rule in Python (`ready`), hands the library's queues (gh_debug_dflow_schedule,
host only) to a pool of simulated workers that claim exactly as the kernel does -- queues 0 and 1 through ready lists that
whoever finishes a task's input fills (gh_debug_dflow_candidates), the others by a ticket from the first open
bucket, one held task per queue and worker -- with random task durations, and checks

  * every claim finds its inputs FINAL in the true state (not only in the counters),
  * no two tasks in flight touch the same half tile,
  * every k step reaches every tile exactly once, in ascending order,
  * the run never stalls (forward progress whatever the timing), every task is consumed,
  * the tiles, executed with NumPy on a small tile size, give the Cholesky factor.
"""
import ctypes as C
import random

import numpy as np
import pytest

from george_amd import _native

PW = 8
SCAN = 8
NQ = 5
NPEEK = 3
NLIST = 2


def schedule(nt):
    """-> per queue: (tasks, buckets); task = (i, j, k0, k1, half, fin); bucket = [start, size, gate_word, gate_val]"""
    lib = _native.lib
    counts = (C.c_int32 * NQ)()
    assert lib.gh_debug_dflow_schedule(nt, counts, None, 0) == 0
    tot = sum(counts)
    out = (C.c_int32 * (10 * max(tot, 1)))()
    assert lib.gh_debug_dflow_schedule(nt, counts, out, tot) == 0
    rows = np.frombuffer(out, dtype=np.int32).reshape(-1, 10)[:tot]
    qs = [([], []) for _ in range(NQ)]
    for r in rows:
        tasks, buckets = qs[r[0]]
        if r[7] < 0:
            tasks.append(tuple(int(v) for v in r[1:7]))
            continue
        if r[7] == len(buckets):
            buckets.append([len(tasks), 0, int(r[8]), int(r[9])])
        assert r[7] == len(buckets) - 1
        buckets[-1][1] += 1
        tasks.append(tuple(int(v) for v in r[1:7]))
    assert [len(q[0]) for q in qs] == list(counts)
    return qs


def candidates(nt, total):
    lib = _native.lib
    n = C.c_int64(0)
    assert lib.gh_debug_dflow_candidates(nt, None, None, 0, C.byref(n)) == 0
    ptr = (C.c_uint32 * (total + 2 * nt + 1))()
    cand = (C.c_uint32 * max(1, n.value))()
    assert lib.gh_debug_dflow_candidates(nt, ptr, cand, n.value, C.byref(n)) == 0
    ptr, cand = list(ptr), list(cand)[:n.value]
    assert ptr[0] == 0 and ptr[-1] == n.value
    return [cand[ptr[p]:ptr[p + 1]] for p in range(total + 2 * nt)]


class Machine:
    """True state + the counters the kernel sees + optional numerics on b x b tiles."""

    def __init__(self, nt, b=0, seed=0):
        self.nt, self.b = nt, b
        self.D = 0
        self.rowh = [[0, 0] for _ in range(nt)]
        self.kd = {}
        self.kd_true = {}
        self.final = set()          # (i, j, h) of final L half tiles; (j, j, 2) for a factored diagonal tile
        self.busy = set()
        if b:
            rng = np.random.RandomState(seed)
            n = nt * b
            m = rng.randn(n, n)
            self.A0 = m @ m.T + n * np.eye(n)
            self.A = np.tril(self.A0).copy()
            self.dinv = [None] * nt

    def k(self, i, j, h):
        return self.kd.get((i, j, h), 0)

    def halves(self, half):
        return (0, 1) if half == 2 else (half,)

    def word(self, w):
        """the counter array as the kernel's gates address it: 0 = D, 256 + 2 r + h = rowh[r, h]"""
        if w == 0:
            return self.D
        assert w >= 256 and (w - 256) // 2 < self.nt
        return self.rowh[(w - 256) // 2][(w - 256) % 2]

    # the kernel's df_ready()
    def ready(self, t):
        i, j, k0, k1, half, fin = t
        ok = all(self.k(i, j, h) >= k0 for h in self.halves(half)) if k0 else True
        if k1 > k0:
            ok &= self.rowh[j][0] >= k1 and self.rowh[j][1] >= k1
            if i != j:
                ok &= all(self.rowh[i][h] >= k1 for h in self.halves(half))
        if fin:
            ok &= self.D >= j + 1
        return ok

    def rows(self, i, h):
        b = self.b
        if h == 2:
            return slice(i * b, (i + 1) * b)
        return slice(i * b + h * b // 2, i * b + (h + 1) * b // 2)

    def start(self, t):
        """Checks against the TRUE state and the arithmetic, at claim time; returns the effects for completion."""
        i, j, k0, k1, half, fin = t
        hs = self.halves(half)
        for h in hs:
            assert (i, j, h) not in self.busy, ("two tasks on one half tile", t)
            assert (i, j, h) not in self.final, ("task on a final tile", t)
            assert self.kd_true.get((i, j, h), 0) == k0, ("k steps out of order", t, self.kd_true.get((i, j, h), 0))
            self.busy.add((i, j, h))
        for k in range(k0, k1):
            assert (j, k, 0) in self.final and (j, k, 1) in self.final, ("B operand not final", t, k)
            for h in hs:
                assert (i, k, h) in self.final, ("A operand not final", t, k)
        if fin:
            assert i >= j + 2
            assert k1 == j and (j, j, 2) in self.final, ("L_jj^-1 not there", t)
        new = None
        if self.b:
            b = self.b
            r = self.rows(i, half)
            c = self.A[r, j * b:(j + 1) * b].copy()
            if k1 > k0:
                c -= self.A[r, k0 * b:k1 * b] @ self.A[j * b:(j + 1) * b, k0 * b:k1 * b].T
            if fin:
                c = c @ self.dinv[j].T
            new = (r, j, c)
        return (t, new)

    def finish(self, eff):
        (i, j, k0, k1, half, fin), new = eff
        if new is not None:
            r, jj, c = new
            self.A[r, jj * self.b:(jj + 1) * self.b] = c
        for h in self.halves(half):
            self.busy.discard((i, j, h))
            self.kd_true[(i, j, h)] = k1
            if fin:                                  # (the kernel publishes the row counter only: nobody reads kd of a final tile)
                self.final.add((i, j, h))
                assert self.rowh[i][h] == j, ("row counter out of order", i, j, h, self.rowh[i][h])
                self.rowh[i][h] = j + 1
            else:
                self.kd[(i, j, h)] = k1

    # the diagonal worker, step j, in two parts (the sub-diagonal tile is published before the 128 x 128 kernel runs)
    def diag_can_start(self, j):
        if j <= 1:
            return True
        return all(self.k(j, j - 1, h) >= j - 1 and self.k(j, j, h) >= j - 1 for h in (0, 1))

    def diag_part1(self, j):
        if j == 0:
            return
        for h in (0, 1):
            assert self.kd_true.get((j, j - 1, h), 0) == j - 1 and self.kd_true.get((j, j, h), 0) == j - 1, ("diagonal worker too early", j)
            assert (j, j - 1, h) not in self.busy and (j, j, h) not in self.busy
        if self.b:
            b = self.b
            r = slice(j * b, (j + 1) * b)
            self.A[r, (j - 1) * b:j * b] = self.A[r, (j - 1) * b:j * b] @ self.dinv[j - 1].T
        for h in (0, 1):
            self.final.add((j, j - 1, h))
            assert self.rowh[j][h] == j - 1
            self.rowh[j][h] = j

    def diag_part2(self, j):
        if self.b:
            b = self.b
            r = slice(j * b, (j + 1) * b)
            if j > 0:
                sub = self.A[r, (j - 1) * b:j * b]
                self.A[r, r] -= sub @ sub.T
            ljj = np.linalg.cholesky(np.tril(self.A[r, r]) + np.tril(self.A[r, r], -1).T)
            self.A[r, r] = ljj
            self.dinv[j] = np.linalg.inv(ljj)
        self.final.add((j, j, 2))
        self.D = j + 1


def run(nt, nworkers, b=0, seed=0, max_flight=12):
    qs = schedule(nt)
    m = Machine(nt, b, seed)
    rng = random.Random(seed)
    hint = [0] * NQ
    nxt = [[0] * len(q[1]) for q in qs]          # ticket counters
    held = [[None] * NQ for _ in range(nworkers)]  # per worker and queue: a claimed task waiting for its inputs
    flight = [None] * nworkers                    # (remaining ticks, producer index, effects)
    dj, dphase, dwait = 0, 0, 0                   # diagonal worker: step, part, remaining ticks
    done_tasks = 0
    total = sum(len(q[0]) for q in qs)
    stall = 0
    qoff = [0]
    for q in range(NQ):
        qoff.append(qoff[-1] + len(qs[q][0]))
    gid = {}
    for q in range(NQ):
        for x, t in enumerate(qs[q][0]):
            gid[(q, x)] = qoff[q] + x
    cands = candidates(nt, total)
    listed = [[False] * len(qs[q][0]) for q in range(NLIST)]
    ready_list = [[] for _ in range(NLIST)]
    head = [0] * NLIST

    def list_candidates(producer):
        """what a producer does after publishing: the runnable ones of its candidates go onto their ready lists"""
        for g in cands[producer]:
            q = 1 if g >= qoff[1] else 0
            assert g < qoff[NLIST]
            x = g - qoff[q]
            if not listed[q][x] and m.ready(qs[q][0][x]):
                listed[q][x] = True
                ready_list[q].append(x)

    def claim(q):
        """the kernel's claim: from the first bucket that is not used up on, at most SCAN of them, none behind a closed D
        gate; crit and hi look at the bucket's next task first and take a ticket only when it is runnable"""
        tasks, buckets = qs[q]
        front, scan = True, 0
        bk = hint[q]
        while bk < len(buckets) and scan < SCAN:
            start, size, gw, gv = buckets[bk]
            if nxt[q][bk] >= size:
                if front:
                    hint[q] = max(hint[q], bk + 1)
                bk += 1
                continue
            front = False
            scan += 1
            if m.word(gw) < gv:
                if gw == 0:
                    break
                bk += 1
                continue
            if q < NPEEK and not m.ready(tasks[start + nxt[q][bk]]):
                bk += 1
                continue
            tk = nxt[q][bk]
            nxt[q][bk] += 1
            return start + tk
        return None

    while dj < nt or done_tasks < total:
        progressed = False
        if dj < nt:
            if dwait > 0:
                dwait -= 1
                progressed = True
            elif dphase == 0:
                if m.diag_can_start(dj):
                    m.diag_part1(dj)
                    if dj > 0:
                        list_candidates(total + dj)
                    dphase, dwait = 1, rng.randint(0, 3)
                    progressed = True
            else:
                m.diag_part2(dj)
                list_candidates(total + nt + dj)
                dj, dphase, dwait = dj + 1, 0, rng.randint(0, 3)
                progressed = True
        order = list(range(nworkers))
        rng.shuffle(order)
        for w in order:
            if flight[w] is not None:
                rem, prod, eff = flight[w]
                if rem > 0:
                    flight[w] = (rem - 1, prod, eff)
                else:
                    m.finish(eff)
                    list_candidates(prod)
                    flight[w] = None
                    done_tasks += 1
                progressed = True
                continue
            if rng.random() < 0.3:       # this worker does not poll in this tick
                continue
            started = False
            for q in range(NLIST):       # the ready lists first
                if head[q] < len(ready_list[q]):
                    x = ready_list[q][head[q]]
                    head[q] += 1
                    assert m.ready(qs[q][0][x])
                    flight[w] = (rng.randint(0, max_flight), gid[(q, x)], m.start(qs[q][0][x]))
                    progressed = started = True
                    break
            if started:
                continue
            for q in range(NLIST, NQ):
                if held[w][q] is None:
                    held[w][q] = claim(q)
                    if held[w][q] is not None:
                        progressed = True
                if held[w][q] is not None and m.ready(qs[q][0][held[w][q]]):
                    x = held[w][q]
                    flight[w] = (rng.randint(0, max_flight), gid[(q, x)], m.start(qs[q][0][x]))
                    held[w][q] = None
                    progressed = True
                    break
        stall = 0 if progressed else stall + 1
        assert stall < 50, ("no forward progress", dj, hint, [[t for t in h if t] for h in held if any(h)][:5])
    assert all(h == [None] * NQ for h in held)
    for q in range(NLIST):
        assert all(listed[q]) and head[q] == len(qs[q][0])
    for q in range(NLIST, NQ):
        assert all(nxt[q][bk] >= qs[q][1][bk][1] for bk in range(len(qs[q][1])))
    return m, [q[0] for q in qs]


def check_coverage(nt, m):
    for j in range(nt):
        for i in range(j, nt):
            want = max(0, j - 1 if i == j else j)      # the workers' share
            for h in (0, 1):
                assert m.kd_true.get((i, j, h), 0) == want, (i, j, h)
                if i >= j + 1:
                    assert (i, j, h) in m.final
        assert (j, j, 2) in m.final


@pytest.mark.parametrize("nt,workers,seed", [(1, 3, 0), (2, 3, 1), (3, 2, 2), (9, 4, 3), (17, 1, 4), (20, 7, 5), (26, 40, 6), (33, 13, 7)])
def test_replay_gives_the_cholesky_factor(nt, workers, seed):
    b = 4
    m, _ = run(nt, workers, b=b, seed=seed)
    check_coverage(nt, m)
    ref = np.linalg.cholesky(m.A0)
    got = np.tril(m.A)
    assert np.allclose(got, ref, rtol=1e-10, atol=1e-10 * np.abs(ref).max())


@pytest.mark.parametrize("nt,workers", [(64, 64), (128, 200)])
def test_structure_at_product_sizes(nt, workers):
    m, qs = run(nt, workers, b=0, seed=nt, max_flight=6)
    check_coverage(nt, m)
    # flops: every tile product exactly once
    steps = sum((t[3] - t[2]) * (2 if t[4] == 2 else 1) for q in qs for t in q)
    want = sum(2 * max(0, j - 1 if i == j else j) for j in range(nt) for i in range(j, nt))
    assert steps == want


def test_queue_shapes():
    qs = schedule(72)
    # lo (queues 3, 4): whole tiles, one pass per panel older than the previous one; one bucket each, always open
    assert len(qs[3][1]) == 1 and qs[3][1][0][3] == 0 and len(qs[4][1]) == 1
    for (i, j, k0, k1, half, fin) in qs[3][0] + qs[4][0]:
        assert half == 2 and not fin and 0 < k1 - k0 <= PW and k1 <= PW * (j // PW - 1) and i >= PW * (j // PW) + 16
    # crit and the next-block queue: one k step or a multiply; the next-block queue also has the old panels' passes into the
    # next diagonal blocks' tiles
    for q in (0, 1):
        for (i, j, k0, k1, half, fin) in qs[q][0]:
            assert (k1 - k0 == 1 and not fin) or (k1 == k0 == j and fin) or (q == 1 and not fin and k1 - k0 == PW and k1 <= PW * (j // PW - 1) and i < PW * (j // PW) + 16)
    # multiplies by L_jj^-T never for the diagonal worker's rows; D gates open in order inside a queue
    for tasks, buckets in qs:
        for (i, j, k0, k1, half, fin) in tasks:
            assert i >= j and (not fin or (i >= j + 2 and k1 == j))
        d_gates = [b[3] for b in buckets if b[2] == 0]
        assert d_gates == sorted(d_gates)
        assert (not buckets) or sum(b[1] for b in buckets) == len(tasks)
    assert not qs[0][1] and not qs[1][1] and qs[2][1]            # queues 0, 1 are listed, the others have buckets
