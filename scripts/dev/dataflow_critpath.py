"""Walks the dataflow trace (scripts/dev/dataflow_trace.py) backwards from the diagonal worker's longest waits: for the task
that ended a wait, when were its inputs there (the end of the tasks that produced them), how long after that did it start
(nobody took it = a scheduling problem) and which input was the last -- then the same for that producer, down the chain.
python scripts/dev/dataflow_critpath.py gpurun_out/dflow_trace_N16384.npy [how many waits]"""
import sys

import numpy as np


def main():
    rec = np.load(sys.argv[1])
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    kind = (rec[:, 3] & 0xff).astype(int)
    fin = ((rec[:, 3] >> 16) & 0xff).astype(int)
    half = ((rec[:, 3] >> 8) & 0xff).astype(int)
    ti = (rec[:, 2] & 0xffff).astype(int)
    tj = ((rec[:, 2] >> 16) & 0xffff).astype(int)
    k0 = ((rec[:, 2] >> 32) & 0xffff).astype(int)
    k1 = ((rec[:, 2] >> 48) & 0xffff).astype(int)
    base = t0.min()
    us = lambda t: (t - base) / 100.0
    # producers: kd[(i, j)][k] = record that brought tile (i, j) to k steps; fin_of[(i, j)] = record(s) that made L(i, j) final
    by_tile, fin_of, potf2_end, tpub = {}, {}, {}, {}
    for r in range(len(rec)):
        if kind[r] < 8:
            by_tile.setdefault((ti[r], tj[r]), []).append(r)
            if fin[r]:
                fin_of.setdefault((ti[r], tj[r]), []).append(r)
        elif kind[r] == 11:
            potf2_end[ti[r]] = r
        elif kind[r] == 9:
            tpub[(ti[r], tj[r])] = r            # L(j, j-1) by the diagonal worker

    def name(r):
        if kind[r] < 8:
            return "q%d (%d,%d) k[%d,%d)%s%s" % (kind[r], ti[r], tj[r], k0[r], k1[r], " h%d" % half[r] if half[r] < 2 else "", " fin" if fin[r] else "")
        return {8: "diag wait", 9: "diag T", 10: "diag U", 11: "potf2"}[kind[r]] + " j=%d" % ti[r]

    def final_rec(i, j):
        """record that made L(i, j) final (the later half)"""
        if i == j + 1:
            return tpub.get((i, j))
        rs = fin_of.get((i, j))
        return max(rs, key=lambda r: t1[r]) if rs else None

    def inputs(r):
        """[(label, producer record)] of task r"""
        out = []
        if kind[r] >= 8:
            return out
        i, j = ti[r], tj[r]
        if k0[r] > 0:
            prev = [x for x in by_tile.get((i, j), []) if k1[x] == k0[r] and not fin[x] and (half[r] == 2 or half[x] == 2 or half[x] == half[r])]
            if prev:
                out.append(("same tile", max(prev, key=lambda x: t1[x])))
        if k1[r] > k0[r]:
            a = final_rec(i, k1[r] - 1) if i != j else None
            b = final_rec(j, k1[r] - 1)
            if a is not None:
                out.append(("A = L(%d,%d)" % (i, k1[r] - 1), a))
            if b is not None:
                out.append(("B = L(%d,%d)" % (j, k1[r] - 1), b))
        if fin[r] and j in potf2_end:
            out.append(("L_%d%d^-1" % (j, j), potf2_end[j]))
        return out

    waits = np.where(kind == 8)[0]
    waits = waits[np.argsort(-(t1[waits] - t0[waits]))][:nw]
    for w in waits:
        j = ti[w]
        print("=== diagonal worker waited %.0f us at step %d (from %.0f us)" % ((t1[w] - t0[w]) / 100.0, j, us(t0[w])))
        cands = [x for x in by_tile.get((j, j), []) + by_tile.get((j, j - 1), []) if not fin[x]]
        r = max(cands, key=lambda x: t1[x])
        for depth in range(14):
            ins = inputs(r)
            if not ins:
                print("   %-34s %9.0f -> %9.0f" % (name(r), us(t0[r]), us(t1[r])))
                break
            lab, last = max(ins, key=lambda e: t1[e[1]])
            lag = (t0[r] - t1[last]) / 100.0
            print("   %-34s %9.0f -> %9.0f  (%5.0f us)  inputs ready at %9.0f, started %6.0f us later; last input: %s" %
                  (name(r), us(t0[r]), us(t1[r]), (t1[r] - t0[r]) / 100.0, us(t1[last]), lag, lab))
            r = last


if __name__ == "__main__":
    main()
