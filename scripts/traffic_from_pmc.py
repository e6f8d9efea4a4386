#!/usr/bin/env python
"""HBM/fabric traffic of the trailing-update (SYRK) launches from two rocprofv3 --pmc passes.

usage: traffic_from_pmc.py <fetch.db> <write.db> <N> <nb> [<out.json>] [--uniform]
(--uniform: panels of nb columns throughout; default: the solver's adaptive widths, round 6)

Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
passes (they do not fit one TCC pass); both are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced read stream -- confirmed here on a 16-B/lane copy of known size
(profiles/r01/pmc_calibration_*: 2 GiB read -> FETCH_SIZE 1 048 590 KiB, 2 GiB written ->
WRITE_SIZE 2 097 150 KiB) -- so  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The trailing launches are the dispatches of the LOWER instantiation gemm_f64_mfma_dma<1,1,1> whose
grid is one of the factorisation's trailing sizes (a triangular number of 128x128 tiles).  Infinity-Cache hits are counted by these counters, so
this is traffic leaving the L2s, an upper bound on HBM traffic.
"""
import json
import sqlite3
import sys


def panel_starts(np_, nb, adaptive=True, wide_min_trailing=25600):
    """gh_chol.hip, panel_starts(): with the width left to the solver (nb = 1024 here, adaptive) the panels are 2 nb wide while
    more than `wide_min_trailing` columns of trailing matrix lie behind them"""
    pc, k0 = [], 0
    while k0 < np_:
        pc.append(k0)
        w = 2 * nb if (adaptive and np_ - (k0 + 2 * nb) >= wide_min_trailing) else nb
        k0 += min(w, np_ - k0)
    pc.append(np_)
    return pc


def trailing_grids(n, nb, adaptive=True, single_stream=True):
    """{workgroups of a wide trailing launch: (tile rows m2, K of the launch)}: W(j) updates columns from panel j + 2 on"""
    np_ = (n + 127) // 128 * 128
    pc = panel_starts(np_, nb, adaptive)
    grids = {}
    ahead = 1 if single_stream else 2               # W(j) starts at panel j + 2 under look-ahead, at j + 1 without
    for j in range(len(pc) - 1):
        if j + ahead > len(pc) - 2:
            break
        m2 = (np_ - pc[j + ahead]) // 128
        if m2 > 0:
            grids[m2 * (m2 + 1) // 2] = (m2, pc[j + 1] - pc[j])
    return grids


def collect(db, counter, grids):
    con = sqlite3.connect(db)
    cur = con.cursor()
    q = ("select d.dispatch_id, d.grid_size_x/256, d.end-d.start, e.value from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "where (s.kernel_name like '%gemm_f64_mfma_dmaILb1ELb1ELb1EE%' or s.kernel_name like '%gemm_f64_mfma_dma_spILb1EE%') and p.name = ?")
    per = {}
    for did, g, dt, v in cur.execute(q, (counter,)):
        if g in grids and g > 36:
            e = per.setdefault(did, [g, dt, 0.0])
            e[2] += v
    return per


def main():
    fdb, wdb, n, nb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    grids = trailing_grids(n, nb, adaptive="--uniform" not in sys.argv)
    f = collect(fdb, "FETCH_SIZE", grids)
    w = collect(wdb, "WRITE_SIZE", grids)
    nf, nw = len(f), len(w)
    fetch_kib = sum(v[2] for v in f.values())
    write_kib = sum(v[2] for v in w.values())
    alg = 0.0
    for did, (g, dt, _) in f.items():
        m2, b = grids[g]
        alg += g * 2.0 * 128 * 128 * 8 + m2 * 128.0 * b * 8      # C read + write, panel read once
    out = {
        "N": n, "nb": nb, "launches_fetch_pass": nf, "launches_write_pass": nw,
        "FETCH_SIZE_KiB_sum": fetch_kib, "WRITE_SIZE_KiB_sum": write_kib,
        "bytes_per_launch": ((2.0 * fetch_kib / max(nf, 1)) + write_kib / max(nw, 1)) * 1024.0,
        "read_bytes_per_launch": 2.0 * fetch_kib / max(nf, 1) * 1024.0,
        "write_bytes_per_launch": write_kib / max(nw, 1) * 1024.0,
        "algorithmic_bytes_per_launch": alg / max(nf, 1),
        "avg_launch_ms_under_pmc": sum(v[1] for v in f.values()) / max(nf, 1) / 1e6,
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 of every 128 B)",
    }
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
