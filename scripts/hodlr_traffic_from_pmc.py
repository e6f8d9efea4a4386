#!/usr/bin/env python
"""Traffic leaving the L2s during one HODLR compute()+log_likelihood() (BASELINE config C4) from two rocprofv3 --pmc passes.

usage: hodlr_traffic_from_pmc.py <fetch.db> <write.db> <n> <steps> [<out.json>]

The profiled command is `python scripts/hodlr_traffic_from_pmc.py --job <n> <steps>`: <steps> identical steps and nothing else,
so every counter is summed over ALL dispatches of the process and divided by <steps>.  Counter units and the gfx950 correction are
those of scripts/traffic_from_pmc.py (bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE and WRITE_SIZE in separate passes).
Infinity-Cache hits are counted, so this is an upper bound on HBM traffic.
"""
import json
import os
import sqlite3
import sys


def job(n, steps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    j = bench.HodlrJob(n, 0)
    el, ll = bench.run_timed(j, steps, 0, lambda: None)
    print("hodlr job: %d steps, %.3f ms per step, log-likelihood %r" % (steps, el / steps * 1e3, ll))
    j.close()


def total(db, counter):
    con = sqlite3.connect(db)
    q = ("select s.kernel_name, count(distinct d.dispatch_id), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by s.kernel_name")
    rows = list(con.execute(q, (counter,)))
    return sum(r[2] for r in rows), {r[0][:48]: [r[1], r[2]] for r in rows}


def main():
    if sys.argv[1] == "--job":
        return job(int(sys.argv[2]), int(sys.argv[3]))
    fdb, wdb, n, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    f, fk = total(fdb, "FETCH_SIZE")
    w, wk = total(wdb, "WRITE_SIZE")
    top = sorted(set(fk) | set(wk), key=lambda k: -(2.0 * fk.get(k, [0, 0])[1] + wk.get(k, [0, 0])[1]))[:8]
    out = {"N": n, "steps": steps, "FETCH_SIZE_KiB_sum": f, "WRITE_SIZE_KiB_sum": w,
           "read_bytes_per_step": 2.0 * f * 1024.0 / steps, "write_bytes_per_step": w * 1024.0 / steps,
           "bytes_per_step": (2.0 * f + w) * 1024.0 / steps,
           "largest_kernels_GB_per_step": {k: round((2.0 * fk.get(k, [0, 0])[1] + wk.get(k, [0, 0])[1]) * 1024.0 / steps * 1e-9, 3) for k in top},
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 of every 128 B)"}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
