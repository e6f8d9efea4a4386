#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text files committed under profiles/.

usage: summarize_prof.py <results.db> [<out.md>]      per-kernel time table (+ PMC sums if present)
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, "
        "min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    out.write("| kernel | calls | total ms | %% | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        out.write("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |\n" % (r[0][:110], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
    out.write("\ntotal kernel time: %.3f ms\n" % tot)
    # The trailing SYRK updates (bench.py's roofline kernel) are the launches of the LOWER
    # instantiation gemm_f64_mfma_dma<true, true, true> (triangular grid of t(t+1)/2 workgroups)
    # with t >= tmin = panel width / 128: the in-panel updates are triangular too, but smaller.
    if len(sys.argv) > 3:
        tmin = int(sys.argv[3])
        g = list(cur.execute(
            "select d.grid_size_x / d.workgroup_size_x, d.end - d.start, s.kernel_name "
            "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
            "where s.kernel_name like '%gemm_f64_mfma%'"))
        # (round 5: the k-major x k-major launches are gemm_f64_mfma_dma_sp<LOWER>)
        is_tr = lambda grid, name: ("dmaILb1ELb1ELb1EE" in name or "dma_spILb1EE" in name) and grid >= tmin * (tmin + 1) // 2
        tr = [dur for grid, dur, name in g if is_tr(grid, name)]
        ot = [dur for grid, dur, name in g if not is_tr(grid, name)]
        out.write("\n`gemm_f64_mfma*` dispatches by role (trailing update <=> the LOWER instantiation with a grid of "
                  "t(t+1)/2 workgroups, t >= %d):\n\n" % tmin)
        out.write("| role | launches | total ms | avg ms |\n|---|---|---|---|\n")
        for name, v in (("trailing SYRK update", tr), ("panel / block-column GEMMs", ot)):
            if v:
                out.write("| %s | %d | %.3f | %.3f |\n" % (name, len(v), sum(v) / 1e6, sum(v) / 1e6 / len(v)))
    # PMC counters (if this was a --pmc run)
    try:
        pm = list(cur.execute(
            "select s.kernel_name, p.name, count(*), sum(e.value) "
            "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
            "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
            "group by s.kernel_name, p.name order by 4 desc"))
    except sqlite3.Error as exc:
        pm = []
        out.write("\n(no PMC tables: %s)\n" % exc)
    if pm:
        out.write("\n| kernel | counter | dispatches | sum | per dispatch |\n|---|---|---|---|---|\n")
        for k, name, n, v in pm:
            out.write("| `%s` | %s | %d | %.6g | %.6g |\n" % (k[:90], name, n, v, v / max(n, 1)))


if __name__ == "__main__":
    main()
