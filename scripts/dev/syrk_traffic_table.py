"""Decode the dispatches of scripts/dev/syrk_traffic_ab.py (REPS=1) from a rocprofv3 --pmc <counter> --kernel-trace database:
one row per arm with the counter's sum over its instances.  python scripts/dev/syrk_traffic_table.py fetch.db write.db"""
import sqlite3, sys
SHAPES = [(32768, 1024), (32768, 2048), (57344, 1024)]
ARMS = [(0, 0), (0, 1), (1, 0), (1, 1)]

def collect(db, counter):
    con = sqlite3.connect(db); cur = con.cursor()
    q = ("select d.dispatch_id, s.kernel_name, d.grid_size_x/256, d.end-d.start, sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%gemm_f64_mfma_dma%' and p.name = ? "
         "group by d.dispatch_id order by d.dispatch_id")
    return [r for r in cur.execute(q, (counter,)) if r[2] > 1000]

f = collect(sys.argv[1], "FETCH_SIZE"); w = collect(sys.argv[2], "WRITE_SIZE")
print("| M | K | sp | grouped | kernel | ms (under the counter pass) | read GB | written GB | total / algorithmic |\n|---|---|---|---|---|---|---|---|---|")
i = 0
for (m, k) in SHAPES:
    tiles = (m // 128) * (m // 128 + 1) // 2
    alg = tiles * 2 * 128 * 128 * 8 + m * k * 8
    for (sp, gr) in ARMS:
        if i >= len(f) or i >= len(w): break
        rd = 2.0 * f[i][4] * 1024; wr = w[i][4] * 1024          # (gfx950: FETCH_SIZE counts half the bytes of a wide read stream -- profiles/r01/pmc_calibration_*)
        print("| %d | %d | %d | %d | %s | %.2f | %.2f | %.2f | %.2f |" % (m, k, sp, gr, "sp" if "dma_sp" in f[i][1] else "plain", f[i][3] * 1e-6, rd * 1e-9, wr * 1e-9, (rd + wr) / alg))
        i += 1
