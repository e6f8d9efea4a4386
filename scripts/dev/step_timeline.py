"""Time-ordered kernels of the LAST compute()+log_likelihood() step in a rocprofv3 rocpd database (dense solver):
start relative to the step's kernel-matrix build, duration, gap to the previous kernel's end, queue, grid, name."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute(
    "select d.start,d.end,d.grid_size_x/d.workgroup_size_x,s.kernel_name,d.queue_id from rocpd_kernel_dispatch d "
    "join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
km = [i for i, r in enumerate(rows) if 'kmat' in r[3]]
i0 = km[-1]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 30e3 and 'kmat' not in rows[i0 - 1][3] and 'reduce_final' not in rows[i0 - 1][3]: i0 -= 1
last = rows[i0:]
t0 = last[0][0]; tail = 0
for r in last:
    print("%8.1f us  +%6.1f us  gap %6.1f  q%d  grid %5d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, (r[0] - tail) / 1e3 if tail else 0.0, r[4], r[2], r[3][:70]))
    tail = max(tail, r[1])
print("span %.1f us, kernel time %.1f us, launches %d" % ((tail - t0) / 1e3, sum(r[1] - r[0] for r in last) / 1e3, len(last)))
