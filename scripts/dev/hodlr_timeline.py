"""Time-ordered list of every kernel of the last HODLR compute()+log_likelihood() in a rocprofv3 rocpd database
(start relative to the first ACA launch, duration, gap to the previous end on the same queue, queue, grid, name)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute(
    "select d.start,d.end,d.grid_size_x/d.workgroup_size_x,s.kernel_name,d.queue_id from rocpd_kernel_dispatch d "
    "join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
aca = [r for r in rows if 'hodlr_aca' in r[3]]
groups = [[aca[0]]]
for r in aca[1:]:
    if r[0] - groups[-1][-1][0] < 2.5e6: groups[-1].append(r)
    else: groups.append([r])
g = groups[-1]
t0 = min(r[0] for r in g)
last = [r for r in rows if r[0] >= t0]
endq = {}
tail = 0
for r in last:
    gap = (r[0] - endq[r[4]]) / 1e3 if r[4] in endq else float('nan')
    gl = (r[0] - tail) / 1e3 if tail else float('nan')
    print("%9.1f us  +%7.1f us  (queue gap %6.1f, global gap %6.1f)  q%d  grid %6d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap, gl, r[4], r[2], r[3][:60]))
    endq[r[4]] = r[1]
    tail = max(tail, r[1])
