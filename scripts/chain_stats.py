"""Chain-link statistics from a rocprofv3 rocpd database: for consecutive potf2 dispatches, the
link time (start to start), the potf2 duration and what lies in between on the same queue."""
import sqlite3, sys
import numpy as np
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
rows = list(cur.execute("select d.start, d.end, d.queue_id, d.grid_size_x / d.workgroup_size_x, s.kernel_name from rocpd_kernel_dispatch d "
                        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
pot = [r for r in rows if "potf2" in r[4]]
q = pot[0][2]
same = [r for r in rows if r[2] == q]
links, durs = [], []
for a, b in zip(pot[:-1], pot[1:]):
    lk = (b[0] - a[0]) / 1e3
    if lk < 400:                     # inside one panel
        links.append(lk); durs.append((a[1] - a[0]) / 1e3)
links, durs = np.array(links), np.array(durs)
print("potf2 dispatches %d; in-panel links %d: link mean %.1f us (median %.1f), potf2 mean %.1f us (median %.1f, min %.1f), rest %.1f us"
      % (len(pot), len(links), links.mean(), np.median(links), durs.mean(), np.median(durs), durs.min(), (links - durs).mean()))
# detail of ~3 links in the middle
i0 = same.index(pot[len(pot) // 2])
t0 = same[i0][0]
for r in same[i0:i0 + 14]:
    print("%9.1f us  +%7.1f us  grid=%5d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4][:44]))
