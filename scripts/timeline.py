"""Dump a window of the kernel timeline (per queue) from a rocprofv3 rocpd database."""
import sqlite3, sys
db, lo, cnt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
con = sqlite3.connect(db); cur = con.cursor()
rows = list(cur.execute("select d.start, d.end, d.queue_id, d.stream_id, d.grid_size_x, s.kernel_name from rocpd_kernel_dispatch d "
                        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
t0 = rows[0][0]
for r in rows[lo:lo + cnt]:
    print("%10.1f us  +%8.1f us  q=%s st=%s grid=%7d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3], r[4] // 256, r[5][:40]))
