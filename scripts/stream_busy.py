"""Per-queue busy / idle view of ONE compute() from a rocprofv3 rocpd database (the last one in the file):
python scripts/stream_busy.py <db>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute("select d.start,d.end,d.queue_id,d.grid_size_x/d.workgroup_size_x,s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
km = [r for r in rows if 'kmat' in r[4]]
t0 = km[-1][0]
last = [r for r in rows if r[0] >= t0]
tend = max(r[1] for r in last)
print("compute span %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(last)))
qs = sorted(set(r[2] for r in last))
for q in qs:
    rs = [r for r in last if r[2] == q]
    busy = sum(r[1] - r[0] for r in rs) / 1e6
    first, lastt = (rs[0][0] - t0) / 1e6, (rs[-1][1] - t0) / 1e6
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(rs[:-1], rs[1:])]
    big = [g for g in gaps if g > 30]
    names = {}
    for r in rs:
        k = r[4][:26]; names[k] = names.get(k, 0) + (r[1] - r[0]) / 1e6
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print("queue %s: %4d kernels, busy %.2f ms, active window %.2f..%.2f ms, gaps>30us: %d totalling %.2f ms; %s"
          % (q, len(rs), busy, first, lastt, len(big), sum(big) / 1e3, ", ".join("%s %.2f" % kv for kv in top)))
# potf2 chain: per panel (8 potf2 each) the time from first potf2 start to last potf2 end
pot = [r for r in last if 'potf2' in r[4]]
per = []
for i in range(0, len(pot), 8):
    grp = pot[i:i + 8]
    per.append(((grp[0][0] - t0) / 1e6, (grp[-1][1] - grp[0][0]) / 1e6))
print("panels (start ms, chain ms):", " ".join("%.2f/%.2f" % p for p in per))
