"""Per-launch view of one HODLR compute()+log_likelihood() from a rocprofv3 rocpd database."""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute("select d.start,d.end,d.grid_size_x/d.workgroup_size_x,s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
aca = [r for r in rows if 'aca' in r[3]]
nlev = 11
print("ACA per level (last compute):")
for r in aca[-nlev:]:
    print("  grid %5d  %8.1f us" % (r[2], (r[1] - r[0]) / 1e3))
t0 = aca[-nlev][0]
last = [r for r in rows if r[0] >= t0]
print("kernels %d span %.3f ms sum %.3f ms" % (len(last), (last[-1][1] - t0) / 1e6, sum(r[1] - r[0] for r in last) / 1e6))
agg = collections.OrderedDict()
for r in last:
    k = r[3][:34]; agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += (r[1] - r[0]) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s %4d  %9.1f us" % (k, v[0], v[1]))
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(last[:-1], last[1:])]
print("gaps: sum %.1f us, >20us: %s" % (sum(g for g in gaps if g > 0), [round(g) for g in gaps if g > 20]))
