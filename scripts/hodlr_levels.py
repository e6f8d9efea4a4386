"""Per-launch view of the last HODLR compute()+log_likelihood() in a rocprofv3 rocpd database:
the ACA launches (fused wide launch + one-workgroup levels) with the hardware queue each ran on,
then every kernel of that compute aggregated by name, and the idle gaps on the timeline."""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = list(cur.execute(
    "select d.start,d.end,d.grid_size_x/d.workgroup_size_x,s.kernel_name,d.queue_id from rocpd_kernel_dispatch d "
    "join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
aca = [r for r in rows if 'hodlr_aca' in r[3]]
if not aca:
    sys.exit("no ACA launches in this trace")
# ACA launches of one compute() start within a few ms of each other; computes are further apart
groups = [[aca[0]]]
for r in aca[1:]:
    if r[0] - groups[-1][-1][0] < 2.5e6:
        groups[-1].append(r)
    else:
        groups.append([r])
g = groups[-1]
t0 = min(r[0] for r in g)
nxt = [r[0] for r in rows if r[0] > g[-1][1] + 30e6]
t1 = nxt[0] if nxt else rows[-1][1] + 1
last = [r for r in rows if t0 <= r[0] < t1]
print("ACA launches of the last compute (%d):" % len(g))
for r in g:
    print("  queue %3d  grid %5d  start %8.1f us  %8.1f us" % (r[4], r[2], (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3))
print("kernels %d span %.3f ms sum %.3f ms" % (len(last), (max(r[1] for r in last) - t0) / 1e6,
                                               sum(r[1] - r[0] for r in last) / 1e6))
agg = collections.OrderedDict()
for r in last:
    k = r[3][:34]; agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += (r[1] - r[0]) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s %4d  %9.1f us" % (k, v[0], v[1]))
# idle time: union of busy intervals against the span
iv = sorted((r[0], r[1]) for r in last); busy = 0; cs, ce = iv[0]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce) / 1e3); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("busy (union) %.3f ms; idle gaps: sum %.1f us, >20us: %s" % (busy / 1e6, sum(gaps), [round(x) for x in gaps if x > 20]))
