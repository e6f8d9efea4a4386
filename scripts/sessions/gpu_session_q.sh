#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/prof_hod
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_hod/t" -o hod -- python $R/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu > "$R/gpurun_out/prof_hod/hod.log" 2>&1
f=$(find "$R/gpurun_out/prof_hod/t" -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys, collections
con=sqlite3.connect(sys.argv[1]); cur=con.cursor()
rows=list(cur.execute("select d.start,d.end,d.queue_id,d.grid_size_x/d.workgroup_size_x,d.grid_size_y,d.grid_size_z,s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
# last compute = after the last 'leaf_build'
lb=[r for r in rows if 'leaf_build' in r[6]]
t_leaf=lb[-1][0]
# start: the first aca kernel within 3 ms before
aca=[r for r in rows if 'aca' in r[6] and r[0] > t_leaf-3e6 and r[0] < t_leaf+3e6]
t0=min(r[0] for r in aca)
last=[r for r in rows if r[0]>=t0-100000]
print("span %.3f ms" % ((last[-1][1]-t0)/1e6))
for r in last:
    d=(r[1]-r[0])/1e3
    if d>25: print("%8.1f +%7.1f q=%s grid=(%d,%d,%d) %s" % ((r[0]-t0)/1e3,d,r[2],r[3],r[4],r[5],r[6][:30]))
agg=collections.OrderedDict()
for r in last:
    k=r[6][:30]; agg.setdefault(k,[0,0.0]); agg[k][0]+=1; agg[k][1]+=(r[1]-r[0])/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print("  %-32s %4d %8.1f us" % (k,v[0],v[1]))
PY
