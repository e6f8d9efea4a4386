#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/kmat; O=/root/repo/gpurun_out/kmat; export TMPDIR=/tmp
python scripts/dev/kmat_interp_ab.py 8192 2>&1 | tail -4
cd /tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace -d $O/$tag -o pmc -- python /root/repo/scripts/dev/kmat_interp_ab.py 8192 > $O/$tag.log 2>&1; echo "$tag rc=$?"
  f=$(find $O/$tag -name "*.db" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
q = ("select d.dispatch_id, p.name, sum(e.value), d.end-d.start from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
     "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
     "where s.kernel_name like '%kmat_kernelILb0%' group by d.dispatch_id, p.name order by d.dispatch_id")
rows = list(cur.execute(q))
last = {}
for did, name, v, dt in rows: last.setdefault(did, {})[name] = v; last[did]["ns"] = dt
ids = sorted(last)
for did in ids[2::3][:4]: print(did, {k: ("%.4g" % v) for k, v in last[did].items()})
PY
done
find $O -name "*.db" -size +6M -delete
