#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/syrk; O=/root/repo/gpurun_out/syrk; export TMPDIR=/tmp
REPS=3 timeout -s KILL 300 python scripts/dev/syrk_traffic_ab.py > $O/times.md 2>&1; cat $O/times.md
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o pmc -- python /root/repo/scripts/dev/syrk_traffic_ab.py > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o pmc -- python /root/repo/scripts/dev/syrk_traffic_ab.py > $O/write.log 2>&1; echo "write rc=$?"
cd /root/repo
python scripts/dev/syrk_traffic_table.py "$(find $O/fetch -name '*.db' | head -1)" "$(find $O/write -name '*.db' | head -1)" > $O/traffic.md; cat $O/traffic.md
find $O -name "*.db" -size +6M -delete
