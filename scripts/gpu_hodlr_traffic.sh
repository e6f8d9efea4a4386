#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/hodlr; O=/root/repo/gpurun_out/hodlr; export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/c4_fetch -o pmc -- python /root/repo/scripts/hodlr_traffic_from_pmc.py --job 262144 3 > $O/c4_fetch.log 2>&1; echo "fetch rc=$?"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/c4_write -o pmc -- python /root/repo/scripts/hodlr_traffic_from_pmc.py --job 262144 3 > $O/c4_write.log 2>&1; echo "write rc=$?"
cd /root/repo
python scripts/hodlr_traffic_from_pmc.py "$(find $O/c4_fetch -name '*.db' | head -1)" "$(find $O/c4_write -name '*.db' | head -1)" 262144 3 "$O/traffic_C4_N262144.json" | head -60
find $O -name "*.db" -size +6M -delete
