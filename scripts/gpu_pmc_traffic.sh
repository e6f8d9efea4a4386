#!/bin/bash
R=/root/repo; OUT="$R/gpurun_out/prof_r06"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-extra"
run() { tag=$1; lim=$2; shift; shift; timeout -s KILL $lim rocprofv3 "$@" > "$OUT/$tag.log" 2>&1; echo "$tag rc=$?"; }
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
run pmc_fetch 300 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
run pmc_write 300 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
cd $R
for d in pmc_fetch pmc_write; do f=$(find "$OUT/$d" -name "*.db" | head -1); [ -n "$f" ] && python scripts/summarize_prof.py "$f" "$OUT/$d.md" 8; done
python scripts/traffic_from_pmc.py "$(find $OUT/pmc_fetch -name '*.db' | head -1)" "$(find $OUT/pmc_write -name '*.db' | head -1)" 65536 1024 "$OUT/traffic_N65536.json"
find "$OUT" -name "*.db" -size +6M -delete
