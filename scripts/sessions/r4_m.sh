#!/bin/bash
# round 4: L2-miss traffic of the trailing SYRK launches after the grouped tile order (two --pmc passes, nothing else)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4m; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-extra"
run() { tag=$1; shift; timeout 600 rocprofv3 "$@" > "$OUT/$tag.log" 2>&1; echo "$tag rc=$?"; }
run pmc_fetch --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
run pmc_write --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
cd $R
python scripts/traffic_from_pmc.py "$(find $OUT/pmc_fetch -name '*.db' | head -1)" "$(find $OUT/pmc_write -name '*.db' | head -1)" 65536 1024 "$OUT/traffic_N65536.json"
for d in pmc_fetch pmc_write; do f=$(find "$OUT/$d" -name "*.db" | head -1); [ -n "$f" ] && python scripts/summarize_prof.py "$f" "$OUT/$d.md" 8; done
cat $OUT/traffic_N65536.json
find "$OUT" -name "*.db" -delete
