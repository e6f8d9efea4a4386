#!/bin/bash
# rocprofv3 runs of the headline bench on a gpurun box; summaries land in gpurun_out/prof_<tag>/
# and the reviewed ones are copied into profiles/<tag>/.
#   gpurun --timeout 1500 -- 'bash scripts/profile.sh r01b'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"; rm -f "$OUT/bench_lines_under_rocprof.jsonl"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-extra"
# 1. kernel traces of the headline config (N = 65536) with and without look-ahead, and of configs[1]
rocprofv3 --kernel-trace --stats -d "$OUT/trace64k" -o trace -- $BENCH --steps 2 --warmup 1 > "$OUT/trace64k.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/trace64k_nola" -o trace -- $BENCH --steps 2 --warmup 1 --no-lookahead > "$OUT/trace64k_nola.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/trace16k" -o trace -- $BENCH --n 16384 --steps 3 --warmup 1 > "$OUT/trace16k.log" 2>&1
# 2. L2-egress traffic counters for the headline config, separate passes (single stream so that
#    every trailing launch is measured on an otherwise idle GPU)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead > "$OUT/pmc_write.log" 2>&1
# 3. calibration of the same counters on a kernel of known traffic (16-B/lane copy, 2 GiB in + 2 GiB out per launch)
CAL="import sys, ctypes; sys.path.insert(0, '$R'); from george_amd import _native as N; v = ctypes.c_double(0); N.check(N.lib.gh_microbench_hbm_copy(ctypes.byref(v))); print(v.value)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/cal_fetch" -o pmc -- python -c "$CAL" > "$OUT/cal_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/cal_write" -o pmc -- python -c "$CAL" > "$OUT/cal_write.log" 2>&1
# 4. matrix-pipe busy counters
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma" -o pmc -- $BENCH --n 32768 --steps 1 --warmup 0 --no-lookahead > "$OUT/pmc_mfma.log" 2>&1
cd "$R"
for d in trace64k trace64k_nola trace16k pmc_fetch pmc_write cal_fetch cal_write pmc_mfma; do
  f=$(find "$OUT/$d" -name "*.db" | head -1)
  case $d in trace16k) TMIN=8;; trace64k*|pmc_fetch|pmc_write) TMIN=8;; *) TMIN="";; esac
  if [ -n "$f" ]; then python scripts/summarize_prof.py "$f" "$OUT/$d.md" $TMIN; fi
done
python scripts/traffic_from_pmc.py "$(find $OUT/pmc_fetch -name '*.db' | head -1)" "$(find $OUT/pmc_write -name '*.db' | head -1)" 65536 1024 "$OUT/traffic_N65536.json"
for f in trace64k trace64k_nola trace16k; do grep '^{"metric' "$OUT/$f.log" | tail -1 >> "$OUT/bench_lines_under_rocprof.jsonl"; done
cut -c1-300 "$OUT/bench_lines_under_rocprof.jsonl"
ls "$OUT"
find "$OUT" -name "*.db" -size +8M -delete
