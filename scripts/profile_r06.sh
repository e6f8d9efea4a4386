#!/bin/bash
# Round-6 measurement + rocprofv3 runs on a gpurun box; summaries land in gpurun_out/prof_r06/ and the reviewed ones are
# copied into profiles/r06/.   gpurun --timeout 2400 -- 'bash scripts/profile_r06.sh'
# Every step under its own `timeout -s KILL`; --pmc passes never share a run with anything but --kernel-trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$R/gpurun_out/prof_r06"
mkdir -p "$OUT"; rm -f "$OUT/bench_lines_under_rocprof.jsonl"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu --no-extra"
run() { tag=$1; lim=$2; shift; shift; timeout -s KILL $lim rocprofv3 "$@" > "$OUT/$tag.log" 2>&1; echo "$tag rc=$?"; }
# 0. the un-profiled default line (what the driver runs), the driver's flags, panel width 2048 beside it, the size sweep
timeout -s KILL 600 python $R/bench.py --dump-intervals "$OUT/update_intervals_N65536.json" --detail "$OUT/bench_detail_final.json" > "$OUT/bench_line_final.json" 2> "$OUT/bench_line_final.err"; echo "bench rc=$?"
timeout -s KILL 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra --detail "$OUT/bench_detail_driver_form.json" > "$OUT/bench_line_driver_form.json" 2> /dev/null; echo "bench (driver form) rc=$?"
timeout -s KILL 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra --nb 1024 --detail "$OUT/bench_detail_nb1024.json" > "$OUT/bench_line_nb1024.json" 2> /dev/null; echo "bench (nb 1024 throughout) rc=$?"
timeout -s KILL 400 python $R/scripts/size_sweep.py > "$OUT/size_sweep.md" 2>/dev/null; echo "sweep rc=$?"
# 1. kernel traces: headline (N = 65536), configs[1] (N = 16384), C4 (HODLR)
run trace64k 300 --kernel-trace --stats -d "$OUT/trace64k" -o trace -- $BENCH --steps 2 --warmup 1
run trace16k 200 --kernel-trace --stats -d "$OUT/trace16k" -o trace -- $BENCH --n 16384 --steps 3 --warmup 1
run traceC4 200 --kernel-trace --stats -d "$OUT/traceC4" -o trace -- python $R/bench.py --workload hodlr --steps 3 --warmup 1 --no-cpu
# 2. traffic leaving the L2s during the trailing SYRK launches (single stream: every launch alone on the GPU)
run pmc_fetch 300 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
run pmc_write 300 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc -- $BENCH --steps 1 --warmup 0 --no-lookahead
# 3. matrix pipe and clock under the SYRK (N = 32768, single stream)
run pmc_mfma 200 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma" -o pmc -- $BENCH --n 32768 --steps 1 --warmup 0 --no-lookahead
# 3b. the same two counters over whole HODLR steps (config C4)
run c4_fetch 200 --pmc FETCH_SIZE --kernel-trace -d "$OUT/c4_fetch" -o pmc -- python $R/scripts/hodlr_traffic_from_pmc.py --job 262144 3
run c4_write 200 --pmc WRITE_SIZE --kernel-trace -d "$OUT/c4_write" -o pmc -- python $R/scripts/hodlr_traffic_from_pmc.py --job 262144 3
# 5. calibration of FETCH_SIZE / WRITE_SIZE on a copy of known size
CAL="import sys, ctypes; sys.path.insert(0, '$R'); from george_amd import _native as N; v = ctypes.c_double(0); N.check(N.lib.gh_microbench_hbm_copy(ctypes.byref(v))); print(v.value)"
run cal_fetch 120 --pmc FETCH_SIZE --kernel-trace -d "$OUT/cal_fetch" -o pmc -- python -c "$CAL"
run cal_write 120 --pmc WRITE_SIZE --kernel-trace -d "$OUT/cal_write" -o pmc -- python -c "$CAL"
cd "$R"
for d in trace64k trace16k traceC4 pmc_fetch pmc_write pmc_mfma cal_fetch cal_write; do
  f=$(find "$OUT/$d" -name "*.db" | head -1)
  case $d in trace16k|trace64k|pmc_fetch|pmc_write) TMIN=8;; *) TMIN="";; esac
  if [ -n "$f" ]; then python scripts/summarize_prof.py "$f" "$OUT/$d.md" $TMIN; fi
done
python scripts/traffic_from_pmc.py "$(find $OUT/pmc_fetch -name '*.db' | head -1)" "$(find $OUT/pmc_write -name '*.db' | head -1)" 65536 1024 "$OUT/traffic_N65536.json"
python scripts/hodlr_traffic_from_pmc.py "$(find $OUT/c4_fetch -name '*.db' | head -1)" "$(find $OUT/c4_write -name '*.db' | head -1)" 262144 3 "$OUT/traffic_C4_N262144.json"
f=$(find "$OUT/trace16k" -name "*.db" | head -1); [ -n "$f" ] && python scripts/chain_stats.py "$f" > "$OUT/chain_stats_N16384.txt"
f=$(find "$OUT/traceC4" -name "*.db" | head -1); [ -n "$f" ] && python scripts/hodlr_levels.py "$f" > "$OUT/hodlr_levels_C4.txt"
f=$(find "$OUT/traceC4" -name "*.db" | head -1); [ -n "$f" ] && python scripts/dev/hodlr_timeline.py "$f" > "$OUT/hodlr_timeline_C4.txt"
for f in trace64k trace16k traceC4; do grep -h '^{' "$OUT/$f.log" | tail -1 >> "$OUT/bench_lines_under_rocprof.jsonl"; done
cut -c1-200 "$OUT/bench_lines_under_rocprof.jsonl"
cat "$OUT/traffic_N65536.json"; grep "dma_sp" "$OUT/pmc_mfma.md" | head -5
cat "$OUT/traffic_C4_N262144.json" | head -12; head -30 "$OUT/chain_stats_N16384.txt"
for f in bench_line_final bench_line_driver_form bench_line_nb1024; do tail -1 "$OUT/$f.json" | cut -c1-240; done
find "$OUT" -name "*.db" -size +6M -delete
ls "$OUT"
