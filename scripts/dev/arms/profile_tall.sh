#!/bin/bash
# Counter passes for the 256x128-tile A/B: FETCH_SIZE, WRITE_SIZE and matrix-pipe busy /
# clock for ONE SYRK-shaped launch (M = 32768, K = 1024, lower) with the 128x128 and the 256x128 kernel.
#   gpurun --timeout 900 -- 'bash scripts/profile_tall.sh'      -> gpurun_out/prof_tall/*.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$R/gpurun_out/prof_tall"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 rocprofv3 "$@" > "$OUT/$tag.log" 2>&1; echo "$tag rc=$?"; }
for t in 0 2; do
  run fetch_t$t --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch_t$t" -o pmc -- python $R/scripts/gemm_tall_one.py $t
  run write_t$t --pmc WRITE_SIZE --kernel-trace -d "$OUT/write_t$t" -o pmc -- python $R/scripts/gemm_tall_one.py $t
  run mfma_t$t --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/mfma_t$t" -o pmc -- python $R/scripts/gemm_tall_one.py $t
  run wait_t$t --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace -d "$OUT/wait_t$t" -o pmc -- python $R/scripts/gemm_tall_one.py $t
done
cd "$R"
for d in fetch_t0 write_t0 mfma_t0 wait_t0 fetch_t2 write_t2 mfma_t2 wait_t2; do
  f=$(find "$OUT/$d" -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/summarize_prof.py "$f" "$OUT/$d.md"
  grep "gemm_f64_mfma_dma" "$OUT/$d.md" | head -8
done
find "$OUT" -name "*.db" -delete
