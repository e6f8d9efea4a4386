#!/bin/bash
# rocprofv3 runs of the headline bench on a gpurun box; summaries land in gpurun_out/ and are
# copied into profiles/ by hand after review.
#   gpurun --timeout 1500 -- 'bash scripts/profile.sh r01'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the headline config (N = 65536)
rocprofv3 --kernel-trace --stats -d "$OUT/trace64k" -o trace -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-extra > "$OUT/trace64k.log" 2>&1
# 2. kernel trace of configs[1] (N = 16384)
rocprofv3 --kernel-trace --stats -d "$OUT/trace16k" -o trace -- python "$R/bench.py" --n 16384 --steps 3 --warmup 1 --no-cpu --no-extra > "$OUT/trace16k.log" 2>&1
# 3. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), N = 32768
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc -- python "$R/bench.py" --n 32768 --steps 1 --warmup 0 --no-cpu --no-extra > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc -- python "$R/bench.py" --n 32768 --steps 1 --warmup 0 --no-cpu --no-extra > "$OUT/pmc_write.log" 2>&1
# 4. calibration of the same counters on a kernel of known traffic (16-B/lane copy, 2 GiB in + 2 GiB out per launch)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/cal_fetch" -o pmc -- python -c "
import sys, ctypes; sys.path.insert(0, '$R')
from george_amd import _native as N
v = ctypes.c_double(0); N.check(N.lib.gh_microbench_hbm_copy(ctypes.byref(v))); print(v.value)" > "$OUT/cal_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/cal_write" -o pmc -- python -c "
import sys, ctypes; sys.path.insert(0, '$R')
from george_amd import _native as N
v = ctypes.c_double(0); N.check(N.lib.gh_microbench_hbm_copy(ctypes.byref(v))); print(v.value)" > "$OUT/cal_write.log" 2>&1
# 5. MFMA utilisation counters on the SYRK kernel
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma" -o pmc -- python "$R/bench.py" --n 32768 --steps 1 --warmup 0 --no-cpu --no-extra > "$OUT/pmc_mfma.log" 2>&1
cd "$R"
for d in trace64k trace16k pmc_fetch pmc_write cal_fetch cal_write pmc_mfma; do
  f=$(find "$OUT/$d" -name "*.db" | head -1)
  if [ -n "$f" ]; then python scripts/summarize_prof.py "$f" "$OUT/$d.md"; fi
done
tail -2 "$OUT"/*.log | cut -c1-600
ls -la "$OUT"
# keep the merge-back under the 64 MiB cap: drop the raw databases
find "$OUT" -name "*.db" -size +20M -delete
