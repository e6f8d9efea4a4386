#!/bin/bash
# round 4, session i: GEMM kernel changes -- gemm / solver / fullsize parity, then the headline and the size sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_solver.py tests/test_gpu_fullsize.py -x -q --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py --no-cpu --no-extra --steps 3 --warmup 1 --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r4i/bench_line.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['frac_of_fp64_mfma_peak'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('peak_measured'), d['roofline'].get('frac_of_measured'), d['roofline']['with_overlapped_block_column_launches'], d['parity'])
P
timeout 900 python scripts/size_sweep.py > $O/size_sweep.md 2>/dev/null; cat $O/size_sweep.md
