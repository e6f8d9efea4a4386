#!/bin/bash
# Round-trip script for a gpurun box: unit tests -> parity tests -> bench -> rocprof summary.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [quick|full]'
MODE=${1:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print('cuda', torch.cuda.is_available(), torch.cuda.get_device_name(0))" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -s -p no:cacheprovider > gpurun_out/t_gemm.log 2>&1; echo "gemm rc=$?"
tail -25 gpurun_out/t_gemm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider > gpurun_out/t_kernels.log 2>&1; echo "kernels rc=$?"
tail -30 gpurun_out/t_kernels.log
timeout 1200 python -m pytest tests/test_gpu_solver.py -q -p no:cacheprovider > gpurun_out/t_solver.log 2>&1; echo "solver rc=$?"
tail -40 gpurun_out/t_solver.log
if [ -f tests/test_gpu_hodlr.py ]; then
  timeout 900 python -m pytest tests/test_gpu_hodlr.py -q -p no:cacheprovider > gpurun_out/t_hodlr.log 2>&1; echo "hodlr rc=$?"
  tail -30 gpurun_out/t_hodlr.log
fi
timeout 600 python bench.py --n 16384 --steps 2 --warmup 1 --no-extra --cpu-n 4096 > gpurun_out/bench16k.log 2>&1; echo "bench16k rc=$?"
tail -3 gpurun_out/bench16k.log
if [ "$MODE" = "full" ]; then
  timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu --no-extra > gpurun_out/bench64k.log 2>&1; echo "bench64k rc=$?"
  tail -3 gpurun_out/bench64k.log
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof16k" -o prof16k -- python "$R/bench.py" --n 16384 --steps 2 --warmup 1 --no-cpu --no-extra > "$R/gpurun_out/rocprof16k.log" 2>&1; echo "rocprof rc=$?"
  cd "$R"
  find gpurun_out/prof16k -name "*stats*" | head
  for f in $(find gpurun_out/prof16k -name "*kernel_stats*.csv" | head -1); do head -25 "$f"; done
fi
