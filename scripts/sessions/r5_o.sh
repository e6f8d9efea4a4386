#!/bin/bash
# round 5, session o: is it the polling?  the same build with idle workers polling every ~8 us and every ~60 us
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in 2 16; do
  make -C george_amd/csrc clean > /dev/null 2>&1
  make -C george_amd/csrc -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDF_IDLE_LONG=$L" > gpurun_out/r5_o_build_$L.log 2>&1 || { tail -5 gpurun_out/r5_o_build_$L.log; exit 0; }
  timeout -s KILL 60 python scripts/dev/dataflow_smoke.py 2200 > gpurun_out/r5_o_smoke_$L.log 2>&1 || { tail -5 gpurun_out/r5_o_smoke_$L.log; exit 0; }
  timeout -s KILL 120 python scripts/dev/dataflow_trace.py 8192 > gpurun_out/r5_o_trace_$L.log 2>&1
  echo "== DF_IDLE_LONG=$L"; tail -16 gpurun_out/r5_o_trace_$L.log
done
