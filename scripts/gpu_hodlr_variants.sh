#!/bin/bash
# build variants of gh_hodlr.hip ON THE BOX and time C4 with each (compile-time knobs): VARIANTS="name:flags;name:flags"
cd /root/repo; mkdir -p gpurun_out/hodlr; export TMPDIR=/tmp
O=gpurun_out/hodlr
cp george_amd/csrc/libgeorge_amd.so /tmp/lib_default.so
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  name="${v%%:*}"; flags="${v#*:}"
  ( cd george_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c gh_hodlr.hip -o /tmp/gh_hodlr_$name.o 2>/dev/null && \
    hipcc --offload-arch=gfx950 -shared -fPIC build/gh_kmat.o build/gh_gemm.o build/gh_potf2.o build/gh_chol.o /tmp/gh_hodlr_$name.o build/gh_mgpu.o -ldl -lpthread -o libgeorge_amd.so )
  echo "== variant $name ($flags)"
  timeout -s KILL 200 python scripts/dev/hodlr_wave_ab.py 262144 2>&1 | grep "^| 262144"
done
cp /tmp/lib_default.so george_amd/csrc/libgeorge_amd.so
