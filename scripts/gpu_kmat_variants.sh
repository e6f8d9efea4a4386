#!/bin/bash
# build variants of gh_kmat.hip ON THE BOX and run a command with each: VARIANTS="name:flags;..." CMD="python ..."
cd /root/repo; export TMPDIR=/tmp
cp george_amd/csrc/libgeorge_amd.so /tmp/lib_default.so
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  name="${v%%:*}"; flags="${v#*:}"
  ( cd george_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c gh_kmat.hip -o /tmp/gh_kmat_$name.o 2>/dev/null && \
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gh_kmat_$name.o build/gh_gemm.o build/gh_potf2.o build/gh_chol.o build/gh_hodlr.o build/gh_mgpu.o -ldl -lpthread -o libgeorge_amd.so )
  echo "== variant $name ($flags)"
  eval "$CMD"
done
cp /tmp/lib_default.so george_amd/csrc/libgeorge_amd.so
