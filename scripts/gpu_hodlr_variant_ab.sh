#!/bin/bash
# A/B of a compile-time variant of gh_hodlr.hip on one box: VARIANT="-DGH_ACA_BATCH_ONES=0" bash scripts/gpu_hodlr_variant_ab.sh
cd /root/repo; export TMPDIR=/tmp
cp george_amd/csrc/libgeorge_amd.so /tmp/lib_default.so
( cd george_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $VARIANT -c gh_hodlr.hip -o /tmp/gh_hodlr_var.o 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -fPIC build/gh_kmat.o build/gh_gemm.o build/gh_potf2.o build/gh_chol.o /tmp/gh_hodlr_var.o build/gh_mgpu.o -ldl -lpthread -o /tmp/lib_variant.so )
cat > /tmp/t.py <<'PY'
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, bench, torch
for n in [int(a) for a in sys.argv[2:]]:
    job = bench.HodlrJob(n, 0)
    for i in range(4): job.step()
    ts = []
    for i in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ll = job.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s N = %7d: %.3f / %.3f / %.3f ms  ll %.15g" % (sys.argv[1], n, min(ts), float(np.median(ts)), max(ts), ll), flush=True)
    del job
PY
for rnd in 1 2 3; do
  cp /tmp/lib_default.so george_amd/csrc/libgeorge_amd.so; python /tmp/t.py default ${SIZES:-262144} 2>&1 | grep "N ="
  cp /tmp/lib_variant.so george_amd/csrc/libgeorge_amd.so; python /tmp/t.py variant ${SIZES:-262144} 2>&1 | grep "N ="
done
cp /tmp/lib_default.so george_amd/csrc/libgeorge_amd.so
