#!/bin/bash
# in-kernel stamps of the link kernel (workgroup 0 and the last workgroup of every launch of one panel), N = 1024
cd /root/repo; export TMPDIR=/tmp
cp george_amd/csrc/libgeorge_amd.so /tmp/lib_default.so
( cd george_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGH_LINK_TIMES -c gh_potf2.hip -o /tmp/gh_potf2_times.o 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -fPIC build/gh_kmat.o build/gh_gemm.o /tmp/gh_potf2_times.o build/gh_chol.o build/gh_hodlr.o build/gh_mgpu.o -ldl -lpthread -o libgeorge_amd.so )
python - <<'PY' 2>&1 | tail -40
import sys, ctypes as C; sys.path.insert(0, "/root/repo")
import numpy as np, bench, torch
from george_amd import _native as N
job = bench.DenseJob(1024, 0, 0, profile=False)
for i in range(5): job.step()
torch.cuda.synchronize()
N.lib.gh_debug_link_stamps.restype = C.c_int
buf = (C.c_longlong * 512)()
assert N.lib.gh_debug_link_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(32, 2, 8)
t0 = st[0, 0, 0]
names = ["start", "Linv in LDS", "X done", "strip written", "X in LDS", "product done", "potf2 done"]
for j in range(7):
    for w in range(2):
        r = st[j, w]
        if r[0] == 0: continue
        print("link %d %s: start %+8.2f us | " % (j, "WG0 " if w == 0 else "last", (r[0] - t0) / 100.0) +
              "  ".join("%s %+6.2f" % (names[k], (r[k] - r[0]) / 100.0) for k in range(1, 7 if w == 0 else 6)))
PY
cp /tmp/lib_default.so george_amd/csrc/libgeorge_amd.so
