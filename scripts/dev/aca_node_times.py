"""Per-node durations of the one-workgroup ACA levels: needs libgeorge_amd_c.so built from gh_hodlr.hip with -DGH_ACA_TIMES (the third
compute() of a handle prints one line per level to stderr).  Swaps that build in, runs N (default 262144).  Run on a scratch copy
(gpurun): it overwrites libgeorge_amd.so."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd.so")
shutil.copy(os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd_c.so"), LIB)
CODE = ("import sys; sys.path.insert(0, %r); import bench\n"
        "job = bench.HodlrJob(int(sys.argv[1]), 0)\n"
        "el, ll = bench.run_timed(job, 6, 0, lambda: None)\n"
        "print('ms', el / 6 * 1e3, ll)\n") % ROOT
for n in [int(a) for a in sys.argv[1:]] or [262144]:
    r = subprocess.run([sys.executable, "-c", CODE, str(n)], capture_output=True, text=True, timeout=600)
    print(r.stdout[-200:]); print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[aca]")))
