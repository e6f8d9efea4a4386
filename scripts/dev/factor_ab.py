"""Whole compute()+log_likelihood() A/B of two builds of the library -- the tree's libgeorge_amd.so against a variant built as
george_amd/csrc/libgeorge_amd_c.so -- by swapping the .so between child processes (A B A B).  Run on a scratch copy (gpurun)."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd.so")
A = LIB + ".A"; B = os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd_c.so")
CODE = ("import sys; sys.path.insert(0, %r); import bench\n"
        "n = int(sys.argv[1])\n"
        "job = bench.DenseJob(n, 1024, 0, profile=False)\n"
        "steps = 20 if n <= 16384 else (8 if n <= 32768 else 4)\n"
        "best = 1e30\n"
        "for rep in range(3):\n"
        "    el, ll = bench.run_timed(job, steps, 2, lambda: None)\n"
        "    best = min(best, el / steps)\n"
        "print('RESULT', best * 1e3, repr(float(ll)))\n") % ROOT
def run(n):
    r = subprocess.run([sys.executable, "-c", CODE, str(n)], capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT"):
            _, ms, ll = line.split(); return float(ms), ll
    return float("nan"), r.stderr[-300:]
shutil.copy(LIB, A)
print("| N | tree ms (2 runs) | variant ms (2 runs) | variant/tree (best) | same bits |"); print("|---|---|---|---|---|")
for n in [int(a) for a in sys.argv[1:]] or [65536, 32768, 16384, 8192]:
    ra, rb = [], []
    for rep in range(2):
        shutil.copy(A, LIB); ra.append(run(n))
        shutil.copy(B, LIB); rb.append(run(n))
    shutil.copy(A, LIB)
    a = min(x[0] for x in ra); b = min(x[0] for x in rb)
    print("| %d | %s | %s | %.4f | %s |" % (n, " ".join("%.3f" % x[0] for x in ra), " ".join("%.3f" % x[0] for x in rb), b / a, ra[0][1] == rb[0][1]), flush=True)
