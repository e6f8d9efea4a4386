"""HODLR compute()+log_likelihood() A/B of two builds of the library (the tree's against george_amd/csrc/libgeorge_amd_c.so), by
swapping the .so between child processes (A B A B); also checks that log-likelihood bits and ranks agree.  Run on a scratch copy."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd.so")
A = LIB + ".A"; B = os.path.join(ROOT, "george_amd", "csrc", "libgeorge_amd_c.so")
CODE = ("import sys; sys.path.insert(0, %r); import bench, hashlib\n"
        "n = int(sys.argv[1])\n"
        "job = bench.HodlrJob(n, 0)\n"
        "best = 1e30\n"
        "for rep in range(4):\n"
        "    el, ll = bench.run_timed(job, 20, 3, lambda: None)\n"
        "    best = min(best, el / 20)\n"
        "print('RESULT', best * 1e3, repr(float(ll)), hashlib.md5(str(job.ranks()).encode()).hexdigest()[:8])\n") % ROOT
def run(n):
    r = subprocess.run([sys.executable, "-c", CODE, str(n)], capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT"):
            _, ms, ll, rk = line.split(); return float(ms), ll + " " + rk
    return float("nan"), r.stderr[-300:]
shutil.copy(LIB, A)
print("| N | tree ms (2 runs) | variant ms (2 runs) | variant/tree (best) | same log-likelihood bits and ranks |"); print("|---|---|---|---|---|")
for n in [int(a) for a in sys.argv[1:]] or [262144, 2097152, 32768]:
    ra, rb = [], []
    for rep in range(2):
        shutil.copy(A, LIB); ra.append(run(n))
        shutil.copy(B, LIB); rb.append(run(n))
    shutil.copy(A, LIB)
    a = min(x[0] for x in ra); b = min(x[0] for x in rb)
    print("| %d | %s | %s | %.4f | %s |" % (n, " ".join("%.3f" % x[0] for x in ra), " ".join("%.3f" % x[0] for x in rb), b / a, ra[0][1] == rb[0][1]), flush=True)
    if ra[0][1] != rb[0][1]: print("  tree:", ra[0][1], " variant:", rb[0][1])
