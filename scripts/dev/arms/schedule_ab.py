"""compute()+log_likelihood() at the chain-bound sizes under the two factorisation schedules (GEORGE_AMD_SCHEDULE=panels: depth-1
panel look-ahead; =columns: the column-priority schedule of round 4) and two panel widths, each arm in a process of its own
(the switch is read once).  python scripts/schedule_ab.py [N ...]  ->  markdown table."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys; sys.path.insert(0, %r); import bench\n"
        "n, nb = int(sys.argv[1]), int(sys.argv[2])\n"
        "job = bench.DenseJob(n, nb, 0, profile=False)\n"
        "steps = 20 if n <= 8192 else 10\n"
        "best = 1e30\n"
        "for rep in range(3):\n"
        "    el, ll = bench.run_timed(job, steps, 3, lambda: None)\n"
        "    best = min(best, el / steps)\n"
        "print('RESULT', best * 1e3, repr(float(ll)))\n") % ROOT


def run(n, nb, sched):
    e = dict(os.environ)
    e["GEORGE_AMD_SCHEDULE"] = sched
    r = subprocess.run([sys.executable, "-c", CODE, str(n), str(nb)], env=e, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("RESULT"):
            _, ms, ll = line.split()
            return float(ms), ll
    return float("nan"), r.stderr[-200:]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [3072, 4096, 6144, 8192, 12288, 16384, 20480]
    print("| N | panels nb=1024 (ms) | columns nb=1024 | columns nb=512 | panels nb=512 | best vs panels/1024 | same log-likelihood bits |")
    print("|---|---|---|---|---|---|---|")
    for n in sizes:
        a, la = run(n, 1024, "panels")
        b, lb = run(n, 1024, "columns")
        c, lc = run(n, 512, "columns")
        d, ld = run(n, 512, "panels")
        best = min(b, c, d)
        print("| %d | %.3f | %.3f | %.3f | %.3f | %.1f %% | %s |" % (n, a, b, c, d, 100.0 * (best / a - 1.0), la == lb), flush=True)


if __name__ == "__main__":
    main()
