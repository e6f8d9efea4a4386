"""One compute() per size with the dataflow factorisation, timed and compared with the launch chain's log-determinant: the
first thing a GPU session runs (a scheduling bug shows as 2-second time-outs here, not as a silent hang of the whole suite)."""
import sys, os, time, faulthandler, threading
faulthandler.dump_traceback_later(20, repeat=False)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zoo
from george_amd import kernels, BasicSolver, _native as N

bad = 0
for n in [int(a) for a in sys.argv[1:]] or [300, 1100, 2200, 4096]:
    x, yerr, y = zoo.bench_data(n)
    kernel = np.var(y) * kernels.ExpSquaredKernel(1.0)
    out = []
    for mode in (0, 1, 1):
        print("  N=%d mode %d ..." % (n, mode), flush=True)
        N.lib.gh_debug_set_dataflow(mode)
        s = BasicSolver(kernel)
        t0 = time.perf_counter()
        res = []

        def work():
            try:
                s.compute(x[:, None], yerr)
                res.append(s.log_determinant)
            except Exception as e:
                res.append(repr(e)[:200])

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(6.0)
        if th.is_alive():                      # stuck: what do the counters say?
            import ctypes as C
            buf = (C.c_uint32 * 360)()
            rc = N.lib.gh_debug_dflow_peek(s._handle, buf, 360)
            w = list(buf)
            print("STUCK after 6 s: peek rc %d  D %d abort %d key %#x (word 96: %d) tails %s heads %s" %
                  (rc, w[0], w[32], w[64], w[96], w[128:133], w[136:141]), flush=True)
            print("  rowh %s" % (w[320:330],), flush=True)
            os._exit(4)
        out.append((res[0], time.perf_counter() - t0))
    ok = out[0][0] == out[1][0] == out[2][0]
    bad += 0 if ok else 1
    print("N=%d launch chain %r (%.3f s)  dataflow %r (%.3f s) %r (%.3f s)  %s" %
          (n, out[0][0], out[0][1], out[1][0], out[1][1], out[2][0], out[2][1], "same bits" if ok else "DIFFERENT"), flush=True)
    if max(o[1] for o in out[1:]) > 1.5:
        print("a dataflow compute took more than 1.5 s: stopping", flush=True)
        sys.exit(3)
N.lib.gh_debug_set_dataflow(-1)
sys.exit(1 if bad else 0)
