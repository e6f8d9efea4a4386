"""gh_debug_stream_dispatch in a torch-free process: is a small kernel on stream j held up by a huge grid on stream i?
usage: dispatch_probe.py   (env GEORGE_AMD_NO_NULL_PRIME / GEORGE_AMD_PRIME_EXTRA select the placement state)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from george_amd import GP, kernels, _native as N
n = 8192
rng = np.random.RandomState(1234)
x = np.sort(rng.uniform(0, 10, n)); yerr = 0.1 * np.ones(n); y = np.sin(x)
gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0))
for i in range(3):
    gp.compute(x, yerr); gp.log_likelihood(y)
t0 = time.perf_counter()
for i in range(5):
    gp.compute(x, yerr); gp.log_likelihood(y)
ms = (time.perf_counter() - t0) / 5 * 1e3
out = (C.c_double * 36)()
N.check(N.lib.gh_debug_stream_dispatch(gp.solver._handle, out, 36))
names = ["null", "main", "chain", "rows", "near", "masked"]
print("state prime=%s extra=%s: N=8192 step %.2f ms" % (os.environ.get("GEORGE_AMD_NO_NULL_PRIME", "yes").replace("1", "no"), os.environ.get("GEORGE_AMD_PRIME_EXTRA", "0"), ms))
for i in (1, 2, 3, 4, 5):
    print("   big grid on %-6s -> small kernel done after (ms): " % names[i] + "  ".join("%s %.2f" % (names[j], out[i * 6 + j]) for j in range(6) if j != i))
