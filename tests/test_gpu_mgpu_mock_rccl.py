"""The RCCL branch of the sharded solver (ncclSend / ncclRecv / groups on two communicators, george_amd/csrc/gh_mgpu.hip)
on the ONE GPU of the test box: tests/mock_rccl/libmock_rccl.so stands in for librccl.so (GEORGE_AMD_RCCL_LIB), lets ranks
share a device and checks every send against its receive -- count, type, peer, per-pair issue order, nothing left over.
Each case is a child pytest over tests/test_gpu_mgpu.py (W = 1, 2, 3, 4, 6, 8; grids P x 1, 1 x 2, 2 x 2, 1 x 4, 2 x 3,
2 x 4, 4 x 2; factorisation, sweeps, predict, trace, chain-only, errors, a second compute on the same handle) because the
library is chosen once per process.  First the stand-in itself: does it fail on what it must fail on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")


def _env(**kw):
    env = dict(os.environ)
    env.update({"GEORGE_AMD_RCCL_LIB": LIB, "GEORGE_AMD_TEST_VIRTUAL_TRANSPORT": "rccl", "MOCK_RCCL_TIMEOUT_S": "20"})
    env.update(kw)
    return env


def test_stand_in_library_catches_what_it_must():
    assert os.path.exists(LIB), "tests/mock_rccl/libmock_rccl.so is built by __graft_entry__.build()"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "selftest.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    checks = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(checks) >= 15 and all(checks.values()), checks


@pytest.mark.parametrize("one_comm", [0, 1])
def test_grid_matrix_through_the_rccl_branch(one_comm):
    """one_comm = 0: chain and bulk gather on their own communicators (or one, if gh_mgpu_create's dispatch probe says so);
    one_comm = 1: GH_MGPU_ONE_COMM, the fallback, forced"""
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_mgpu.py"), "-x", "-q", "-s", "-m", "gpu",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_env(GEORGE_AMD_TEST_ONE_COMM=str(one_comm)))
    tail = p.stdout[-4000:] + p.stderr[-2000:]
    assert p.returncode == 0, tail
    assert "stand-in RCCL:" in p.stdout and " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1], tail


def test_a_stuck_transfer_is_bounded_by_the_drain_timeout():
    """the end of a sharded factorisation used to issue its two device-to-host copies (into pageable
    stack variables) BEFORE mg_drain -- such a copy blocks inside hipMemcpyAsync until the stream reaches it, so the bounded
    poll was never reached in the hang it exists for.  Here the stand-in holds one receiver's stream for 8 s in front of the
    40th transfer (MOCK_RCCL_STALL); with GEORGE_AMD_MGPU_TIMEOUT_S=1 compute() must come back with the drain error well
    before the stall ends, the solver must be dead, and a new solver in the same process must work."""
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import zoo
from george_amd import kernels, MultiGPUSolver, BasicSolver
x, yerr, y = zoo.bench_data(3000)
kernel = np.var(y) * kernels.Matern32Kernel(1.0)
X = x[:, None]
s = MultiGPUSolver(kernel, devices=[0, 0, 0, 0], transport="rccl", nb=256)
t0 = time.time()
try:
    s.compute(X, yerr)
    print("NOERROR")
except RuntimeError as e:
    print("ERR %%.2f %%s" %% (time.time() - t0, str(e).replace("\n", " ")))
try:
    s.compute(X, yerr)
    print("SECOND-NOERROR")
except RuntimeError as e:
    print("DEAD " + str(e)[:80])
time.sleep(9.0)                                   # (the held stream drains by itself)
s2 = MultiGPUSolver(kernel, devices=[0, 0, 0, 0], transport="rccl", nb=256)
s2.compute(X, yerr)
d = BasicSolver(kernel); d.compute(X, yerr)
print("AGAIN %%.3e" %% (abs(s2.log_determinant - d.log_determinant) / abs(d.log_determinant)))
''' % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=_env(MOCK_RCCL_STALL="40:8", GEORGE_AMD_MGPU_TIMEOUT_S="1"))
    out = p.stdout
    assert p.returncode == 0, out[-2000:] + p.stderr[-3000:]
    err = [l for l in out.splitlines() if l.startswith("ERR ")]
    assert err and "did not drain within 1 s" in err[0], out
    assert float(err[0].split()[1]) < 6.0, err[0]                # gave up inside the 8-s stall: the poll, not a blocking copy
    assert any(l.startswith("DEAD ") for l in out.splitlines()), out
    again = [l for l in out.splitlines() if l.startswith("AGAIN ")]
    assert again and float(again[0].split()[1]) < 1e-11, out
