"""Does the stand-in RCCL library (mock_rccl.cpp) catch what it is there to catch?  Driven directly through ctypes on
one GPU with two and three virtual ranks, each from its own host thread as the sharded solver drives them:
a correct exchange moves the right bytes in issue order; a count mismatch, a type mismatch, a send nobody receives,
a send answered on the wrong communicator set, a self-send, a bad peer and an unbalanced group all fail loudly.
Prints one JSON line; exit status 0 only when every check held.  (tests/test_gpu_mgpu_mock_rccl.py runs it.)"""
import ctypes as C
import json
import os
import sys
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DOUBLE, FLOAT = 8, 7


def main():
    os.environ["MOCK_RCCL_TIMEOUT_S"] = "1.5"
    lib = C.CDLL(os.path.join(HERE, "libmock_rccl.so"))
    vp = C.c_void_p
    lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclGetErrorString.restype = C.c_char_p

    def comm_set(n):
        cs = (vp * n)()
        devs = (C.c_int * n)(*([0] * n))
        assert lib.ncclCommInitAll(cs, n, devs) == 0
        return [vp(c) for c in cs]

    def run(*fns):
        """each function on its own thread (a rank); returns their results"""
        out = [None] * len(fns)

        def w(i):
            torch.cuda.set_device(0)
            out[i] = fns[i]()
        ts = [threading.Thread(target=w, args=(i,)) for i in range(len(fns))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        return out

    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    sp = [vp(s.cuda_stream) for s in streams]
    checks = {}
    A = comm_set(2)
    B = comm_set(2)
    x0 = torch.arange(1000, dtype=torch.float64, device=dev)
    x1 = -torch.arange(1000, dtype=torch.float64, device=dev)
    r0 = torch.zeros(1000, dtype=torch.float64, device=dev)
    r1 = torch.zeros(1000, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    # 1. two sends 0 -> 1, two receives: issue order decides which lands where
    rc = run(lambda: (lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]), lib.ncclSend(x1.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0])),
             lambda: (lib.ncclRecv(r0.data_ptr(), 1000, DOUBLE, 0, A[1], sp[1]), lib.ncclRecv(r1.data_ptr(), 1000, DOUBLE, 0, A[1], sp[1])))
    torch.cuda.synchronize()
    checks["in_order"] = rc == [(0, 0), (0, 0)] and bool((r0 == x0).all()) and bool((r1 == x1).all())

    # 2. an exchange: both ranks send and receive inside one group (ungrouped, each would wait for the other's receive)
    def swap(me, mine, theirs):
        def f():
            lib.ncclGroupStart()
            a = lib.ncclSend(mine.data_ptr(), 1000, DOUBLE, 1 - me, A[me], sp[me])
            b = lib.ncclRecv(theirs.data_ptr(), 1000, DOUBLE, 1 - me, A[me], sp[me])
            return a, b, lib.ncclGroupEnd()
        return f
    r0.zero_(); r1.zero_(); torch.cuda.synchronize()
    rc = run(swap(0, x0, r0), swap(1, x1, r1))
    torch.cuda.synchronize()
    checks["grouped_exchange"] = rc == [(0, 0, 0), (0, 0, 0)] and bool((r0 == x1).all()) and bool((r1 == x0).all())
    # 3. stream order on the sender: the buffer is overwritten right after the send returns; the receiver must see the old values
    big = torch.ones(1 << 22, dtype=torch.float64, device=dev)
    got = torch.zeros(1 << 22, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def send_then_clobber():
        with torch.cuda.stream(streams[0]):
            big.mul_(3.0)                                                   # queued BEFORE the send: must be seen
            rc_ = lib.ncclSend(big.data_ptr(), big.numel(), DOUBLE, 1, A[0], sp[0])
            big.zero_()                                                     # queued AFTER: must not be seen
        return rc_
    rc = run(send_then_clobber, lambda: lib.ncclRecv(got.data_ptr(), got.numel(), DOUBLE, 0, A[1], sp[1]))
    torch.cuda.synchronize()
    checks["stream_order"] = rc == [0, 0] and bool((got == 3.0).all())
    # 4. all-reduce over three ranks
    T = comm_set(3)
    vals = [torch.full((4,), float(i + 1), dtype=torch.float64, device=dev) for i in range(3)]
    outs = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    rc = run(*[(lambda i=i: lib.ncclAllReduce(vals[i].data_ptr(), outs[i].data_ptr(), 4, DOUBLE, 0, T[i], sp[i])) for i in range(3)])
    torch.cuda.synchronize()
    checks["all_reduce"] = rc == [0, 0, 0] and all(bool((o == 6.0).all()) for o in outs)
    st = (C.c_longlong * 8)()
    lib.mock_rccl_stats(st)
    checks["clean_so_far"] = st[4] == 0 and st[5] == 0 and st[1] == 5
    # ---- what must FAIL
    rc = run(lambda: lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]), lambda: lib.ncclRecv(r0.data_ptr(), 999, DOUBLE, 0, A[1], sp[1]))
    checks["count_mismatch_fails_both"] = rc[0] != 0 and rc[1] != 0
    rc = run(lambda: lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]), lambda: lib.ncclRecv(r0.data_ptr(), 1000, FLOAT, 0, A[1], sp[1]))
    checks["type_mismatch_fails_both"] = rc[0] != 0 and rc[1] != 0
    rc = run(lambda: lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]))
    checks["unanswered_send_fails"] = rc[0] != 0
    rc = run(lambda: (lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]), lib.ncclGetErrorString(3)),
             lambda: lib.ncclRecv(r0.data_ptr(), 1000, DOUBLE, 0, B[1], sp[1]))
    checks["other_communicator_never_matches"] = rc[0][0] != 0 and rc[1] != 0 and b"never answered" in rc[0][1]
    # two ranks that BOTH send first, ungrouped: the textbook deadlock
    rc = run(lambda: lib.ncclSend(x0.data_ptr(), 1000, DOUBLE, 1, A[0], sp[0]), lambda: lib.ncclSend(x1.data_ptr(), 1000, DOUBLE, 0, A[1], sp[1]))
    checks["send_send_deadlock_detected"] = rc[0] != 0 and rc[1] != 0
    checks["self_send_refused"] = lib.ncclSend(x0.data_ptr(), 10, DOUBLE, 0, A[0], sp[0]) != 0
    checks["bad_peer_refused"] = lib.ncclSend(x0.data_ptr(), 10, DOUBLE, 2, A[0], sp[0]) != 0
    checks["bad_type_refused"] = lib.ncclSend(x0.data_ptr(), 10, 99, 1, A[0], sp[0]) != 0
    checks["group_end_without_start_refused"] = lib.ncclGroupEnd() != 0
    lib.mock_rccl_stats(st)
    checks["errors_counted_nothing_left_pending"] = st[4] >= 12 and st[5] == 0
    for cs in (A, B, T):
        for c in cs:
            assert lib.ncclCommDestroy(c) == 0
    checks["double_destroy_refused"] = lib.ncclCommDestroy(A[0]) != 0
    print(json.dumps(checks))
    return 0 if all(checks.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
