"""What RCCL demands of george_amd/distributed.py and gloo does not check: every member of a communicator must issue the
SAME collectives in the SAME order with matching sizes and roots -- under gloo a slip can still complete, under RCCL it
hangs or corrupts -- and two ranks that share several communicators must interleave them identically (all of them are
queued on one stream per rank).  Here every rank of a gloo world records each collective it issues (operation,
communicator members, root, element count / split sizes) through recording wrappers around torch.distributed, the
branches gloo cannot take itself are made to run (all_gather_into_tensor is emulated with all_gather so the driver keeps
the branch RCCL takes; the uneven all_to_all_single of the row exchange is forced with GEORGE_AMD_DIST_ROWXCHG=a2a), and
rank 0 checks the logs of all ranks against each other: compute (factorisation with look-ahead off and on), the sweeps
with one and several right-hand sides, apply_sqrt, get_inverse, and a not-positive-definite matrix."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _install_recorder(log):
    """wrap the collectives distributed.py uses; returns the originals"""
    orig = {name: getattr(dist, name) for name in ("broadcast", "all_reduce", "reduce", "all_gather", "all_gather_into_tensor", "all_to_all_single")}

    def members(group):
        return tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))

    def broadcast(t, src, group=None, **kw):
        log.append(("broadcast", members(group), int(src), int(t.numel())))
        return orig["broadcast"](t, src=src, group=group, **kw)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, **kw):
        log.append(("all_reduce", members(group), str(op), int(t.numel())))
        return orig["all_reduce"](t, op=op, group=group, **kw)

    def reduce(t, dst, op=dist.ReduceOp.SUM, group=None, **kw):
        log.append(("reduce", members(group), int(dst), int(t.numel())))
        return orig["reduce"](t, dst=dst, op=op, group=group, **kw)

    def all_gather(lst, t, group=None, **kw):
        log.append(("all_gather", members(group), -1, int(t.numel())))
        return orig["all_gather"](lst, t, group=group, **kw)

    def all_gather_into_tensor(out, t, group=None, **kw):
        # gloo has no such operation: the same bytes through all_gather, recorded under the name RCCL will run
        m = members(group)
        log.append(("all_gather_into_tensor", m, -1, int(t.numel())))
        assert out.numel() == len(m) * t.numel() and out.is_contiguous() and t.is_contiguous()
        parts = list(out.view(len(m), -1).unbind(0))
        orig["all_gather"](parts, t.reshape(-1), group=group)

    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, **kw):
        log.append(("all_to_all_single", members(group), tuple(int(v) for v in output_split_sizes), tuple(int(v) for v in input_split_sizes)))
        assert out.numel() == sum(output_split_sizes) and inp.numel() == sum(input_split_sizes)
        return orig["all_to_all_single"](out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group, **kw)

    for name, fn in (("broadcast", broadcast), ("all_reduce", all_reduce), ("reduce", reduce), ("all_gather", all_gather),
                     ("all_gather_into_tensor", all_gather_into_tensor), ("all_to_all_single", all_to_all_single)):
        setattr(dist, name, fn)
    return orig


def check_logs(logs):
    """logs[rank] = [(op, members, root-or-splits, count), ...] in issue order.  Returns counts per operation name."""
    world = len(logs)
    groups = sorted({e[1] for lg in logs for e in lg})
    for g in groups:
        seqs = {r: [e for e in logs[r] if e[1] == g] for r in g}
        for r in range(world):
            if r not in g:
                assert not [e for e in logs[r] if e[1] == g], "rank %d issued a collective on a communicator it is not in: %r" % (r, g)
        n0 = len(seqs[g[0]])
        for r in g:
            assert len(seqs[r]) == n0, "communicator %r: rank %d issued %d collectives, rank %d issued %d" % (g, g[0], n0, r, len(seqs[r]))
        for i in range(n0):
            first = seqs[g[0]][i]
            for r in g:
                e = seqs[r][i]
                assert e[0] == first[0], "communicator %r, call %d: %s on rank %d, %s on rank %d" % (g, i, first[0], g[0], e[0], r)
                if e[0] == "all_to_all_single":
                    for peer in g:          # what I expect from `peer` is what `peer` sends to me
                        assert e[2][g.index(peer)] == seqs[peer][i][3][g.index(r)], (g, i, r, peer, e, seqs[peer][i])
                else:
                    assert e[2:] == first[2:], "communicator %r, call %d: %r on rank %d, %r on rank %d" % (g, i, first, g[0], e, r)
    # two ranks that share communicators interleave them the same way
    for a in range(world):
        for b in range(a + 1, world):
            sa = [(e[0], e[1]) for e in logs[a] if a in e[1] and b in e[1]]
            sb = [(e[0], e[1]) for e in logs[b] if a in e[1] and b in e[1]]
            assert sa == sb, "ranks %d and %d interleave their shared communicators differently" % (a, b)
    out = {}
    for lg in logs:
        for e in lg:
            out[e[0]] = out.get(e[0], 0) + 1
    return out, groups


def _worker(rank, world, port, n, nb, grid, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GEORGE_AMD_DIST_ROWXCHG"] = "a2a"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import george_amd.kernels as K
        from george_amd.distributed import DistributedBasicSolver
        from np_tile_ops import NumpyTileOps
        log = []
        orig = _install_recorder(log)
        rng = np.random.RandomState(7)
        x = np.sort(rng.uniform(0, 10, n))
        y = np.sin(x)
        kernel = 0.5 * K.Matern32Kernel(1.3)
        solver = DistributedBasicSolver(kernel, nb=nb, ops=NumpyTileOps(kernel), grid=grid)
        solver.compute(x[:, None], 0.1)
        took_rccl_gather = solver._chol._gather_into_tensor if solver._chol.Pr > 1 else True
        solver.dot_solve(y)
        solver.apply_inverse(np.stack([y, np.cos(3 * x), x ** 2], axis=1))
        solver.apply_sqrt(np.stack([y, x]).copy())
        solver.get_inverse()
        bad = DistributedBasicSolver(K.CosineKernel(log_period=0.0), nb=nb, ops=NumpyTileOps(K.CosineKernel(log_period=0.0)), grid=grid)
        try:
            bad.compute(x[:, None], 0.0)
            raise AssertionError("singular matrix accepted")
        except np.linalg.LinAlgError:
            pass
        for name, fn in orig.items():
            setattr(dist, name, fn)
        logs = [None] * world
        dist.all_gather_object(logs, log)
        if rank == 0:
            q.put((logs, bool(took_rccl_gather), solver.log_determinant))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,grid", [(4, 900, 128, (2, 2)), (8, 1300, 128, (2, 4)), (3, 800, 128, None), (4, 700, 128, (1, 4))])
def test_every_rank_issues_the_same_collectives_in_the_same_order(world, n, nb, grid):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, grid, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        logs, took_rccl_gather, logdet = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(120)
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    counts, groups = check_logs(logs)
    Pr, Pc = grid if grid is not None else (world, 1)
    assert took_rccl_gather
    if Pr > 1:
        assert counts.get("all_gather_into_tensor", 0) > 0 and counts.get("all_gather", 0) == 0       # the branch RCCL takes
    if Pc > 1:
        assert counts.get("all_to_all_single", 0) > 0                                                  # the all-links row exchange
    assert counts.get("broadcast", 0) > 0 and counts.get("reduce", 0) > 0 and counts.get("all_reduce", 0) > 0
    assert np.isfinite(logdet)


def test_the_checker_itself():
    """a log with a swapped pair, a wrong root, a wrong count and a one-sided split must each be refused"""
    g = (0, 1)
    good = [[("broadcast", g, 0, 10), ("all_reduce", g, "SUM", 1)], [("broadcast", g, 0, 10), ("all_reduce", g, "SUM", 1)]]
    check_logs(good)
    for bad in ([good[0], good[1][::-1]],
                [good[0], [("broadcast", g, 1, 10), good[1][1]]],
                [good[0], [("broadcast", g, 0, 11), good[1][1]]],
                [good[0], good[1][:1]],
                [[("all_to_all_single", g, (0, 4), (0, 4))], [("all_to_all_single", g, (3, 0), (4, 0))]]):
        with pytest.raises(AssertionError):
            check_logs(bad)
    check_logs([[("all_to_all_single", g, (0, 4), (0, 3))], [("all_to_all_single", g, (3, 0), (4, 0))]])
    # two communicators shared by both ranks, interleaved differently
    h = (0, 1, 2)
    with pytest.raises(AssertionError):
        check_logs([[("broadcast", g, 0, 1), ("broadcast", h, 0, 1)], [("broadcast", h, 0, 1), ("broadcast", g, 0, 1)], [("broadcast", h, 0, 1)]])
