"""Both several-devices-in-one-process solvers of the C ABI -- gh_mgpu_* (dense, 2-D block-cyclic, RCCL) and
gh_hodlr_mgpu_* (HODLR, tree split) -- on whatever devices this process sees, each against the single-GPU solver
on the same inputs.  Prints ONE JSON object.  bench.py runs it in a child process with a time limit after its own
timed region (N = 1 only): on a box with several MI355X this is the only place where either form meets a second
physical device; on a one-GPU box the HODLR split runs on "virtual devices" (the same GPU listed twice).

usage: abi_multi_device_probe.py [--dense-n N] [--hodlr-n N ...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def inputs(n):
    rng = np.random.RandomState(1234)
    x = np.sort(rng.uniform(0, 10, n))
    return x, 0.1 * np.ones(n), np.sin(x)


def best_of(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        v = fn()
        best = min(best, time.perf_counter() - t0)
    return best, v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dense-n", type=int, default=32768)
    ap.add_argument("--hodlr-n", type=int, nargs="*", default=[262144, 2097152])
    ap.add_argument("--dense-kernel", default="matern32", choices=["expsquared", "matern32"])
    ap.add_argument("--devices", type=int, default=0, help="use the first D devices (0 = all visible, at most 16)")
    ap.add_argument("--no-hodlr", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="dense: do not time the single-GPU solver beside the sharded one")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import george_amd
    from george_amd import kernels, BasicSolver, HODLRSolver, MultiGPUSolver, MultiGPUHODLRSolver
    ndev = george_amd.device_count()
    out = {"devices_visible": ndev}
    if args.devices > 0:
        if ndev < args.devices:
            out["dense_rccl"] = {"error": "%d devices asked for, %d visible" % (args.devices, ndev)}
            print(json.dumps(out))
            return
        ndev = args.devices
    if ndev < 1:
        print(json.dumps(out))
        return
    # ---- HODLR, tree split
    P = 1
    while P * 2 <= min(ndev, 16):
        P *= 2
    devs = list(range(P)) if P > 1 else [0, 0]
    if not args.no_hodlr:
        out["hodlr_split"] = {"devices": devs, "virtual": P == 1, "cases": []}
    for n in ([] if args.no_hodlr else args.hodlr_n):
        try:
            x, yerr, y = inputs(n)
            kernel = float(np.var(y)) * kernels.ExpSquaredKernel(1.0)
            X = np.ascontiguousarray(x[:, None])
            kw = dict(tol=1e-10, min_size=100, seed=42)
            one = HODLRSolver(kernel, **kw)
            split = MultiGPUHODLRSolver(kernel, devices=devs, **kw)

            def step(s):
                s.compute(X, yerr)
                return -0.5 * (s.dot_solve(y) + s.log_determinant + n * np.log(2 * np.pi))
            t1, ll1 = best_of(lambda: step(one))
            tp, llp = best_of(lambda: step(split))
            out["hodlr_split"]["cases"].append({
                "n": n, "single_gpu_s": t1, "split_s": tp, "speedup": t1 / tp, "ll_single": ll1, "ll_split": llp,
                "rel": abs(llp - ll1) / abs(ll1), "same_ranks": split.ranks() == one.ranks(), "rows": split.rows()})
            del one, split
        except Exception as e:                                   # keep going: the other cases still tell something
            out["hodlr_split"]["cases"].append({"n": n, "error": repr(e)})
    # ---- dense, 2-D block-cyclic over every visible device (RCCL refuses one device listed twice: real devices only)
    if ndev >= 2:
        try:
            n = args.dense_n
            x, yerr, y = inputs(n)
            K_ = kernels.Matern32Kernel if args.dense_kernel == "matern32" else kernels.ExpSquaredKernel
            kernel = float(np.var(y)) * K_(1.0)
            X, sig = np.ascontiguousarray(x[:, None]), np.sqrt(yerr ** 2 + 1.25e-12)
            s = MultiGPUSolver(kernel, devices=list(range(min(ndev, 16))), transport="rccl")

            def stepd(q):
                q.compute(X, sig)
                return -0.5 * (n * np.log(2 * np.pi) + q.log_determinant) - 0.5 * q.dot_solve(y)
            tp, llp = best_of(lambda: stepd(s), reps=args.reps)
            pr, pc, nb = s.grid_shape()
            out["dense_rccl"] = {"n": n, "kernel": args.dense_kernel, "devices": min(ndev, 16), "grid": "%dx%d" % (pr, pc), "nb": nb,
                                 "sharded_s": tp, "ll_sharded": llp}
            del s
            if not args.no_single:
                d = BasicSolver(kernel)
                t1, ll1 = best_of(lambda: stepd(d), reps=args.reps)
                out["dense_rccl"].update({"single_gpu_s": t1, "speedup": t1 / tp, "ll_single": ll1, "rel": abs(llp - ll1) / abs(ll1)})
        except Exception as e:
            out["dense_rccl"] = {"error": repr(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
