"""The 8-GPU question put in numbers: does compute()+log_likelihood() at N = 65536 fit
1.407 s / 6 = 0.234 s on 8 MI355X, for which grid and tile edge, and what is the chain made of?

Two halves.

  collect   (on a GPU box)   python scripts/scale_model.py collect OUT.json [--n 65536]
      For every configuration (grid Pr x Pc, tile edge nb) the ABI multi-GPU solver (gh_mgpu_*, george_amd/csrc/gh_mgpu.hip)
      is run as W = Pr Pc "virtual devices" on the ONE GPU of the box (transport = peer copies) in trace mode: every phase
      of every rank and step -- kernel-matrix build, potrf, column TRSM, update of block column k+1, the rest of the trailing
      update, and the four kinds of transfer -- is bracketed by HIP events, and COMPUTE phases run one at a time over all
      ranks and to completion, so each duration is that of a rank alone on its GPU: real kernels, real shapes, real data
      (the log-determinant of the traced run is checked against the single-GPU solver).  Also measured: the wall clock of
      the same virtual configuration untraced, full and chain-only (GH_MGPU_CHAIN_ONLY: no trailing update but block column
      k+1, no bulk gather), and the one-device run (W = 1) with its trace.

  report    (anywhere)       python scripts/scale_model.py report OUT.json > profiles/r04/scale_model.md
      Replays the schedule of rank_factor() -- three in-order queues per rank (chain, bulk, update), the event waits between
      them, rendezvous at every transfer -- with the measured compute durations and a LINK MODEL for the transfers
      (time = latency + bytes over one xGMI link at the achieved bandwidth assumed; a broadcast to g - 1 peers uses g - 1
      links at once).  Check of the replay itself: the W = 1 trace replayed must give the measured W = 1 wall clock.
"""
import json
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PH = {"potrf": 0, "trsm": 1, "bcol": 2, "rest": 3, "lkk": 4, "rowx": 5, "ahead": 6, "gather": 7, "build": 8}
CONFIGS = [(8, 1, 1024), (8, 1, 512), (4, 2, 1024), (4, 2, 512), (2, 4, 1024), (2, 4, 512)]


# ------------------------------------------------------------------------------------------------ collect
def collect(out_path, n, configs=None, merge=None):
    import george_amd
    from george_amd import kernels, BasicSolver, MultiGPUSolver
    rng = np.random.RandomState(1234)
    x = np.sort(rng.uniform(0, 10, n))
    yerr, y = 0.1 * np.ones(n), np.sin(x)
    kernel = float(np.var(y)) * kernels.ExpSquaredKernel(1.0)
    X, sig = np.ascontiguousarray(x[:, None]), np.sqrt(yerr ** 2 + 1.25e-12)
    res = {"n": n, "configs": []}
    d = BasicSolver(kernel)
    d.compute(X, sig)
    t0 = time.perf_counter(); d.compute(X, sig); q0 = d.dot_solve(y); res["single_gpu_s"] = time.perf_counter() - t0
    ld0 = d.log_determinant
    del d
    BasicSolver.release_pool()

    def wall(s, reps=2):
        s.compute(X, sig)
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); s.compute(X, sig); best = min(best, time.perf_counter() - t0)
        return best

    for (Pr, Pc, nb) in (configs if configs else [(1, 1, 1024)] + CONFIGS):
        W = Pr * Pc
        c = {"Pr": Pr, "Pc": Pc, "nb": nb, "W": W, "snake": bool(Pc == 1 and Pr > 1)}
        try:
            kw = dict(devices=[0] * W, nb=nb, grid=(Pr, Pc), transport="copy")
            s = MultiGPUSolver(kernel, **kw)
            c["wall_full_s"] = wall(s)
            c["logdet_rel"] = abs(s.log_determinant - ld0) / abs(ld0)
            t0 = time.perf_counter(); qq = s.dot_solve(y); c["dot_solve_s"] = time.perf_counter() - t0
            c["quad_rel"] = abs(qq - q0) / abs(q0)
            del s
            s = MultiGPUSolver(kernel, chain_only=True, **kw)
            c["wall_chain_only_s"] = wall(s)
            del s
            s = MultiGPUSolver(kernel, trace=True, **kw)
            s.compute(X, sig)
            s.compute(X, sig)
            c["traced_logdet_rel"] = abs(s.log_determinant - ld0) / abs(ld0)
            c["trace"] = np.round(s.trace(), 5).tolist()
            del s
        except Exception as e:                                   # keep the other configurations
            c["error"] = repr(e)
        res["configs"].append(c)
        print("collected", Pr, Pc, nb, {k: v for k, v in c.items() if k != "trace"}, flush=True)
    if merge:                                                    # add to an earlier collection (its single-GPU time is kept beside this one's)
        import gzip
        old = json.load(gzip.open(merge, "rt") if merge.endswith(".gz") else open(merge))
        have = {(c["Pr"], c["Pc"], c["nb"]) for c in res["configs"]}
        for c in res["configs"]:
            c["single_gpu_s_of_its_session"] = res["single_gpu_s"]
        res["configs"] = [c for c in old["configs"] if (c["Pr"], c["Pc"], c["nb"]) not in have] + res["configs"]
        res["single_gpu_s"] = old["single_gpu_s"]
    with open(out_path, "w") as f:
        json.dump(res, f)


# ------------------------------------------------------------------------------------------------ replay
def prow(i, Pr, snake):
    if not snake:
        return i % Pr
    t = i % (2 * Pr)
    return t if t < Pr else 2 * Pr - 1 - t


def replay(cfg, n, link_gbs, lat_us, shared_gpu=False, chain_only=False, hbm_copy_gbs=2500.0, median_rate=False):
    """finish time (s) of rank_factor()'s schedule with measured compute durations and modelled transfers.
    shared_gpu: all ranks' compute phases contend for ONE device (exclusive use, earliest-ready first) and transfers are
    device-to-device copies -- the virtual-device configuration the trace was taken on."""
    Pr, Pc, nb, W, snake = cfg["Pr"], cfg["Pc"], cfg["nb"], cfg["W"], cfg["snake"]
    nt = -(-n // nb)
    T = {}
    for r_, k_, ph_, ms_, u_ in cfg["trace"]:
        T[(int(r_), int(k_), int(ph_))] = (ms_ * 1e-3, u_)
    if median_rate:
        # the update phases priced at the MEDIAN rate the ranks reached at that step: on virtual devices the rank whose update
        # runs first shares the one GPU's HBM with the other ranks' gather copies (rank 0: 57-58 TFLOP/s against 66-67 for the
        # others at the same shapes) -- an artefact of eight ranks on one device, not of the rank
        for ph_ in (PH["bcol"], PH["rest"]):
            for k_ in range(-(-n // cfg["nb"])):
                rates = sorted(T[(r_, k_, ph_)][1] / T[(r_, k_, ph_)][0] for r_ in range(cfg["W"]) if (r_, k_, ph_) in T and T[(r_, k_, ph_)][0] > 0)
                if not rates:
                    continue
                med = rates[len(rates) // 2]
                for r_ in range(cfg["W"]):
                    if (r_, k_, ph_) in T:
                        T[(r_, k_, ph_)] = (T[(r_, k_, ph_)][1] / med, T[(r_, k_, ph_)][1])
    bw = (hbm_copy_gbs if shared_gpu else link_gbs) * 1e9
    lat = (5.0 if shared_gpu else lat_us) * 1e-6

    def xfer(nbytes, links=1):
        return lat + nbytes / links / bw

    rk = lambda pr, pc: pr * Pc + pc
    # per rank: three in-order queues of ops; an op = dict(kind, dur, waits=[event keys], record=event key, group=collective id)
    Q = {(r, s): [] for r in range(W) for s in ("sp", "sg", "st")}
    for r in range(W):
        pr, pc = divmod(r, Pc)
        rows = [i for i in range(nt) if prow(i, Pr, snake) == pr]
        cols = [j for j in range(nt) if j % Pc == pc]

        def comp(stream, k, ph, waits=(), record=None):
            if (r, k, ph) in T:
                Q[(r, stream)].append(dict(kind="c", dur=T[(r, k, ph)][0], waits=list(waits), record=record))
            elif waits or record:
                Q[(r, stream)].append(dict(kind="n", dur=0.0, waits=list(waits), record=record))

        def coll(stream, gid, members, nbytes, links=1, waits=(), record=None):
            Q[(r, stream)].append(dict(kind="x", dur=xfer(nbytes, links), waits=list(waits), record=record, gid=gid, members=members))

        comp("st", -1, PH["build"], record=("bcol", r, 0))

        def panel(k):
            kr, kc = prow(k, Pr, snake), k % Pc
            in_col = pc == kc
            Q[(r, "sp")].append(dict(kind="n", dur=0.0, waits=[("bcol", r, k)], record=None))
            if pr == kr and in_col:
                comp("sp", k, PH["potrf"])
            if k < nt - 1:
                if in_col and Pr > 1:
                    coll("sp", ("lkk", k), [rk(p, kc) for p in range(Pr)], 8.0 * (nb * nb + (nb // 128) * 128 * 128))
                m = len([i for i in rows if i > k]) * nb
                if in_col and m > 0:
                    comp("sp", k, PH["trsm"])
                if Pc > 1 and m > 0:
                    coll("sp", ("rowx", k, pr), [rk(pr, c) for c in range(Pc)], 8.0 * m * nb)
                if pc == (k + 1) % Pc and Pr > 1:
                    coll("sp", ("ahead", k), [rk(p, pc) for p in range(Pr)], 8.0 * nb * nb)
            Q[(r, "sp")].append(dict(kind="n", dur=0.0, waits=[], record=("fast", r, k)))

        def gather(k):
            waits = [("fast", r, k)]
            if k < nt - 1 and not chain_only and Pr > 1:
                mine = [j for j in cols if j >= k + 2]
                if mine:
                    # my incoming tiles by source process row: each source has its own link
                    per_src = {}
                    for j in mine:
                        per_src[prow(j, Pr, snake)] = per_src.get(prow(j, Pr, snake), 0) + 1
                    per_src.pop(pr, None)
                    worst = max(per_src.values()) if per_src else 0
                    # (in process column pc every member both sends and receives; the busiest link of the column decides)
                    coll("sg", ("gather", k, pc), [rk(p, pc) for p in range(Pr)], 8.0 * nb * nb * max(worst, 1), waits=waits)
                    waits = []
            Q[(r, "sg")].append(dict(kind="n", dur=0.0, waits=waits, record=("panel", r, k)))

        panel(0)
        gather(0)
        for k in range(nt):
            if k == nt - 1:
                Q[(r, "st")].append(dict(kind="n", dur=0.0, waits=[("fast", r, k)], record=None))
                break
            comp("st", k, PH["bcol"], waits=[("fast", r, k)], record=("bcol", r, k + 1))
            if (r, k, PH["bcol"]) not in T:
                pass
            panel(k + 1)
            gather(k + 1)
            if chain_only:
                Q[(r, "st")].append(dict(kind="n", dur=0.0, waits=[("panel", r, k)], record=None))
            else:
                comp("st", k, PH["rest"], waits=[("panel", r, k)])
    # the gather's worst-link figure must be one number per collective: take the maximum over its members
    worst = {}
    for q in Q.values():
        for op in q:
            if op["kind"] == "x":
                worst[op["gid"]] = max(worst.get(op["gid"], 0.0), op["dur"])
    for q in Q.values():
        for op in q:
            if op["kind"] == "x":
                op["dur"] = worst[op["gid"]]
    # ---- run
    ready = {k: 0.0 for k in Q}                   # when the queue's previous op ended
    head = {k: 0 for k in Q}
    ev = {}
    gpu_free = 0.0
    busy = {"c": 0.0, "x": 0.0}
    total_ops = sum(len(q) for q in Q.values())
    done = 0
    while done < total_ops:
        # candidate ops: head of each queue whose waits are recorded; collectives need every member at the same op
        best, best_t = None, None
        for key, q in Q.items():
            if head[key] >= len(q):
                continue
            op = q[head[key]]
            if any(w not in ev for w in op["waits"]):
                continue
            t = max([ready[key]] + [ev[w] for w in op["waits"]])
            if op["kind"] == "x":
                ok, tt = True, t
                for m in op["members"]:
                    found = False
                    for s_ in ("sp", "sg"):
                        kk = (m, s_)
                        if head[kk] < len(Q[kk]):
                            o2 = Q[kk][head[kk]]
                            if o2["kind"] == "x" and o2["gid"] == op["gid"]:
                                if any(w not in ev for w in o2["waits"]):
                                    break
                                tt = max([tt, ready[kk]] + [ev[w] for w in o2["waits"]])
                                found = True
                                break
                    if not found:
                        ok = False
                        break
                if not ok:
                    continue
                t = tt
            if op["kind"] == "c" and shared_gpu:
                t = max(t, gpu_free)
            if best is None or t < best_t:
                best, best_t = (key, op), t
        if best is None:
            raise RuntimeError("replay dead-locked (schedule bug)")
        key, op = best
        end = best_t + op["dur"]
        if op["kind"] == "x":
            for m in op["members"]:
                for s_ in ("sp", "sg"):
                    kk = (m, s_)
                    if head[kk] < len(Q[kk]):
                        o2 = Q[kk][head[kk]]
                        if o2["kind"] == "x" and o2["gid"] == op["gid"]:
                            ready[kk] = end
                            head[kk] += 1
                            done += 1
                            if o2["record"]:
                                ev[o2["record"]] = end
                            break
            busy["x"] += op["dur"]
        else:
            if op["kind"] == "c":
                busy["c"] += op["dur"]
                if shared_gpu:
                    gpu_free = end
            ready[key] = end
            head[key] += 1
            done += 1
            if op["record"]:
                ev[op["record"]] = end
    return max(ready.values())


def terms(cfg, n):
    """sums over the steps of what the chain and the update are made of (ms): potrf, max-over-ranks TRSM / block column /
    rest per step, and the per-rank update totals"""
    W = cfg["W"]
    by = {}
    for r_, k_, ph_, ms_, u_ in cfg["trace"]:
        by.setdefault((int(k_), int(ph_)), {})[int(r_)] = (ms_, u_)
    out = {}
    for name in ("potrf", "trsm", "bcol", "rest", "build"):
        ph = PH[name]
        out[name + "_max_ms"] = sum(max(v[0] for v in d.values()) for (k, p), d in by.items() if p == ph)
    per_rank_update = [sum(d[r][0] for (k, p), d in by.items() if p in (PH["bcol"], PH["rest"]) and r in d) for r in range(W)]
    per_rank_flops = [sum(d[r][1] for (k, p), d in by.items() if p in (PH["bcol"], PH["rest"]) and r in d) for r in range(W)]
    out["update_per_rank_ms"] = per_rank_update
    out["update_tflops_per_rank"] = [f / (t * 1e-3) * 1e-12 if t > 0 else 0.0 for f, t in zip(per_rank_flops, per_rank_update)]
    out["update_imbalance"] = max(per_rank_update) / (sum(per_rank_update) / W)
    nb, Pr, Pc = cfg["nb"], cfg["Pr"], cfg["Pc"]
    nt = -(-n // nb)
    # bytes on ONE link inside the chain, summed over the steps
    chain_bytes = 0.0
    for k in range(nt - 1):
        if Pr > 1:
            chain_bytes += 8.0 * (nb * nb + (nb // 128) * 128 * 128) + 8.0 * nb * nb
        if Pc > 1:
            chain_bytes += 8.0 * nb * nb * max(len([i for i in range(k + 1, nt) if prow(i, Pr, cfg["snake"]) == p]) for p in range(Pr))
    out["chain_link_bytes"] = chain_bytes
    return out


def report(path):
    import gzip
    res = json.load(gzip.open(path, "rt") if path.endswith(".gz") else open(path))
    n = res["n"]
    t1 = res["single_gpu_s"]
    budget = t1 / 6.0
    P = print
    P("# Scale model: N = %d on 8 MI355X -- which grid, which tile edge, and does it fit %.0f ms?\n" % (n, budget * 1e3))
    P("Source: `scripts/scale_model.py` (`collect` on a 1-GPU box -> `%s`; `report` replays the schedule).  Single-GPU" % os.path.basename(path))
    P("`compute()+dot_solve()` measured in the same session: **%.4f s** -> a 6x speed-up allows **%.1f ms**.\n" % (t1, budget * 1e3))
    P("Every compute duration below was MEASURED (HIP events) on the real kernels at the real per-rank shapes on real data: the ABI")
    P("multi-GPU solver run as 8 virtual devices on one GPU in trace mode, compute phases one at a time (`GH_MGPU_TRACE`,")
    P("`george_amd/csrc/gh_mgpu.hip`).  Transfers are MODELLED: `latency + bytes / link bandwidth`, a broadcast to g - 1 peers over g - 1")
    P("links at once, a gather bounded by its busiest link.  Link: xGMI, 153.6 GB/s per link both directions = 76.8 GB/s one way;")
    P("three assumptions for the achieved one-way rate (45 / 60 / 75 GB/s) and 25 us per transfer (RCCL kernel launch + rendezvous).\n")
    one = [c for c in res["configs"] if c["W"] == 1 and "trace" in c]
    if one:
        c = one[0]
        sim = replay(c, n, 60.0, 25.0)
        P("## Check of the replay: one device\n")
        P("W = 1 (grid 1x1, nb = %d): measured wall clock of `gh_mgpu_compute` **%.4f s**; its trace replayed: **%.4f s** (%.1f %%).\n"
          % (c["nb"], c["wall_full_s"], sim, 100.0 * (sim / c["wall_full_s"] - 1.0)))
    P("## Configurations (W = 8)\n")
    P("| grid | nb | snake | chain compute: sum_k potrf / max TRSM / max block column (ms) | bytes on one link inside the chain | update per rank: mean ms (TFLOP/s), max/mean | predicted 8-GPU time at 45 / 60 / 75 GB/s (ms) | speed-up at 60 GB/s | same, updates at the ranks' median rate (ms, x) | virtual 8-on-1: measured full / chain-only (s); replayed as shared GPU |")
    P("|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for c in res["configs"]:
        if c["W"] != 8 or "trace" not in c:
            if "error" in c:
                P("| %dx%d | %d | | error: %s | | | | | |" % (c["Pr"], c["Pc"], c["nb"], c["error"][:80]))
            continue
        tm = terms(c, n)
        preds = [replay(c, n, bw, 25.0) for bw in (45.0, 60.0, 75.0)]
        chain_pred = replay(c, n, 60.0, 25.0, chain_only=True)
        pred_med = replay(c, n, 60.0, 25.0, median_rate=True)
        sh_full = replay(c, n, 60.0, 25.0, shared_gpu=True)
        sh_chain = replay(c, n, 60.0, 25.0, shared_gpu=True, chain_only=True)
        upd = tm["update_per_rank_ms"]
        rows.append((c, tm, preds, chain_pred, pred_med))
        P("| %dx%d | %d | %s | %.1f / %.1f / %.1f | %.2f GB (%.0f ms at 60 GB/s) | %.1f (%.1f), %.3f | **%.1f / %.1f / %.1f** | **%.2fx** | %.1f, %.2fx | %.3f / %.3f; %.3f / %.3f |"
          % (c["Pr"], c["Pc"], c["nb"], "yes" if c["snake"] else "no", tm["potrf_max_ms"], tm["trsm_max_ms"], tm["bcol_max_ms"],
             tm["chain_link_bytes"] * 1e-9, tm["chain_link_bytes"] / 60e9 * 1e3, sum(upd) / len(upd),
             sum(tm["update_tflops_per_rank"]) / len(upd), tm["update_imbalance"], preds[0] * 1e3, preds[1] * 1e3, preds[2] * 1e3,
             t1 / preds[1], pred_med * 1e3, t1 / pred_med, c["wall_full_s"], c["wall_chain_only_s"], sh_full, sh_chain))
    P("")
    P("Columns: *chain compute* = per-step potrf on the diagonal owner + the slowest rank's TRSM + the slowest rank's block-column update,")
    P("summed over the steps (what the chain costs with free transfers); *bytes on one link inside the chain* = L_kk + diagonal inverses +")
    P("panel tile k+1 (+ the row panel of the busiest process row when Pc > 1) per step, summed; *update per rank* = block column + rest of")
    P("the trailing update, per-rank totals; *predicted* = replay of the three-queue schedule; the last column compares the measured wall clock")
    P("of the virtual configuration (8 ranks sharing ONE GPU, un-traced) with the same replay run as if all compute shared one device")
    P("exclusively -- a pessimistic stand-in for eight ranks whose small kernels overlap on the real shared GPU.\n")
    if rows:
        best = min(rows, key=lambda r: r[2][1])
        c, tm, preds, chain_pred, pred_med = best
        P("## Verdict\n")
        P("Best: **%dx%d, nb = %d** -> **%.1f ms at 60 GB/s per link = %.2fx** the single GPU (budget %.1f ms; %.1f ms at 45 GB/s, %.1f at 75)."
          % (c["Pr"], c["Pc"], c["nb"], preds[1] * 1e3, t1 / preds[1], budget * 1e3, preds[0] * 1e3, preds[2] * 1e3))
        P("Its chain alone (chain-only replay, transfers included): %.1f ms; its per-rank update: %.1f ms -- the step is bound by the"
          % (chain_pred * 1e3, max(tm["update_per_rank_ms"])))
        P("%s.  With every rank's updates at the median rate of the eight (the rank whose update runs first on the shared test GPU" % ("UPDATES (the chain hides behind them)" if chain_pred < max(tm["update_per_rank_ms"]) * 1e-3 else "CHAIN"))
        P("competes with the others' gather copies for the one HBM): **%.1f ms = %.2fx**.\n" % (pred_med * 1e3, t1 / pred_med))
    curve = [c for c in res["configs"] if c["Pc"] == 1 and c["nb"] == 1024 and "trace" in c and (c["W"] == 1 or c["snake"])]
    if len(curve) > 2:
        P("## Predicted curve, whole tile rows per GPU (W x 1, snake), nb = 1024, 60 GB/s per link\n")
        P("| GPUs | predicted ms (45 / 60 / 75 GB/s) | speed-up at 60 | efficiency | update per rank ms (TFLOP/s) | chain alone ms |")
        P("|---|---|---|---|---|---|")
        for c in sorted(curve, key=lambda c: c["W"]):
            if c["W"] == 1:
                P("| 1 | %.1f (measured, single-GPU solver `gh_chol_*`) | 1.00x | 1.00 | | |" % (t1 * 1e3))
                continue
            tm = terms(c, n)
            pr = [replay(c, n, bw, 25.0) for bw in (45.0, 60.0, 75.0)]
            upd = tm["update_per_rank_ms"]
            P("| %d | %.1f / %.1f / %.1f | %.2fx | %.2f | %.1f (%.1f) | %.1f |" % (c["W"], pr[0] * 1e3, pr[1] * 1e3, pr[2] * 1e3, t1 / pr[1], t1 / pr[1] / c["W"],
                                                                   sum(upd) / len(upd), sum(tm["update_tflops_per_rank"]) / len(upd),
                                                                   replay(c, n, 60.0, 25.0, chain_only=True) * 1e3))
        P("")
    print(json.dumps({"n": n, "single_gpu_s": t1, "budget_s": budget,
                      "configs": [{"grid": "%dx%d" % (c["Pr"], c["Pc"]), "nb": c["nb"], "pred_ms_45_60_75": [p * 1e3 for p in preds], "pred_ms_60_median_rate": pm * 1e3} for c, tm, preds, _, pm in rows]}),
          file=sys.stderr)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "collect":
        nn = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 65536
        cfgs = None
        if "--configs" in sys.argv:                              # e.g. --configs 2x1x1024,4x1x1024
            cfgs = [tuple(int(v) for v in t.split("x")) for t in sys.argv[sys.argv.index("--configs") + 1].split(",")]
        mg = sys.argv[sys.argv.index("--merge") + 1] if "--merge" in sys.argv else None
        collect(sys.argv[2], nn, cfgs, mg)
    elif len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        sys.exit(__doc__)
