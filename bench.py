#!/usr/bin/env python
"""Headline benchmark: GP ``compute()`` + ``log_likelihood()`` (BASELINE.json metric).

One "step" = one full pass of the hot path on synthetic inputs that are already resident in HBM:
build K(x, x) + diag(yerr^2) on the device, blocked fp64 Cholesky, log-det, forward solve and
r^T K^-1 r -> log-likelihood.  Inputs follow the reference's only benchmark
(docs/tutorials/scaling.rst:56-59,67 / SURVEY.md 8d).

    python bench.py --gpus N --steps K --warmup W [--n 65536] [--nb 512]

N > 1 runs one rank per GPU (RCCL): the dense factorisation is sharded block-cyclically over the
ranks (george_amd/distributed.py) -- same N, strong scaling.  Either the caller launches the ranks
(``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``: WORLD_SIZE is set and must
equal N) or plain ``python bench.py --gpus N`` re-launches itself under torch.distributed.run with N
ranks; it never falls back to one GPU.  Rank 0 prints ONE JSON line; the N > 1 line carries its own
parity block against tests/golden/large.json and the run exits non-zero above 1e-6.

At N = 1 the same line also carries (all timed in this run, on the GPU the driver leased):
  parity            GPU vs the reference CPU path at the cpu_baseline sample size, and vs the committed
                    reference scalar at the headline size; the run FAILS if either is off by > 1e-6
  public_api        the same work through GP.compute / GP.log_likelihood on NumPy inputs (H2D included)
  roofline_kernel_build   the HBM-bound kernel-matrix build of the headline step
  config.also_configs1_N16384, config.also_C4 (HODLR, N = 262144), config.also_C5 (3-D, predict + grad)
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X datasheet fp64 matrix peak; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0            # HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md; 6.29 TB/s measured copy there)


def flops_alg(n):
    return n ** 3 / 3.0 + 2.0 * n ** 2           # potrf + potrs


def make_inputs(n, seed=1234):
    rng = np.random.RandomState(seed)
    x = np.sort(rng.uniform(0, 10, n))
    return x, 0.1 * np.ones(n), np.sin(x)


KERNEL_NAMES = {"expsquared": "ExpSquared", "matern32": "Matern32"}


def make_kernel(name, amp):
    """amp * <Kernel>(1.0): ExpSquared = the north-star target / configs[1], Matern32 = configs[2] (C3)"""
    import george_amd.kernels as K
    return float(amp) * getattr(K, KERNEL_NAMES[name] + "Kernel")(1.0)


def cpu_baseline(n_cpu):
    """The reference's CPU path on this host, bounded sample: its own compiled C++ kernel
    evaluator (oracle/_ref) when present + the same SciPy LAPACK calls as basic.py:64-70,102."""
    import george_amd.kernels as K
    from oracle import solver_np
    x, yerr, y = make_inputs(n_cpu)
    kernel = np.var(y) * K.ExpSquaredKernel(1.0)
    kind = solver_np.evaluator_kind()
    t0 = time.perf_counter()
    solver = solver_np.DenseOracle(kernel)
    ll = solver_np.gp_log_likelihood(solver, x[:, None], yerr, y)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()                     # the split SURVEY.md section 7 asks for: the build alone, again
    solver_np.kernel_matrix(kernel, x[:, None])
    dt_build = time.perf_counter() - t0
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    full = None
    try:                                                     # the only full-size CPU time there is: the golden generator's own run
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "large.json")))["NS"]
        full = {"n": g["n"], "seconds_total": g["seconds_total"], "seconds_kernel_build": g.get("seconds_build"),
                "seconds_factor": g.get("seconds_factor"),
                "where": "build container, 6 OpenBLAS threads (oracle/gen_golden_large.py NS): the reference's evaluator + LAPACK "
                         "on 8192-column blocks at the headline size"}
    except Exception:
        pass
    return {
        "full_size_run": full,
        "value": flops_alg(n_cpu) / dt * 1e-12, "unit": "TFLOP/s", "seconds": dt,
        "seconds_kernel_build": dt_build, "seconds_factor_and_solve": max(dt - dt_build, 0.0), "cores": int(threads),
        "kind": kind, "n": n_cpu, "log_likelihood": float(ll),
        "sample": "same workload at N=%d (one compute()+log_likelihood(); kernel build 1 thread, "
                  "LAPACK dpotrf/dpotrs %d threads); loglike=%.10g" % (n_cpu, threads, ll),
    }


def cpu_baseline_second_point(n_cpu, local_rank):
    """A second, larger same-box CPU point (default N = 32768, ~40 s on the GPU box's host cores): the reference's
    compiled evaluator for K + the LAPACK/BLAS routines of basic.py:68,87 applied to 8192-column blocks
    (oracle/solver_np.BlockedDenseOracle: this image's whole-matrix dpotrf is unreliable from n ~ 20000 up,
    oracle/potrf_probe.py), and the GPU's log-likelihood at the same N beside it."""
    import george_amd.kernels as K
    from oracle import solver_np
    x, yerr, y = make_inputs(n_cpu)
    kernel = np.var(y) * K.ExpSquaredKernel(1.0)
    t0 = time.perf_counter()
    ll = solver_np.gp_log_likelihood(solver_np.BlockedDenseOracle(kernel), x[:, None], yerr, y)
    dt = time.perf_counter() - t0
    jp = DenseJob(n_cpu, 0, local_rank, profile=False)
    jp.step()
    e, llg = run_timed(jp, 2, 0, lambda: None)
    jp.close()
    return {"n": n_cpu, "seconds": dt, "value": flops_alg(n_cpu) / dt * 1e-12, "unit": "TFLOP/s", "log_likelihood": float(ll),
            "gpu_seconds": e / 2, "gpu_over_cpu": dt / (e / 2), "rel": abs(llg - ll) / abs(ll),
            "how": "reference evaluator (1 thread) + dpotrf/dtrsm/dgemm on 8192-column blocks (all host threads)"}


class DenseJob(object):
    """compute()+log_likelihood() straight through the C ABI with device-resident inputs."""

    def __init__(self, n, nb, device, profile=True, lookahead=True, kernel="expsquared"):
        import torch
        from george_amd import _native as N
        from george_amd.program import DeviceKernel
        self.N, self.torch, self.n = N, torch, n
        x, yerr, y = make_inputs(n)
        self.amp = float(np.var(y))
        self.dk = DeviceKernel(make_kernel(kernel, self.amp))
        dev = torch.device("cuda", device)
        self.x = torch.from_numpy(x).to(dev)
        self.yerr = torch.from_numpy(np.sqrt(yerr ** 2 + 1.25e-12)).to(dev)     # gp.py:330 with default white noise
        self.y = torch.from_numpy(y).to(dev)
        torch.cuda.synchronize(dev)
        o = N.gh_chol_opts()
        o.device, o.nb, o.profile, o.lookahead = device, nb, int(profile), int(lookahead)
        self.h = N._vp()
        N.check(N.lib.gh_chol_create(C.byref(o), C.byref(self.h)))

    def step(self):
        N = self.N
        logdet, q = C.c_double(0.0), C.c_double(0.0)
        N.check(N.lib.gh_chol_compute(self.h, self.dk.handle, self.x.data_ptr(), self.n, 1,
                                      self.yerr.data_ptr(), C.byref(logdet)))
        N.check(N.lib.gh_chol_dot_solve(self.h, self.y.data_ptr(), C.byref(q)))
        return -0.5 * (self.n * np.log(2 * np.pi) + logdet.value) - 0.5 * q.value    # gp.py:333-335,396

    def profile(self):
        p = self.N.gh_chol_profile()
        self.N.check(self.N.lib.gh_chol_get_profile(self.h, C.byref(p)))
        return p

    def update_intervals(self):
        """(start ms, end ms, flops) of every trailing-update launch of the last step (HIP events, each pair on the
        stream its launch went to; times from the start of that step's compute())"""
        cnt = C.c_int32(0)
        self.N.check(self.N.lib.gh_chol_get_update_intervals(self.h, None, 0, C.byref(cnt)))
        buf = np.zeros((max(cnt.value, 1), 3))
        self.N.check(self.N.lib.gh_chol_get_update_intervals(self.h, self.N.ptr(buf), cnt.value, C.byref(cnt)))
        return buf[:cnt.value]

    def close(self):
        self.N.lib.gh_chol_destroy(self.h)


def syrk_standalone(job, device, m=32768):
    """The trailing-update kernel alone on the chip: C (m x m, lower tiles) -= A A^T for K = 1024 and 2048 through gh_dev_gemm,
    best of 3 launches each by HIP events on the launch stream (torch's current stream = the one gh_dev_gemm(None) uses)."""
    import torch
    N = job.N
    dp = C.POINTER(C.c_double)
    res = {"M": m}
    c = torch.zeros(m, m, dtype=torch.float64, device="cuda:%d" % device)
    tiles = (m // 128) * (m // 128 + 1) / 2
    for k in (1024, 2048):
        a = torch.randn(m, k, dtype=torch.float64, device="cuda:%d" % device)
        best = 1e30
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            N.check(N.lib.gh_dev_gemm(C.cast(c.data_ptr(), dp), c.stride(0), C.cast(a.data_ptr(), dp), a.stride(0),
                                      C.cast(a.data_ptr(), dp), a.stride(0), m, m, k, -1.0, 1.0, 4, None))
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1))
        tf = tiles * 2 * 128 * 128 * k / best * 1e-9
        res["K%d" % k] = {"ms": best, "tflops": tf, "frac": tf / PEAK_FP64_MFMA_TFLOPS}
        del a
    del c
    return res


def pmc_traffic(n):
    """Per-launch traffic of the trailing SYRK from the committed rocprofv3 --pmc passes
    (profiles/*/traffic_N<n>.json, produced by scripts/profile.sh + scripts/traffic_from_pmc.py:
    separate FETCH_SIZE / WRITE_SIZE passes, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024).  Counters
    cannot be read inside a timed run, so this is the most recent committed measurement."""
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_N%d.json" % n)))
    if not hits:
        return {"traffic": None}
    d = json.load(open(hits[-1]))
    return {"traffic": d["bytes_per_launch"], "traffic_unit": "bytes/launch leaving the L2s (Infinity-Cache hits included)",
            "traffic_algorithmic_bytes": d["algorithmic_bytes_per_launch"],
            "traffic_source": os.path.relpath(hits[-1], ROOT)}


class HodlrJob(object):
    """BASELINE config C4: compute()+log_likelihood() with the HODLR solver (tol = 1e-10)."""

    def __init__(self, n, device, tol=1e-10, min_size=100, seed=42):
        import torch
        import george_amd.kernels as K
        from george_amd import _native as N
        from george_amd.program import DeviceKernel
        self.N, self.n = N, n
        x, yerr, y = make_inputs(n)
        self.dk = DeviceKernel(float(np.var(y)) * K.ExpSquaredKernel(1.0))
        dev = torch.device("cuda", device)
        self.x = torch.from_numpy(x).to(dev)
        self.yerr = torch.from_numpy(np.sqrt(yerr ** 2 + 1.25e-12)).to(dev)
        self.y = torch.from_numpy(y).to(dev)
        torch.cuda.synchronize(dev)
        o = N.gh_hodlr_opts()
        o.device, o.min_size, o.seed, o.max_rank, o.tol = device, min_size, seed, 0, tol
        self.h = N._vp()
        N.check(N.lib.gh_hodlr_create(C.byref(o), C.byref(self.h)))

    def step(self):
        N = self.N
        logdet, q = C.c_double(0.0), C.c_double(0.0)
        N.check(N.lib.gh_hodlr_compute(self.h, self.dk.handle, self.x.data_ptr(), self.n, 1,
                                       self.yerr.data_ptr(), C.byref(logdet)))
        N.check(N.lib.gh_hodlr_dot_solve(self.h, self.y.data_ptr(), C.byref(q)))
        return -0.5 * (self.n * np.log(2 * np.pi) + logdet.value) - 0.5 * q.value

    def ranks(self):
        buf = (C.c_int32 * 65536)()
        cnt = C.c_int32(0)
        self.N.check(self.N.lib.gh_hodlr_ranks(self.h, buf, 65536, C.byref(cnt)))
        return list(buf[:cnt.value])

    def close(self):
        self.N.lib.gh_hodlr_destroy(self.h)


def hodlr_level_ranks(ranks):
    """per-level maximum rank from the breadth-first rank list (2^l nodes on level l)"""
    out, at, l = [], 0, 0
    while at < len(ranks):
        out.append(int(max(ranks[at:at + (1 << l)])))
        at += 1 << l
        l += 1
    return out


def hodlr_report(n, local_rank, steps=10, warmup=3, cpu_n=32768):
    """BASELINE config C4 for the N = 1 line: time, per-level ranks, footprint roofline, CPU sample."""
    job = HodlrJob(n, local_rank)
    ts_, ll = run_steps(job, steps, warmup)
    sec = float(np.mean(ts_))
    ranks = job.ranks()
    job.close()
    lv = hodlr_level_ranks(ranks)
    rtot = sum(lv)
    foot = 2.0 * 8.0 * n * rtot + 8.0 * n * 128            # U and V of every level + the 128-row leaves
    # bytes the factorisation sweep + the log-likelihood's solve actually STREAM (not the footprint touched once): per level l
    # with rank R and off = sum of the shallower ranks -- the reduce reads U[:, 0:off+R) and V_l, the update reads and writes
    # U[:, 0:off) and reads U_l; leaves applied to all of U (read + written) + the leaf inverses read; the solve reads U, V and
    # the leaf inverses once each.  (Fusing update and reduce -- 2/3 of these bytes -- was built in round 4 and is NOT faster:
    # scripts/dev/arms/hodlr_fused_sweep_r04.patch.)
    offs = np.concatenate([[0], np.cumsum(lv)[:-1]]) if lv else np.zeros(0)
    active = [l for l in range(len(lv)) if lv[l] > 0]
    streamed = 2.0 * 8.0 * n * rtot + 8.0 * n * 128        # leaf apply
    for l in active:
        streamed += 8.0 * n * (2.0 * offs[l] + 2.0 * lv[l])
    if active:
        streamed += 8.0 * n * (offs[active[-1]] + lv[active[-1]])
    streamed_fused = streamed
    streamed += sum(8.0 * n * (offs[l] + lv[l]) for l in active[:-1])
    streamed += 2.0 * 8.0 * n * rtot + 8.0 * n * 128       # the solve
    streamed_fused += 2.0 * 8.0 * n * rtot + 8.0 * n * 128
    measured = None
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_C4_N%d.json" % n)))
    if hits:
        measured = json.load(open(hits[-1]))["bytes_per_step"]
    out = {"workload": "N=%d 1-D ExpSquared, HODLRSolver(tol=1e-10, min_size=100, seed=42): compute()+log_likelihood()" % n,
           "seconds_per_step": sec, "per_step_s": spread(ts_), "steps": steps, "log_likelihood": ll, "rank_per_level": lv, "rank_total": rtot,
           "roofline": {"kernel": "whole HODLR compute()+log_likelihood() (ACA, leaf / core factorisation, Woodbury sweeps)",
                        "bound": "hbm", "achieved": foot / sec * 1e-9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": foot / sec * 1e-9 / PEAK_HBM_GBS, "traffic": measured if measured is not None else streamed,
                        "traffic_source": os.path.relpath(hits[-1], ROOT) if hits else "model (streamed_bytes)",
                        "traffic_note": "traffic: bytes per step leaving the L2s, from the committed --pmc passes when present; "
                                        "streamed_bytes: what the sweep + solve stream by construction (model from the measured ranks; the ACA "
                                        "phase evaluates kernel entries, it streams nothing); %.2f GB with update and reduce fused "
                                        "(built in round 4, not faster: scripts/dev/arms)" % (streamed_fused * 1e-9),
                        "streamed_bytes": streamed, "achieved_streamed": streamed / sec * 1e-9,
                        "algorithmic_bytes": foot, "note": "footprint 2*8*N*Rtot + 8*N*128 touched once; the step is "
                        "latency-bound (a chain of ~%d dependent launches), not bandwidth-bound" % (12 * len(lv))}}
    try:
        from oracle import ref_loader
        import george_amd.kernels as K
        H = ref_loader.load_hodlr()
        if H is not None and cpu_n > 0:
            x, yerr, y = make_inputs(cpu_n)
            kernel = float(np.var(y)) * K.ExpSquaredKernel(1.0)
            t0 = time.perf_counter()
            h = H()
            h.compute(kernel, x[:, None], np.sqrt(yerr ** 2 + 1.25e-12), 100, 1e-10, 42)
            llc = -0.5 * (cpu_n * np.log(2 * np.pi) + h.log_determinant) - 0.5 * h.dot_solve(y)
            dt = time.perf_counter() - t0
            gj = HodlrJob(cpu_n, local_rank)
            llg = gj.step()
            gj.close()
            out["cpu_baseline"] = {"value": dt, "unit": "s", "cores": 1, "kind": "reference",
                                   "sample": "the reference's own hodlr.h (unmodified, compiled against oracle/mini_eigen: "
                                             "plain loops where Eigen vectorises) at N=%d, same tol/min_size/seed" % cpu_n,
                                   "log_likelihood": llc}
            out["parity_cpu_sample"] = {"n": cpu_n, "ll_gpu": llg, "ll_ref": llc, "rel": abs(llg - llc) / abs(llc)}
    except Exception as e:                                   # the checker must not take the bench line down
        out["cpu_baseline_error"] = repr(e)
    # parity at the STATED size: the reference's own hodlr.h at N = 262144 (59 s on one core in the build
    # container; tests/golden/large.json[C4], oracle/gen_golden_large.py)
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "large.json"))).get("C4")
        if g is not None and g["n"] == n:
            out["parity"] = {"n": n, "ll_gpu": ll, "ll_ref": g["loglike"], "rel": abs(ll - g["loglike"]) / abs(g["loglike"]),
                             "ref": "tests/golden/large.json[C4]: reference hodlr.h (unmodified) in the build container, "
                                    "%.0f s on one core" % g.get("seconds_factor", float("nan")),
                             "rank_per_level_ref": g.get("rank_per_level")}
            if "cpu_baseline" in out:
                out["cpu_baseline"]["full_size_seconds_build_container"] = g.get("seconds_factor")
        elif "parity_cpu_sample" in out:
            out["parity"] = out["parity_cpu_sample"]
    except Exception as e:
        out["parity_error"] = repr(e)
    return out


def c5_report(local_rank, n=32768, m=4096):
    """BASELINE config C5: 3-D Matern52 + Constant, compute+loglike / predict mean+var /
    grad_log_likelihood through the public GP facade on NumPy inputs."""
    from george_amd import GP, kernels
    rng = np.random.RandomState(1234)
    x = rng.uniform(0, 1, (n, 3))
    x = x[np.argsort(x[:, 0])]
    y = np.sin(x.sum(axis=1))
    t = rng.uniform(0, 1, (m, 3))
    kernel = kernels.Matern52Kernel(0.5, ndim=3) + kernels.ConstantKernel(log_constant=np.log(0.1 / 3), ndim=3)
    gp = GP(kernel, device=local_rank)
    res = {}
    for rep in range(2):                                     # second pass is the measurement (buffers exist)
        t0 = time.perf_counter(); gp.compute(x, 0.1); ll = gp.log_likelihood(y); res["compute_loglike_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); mu, var = gp.predict(y, t, return_var=True); res["predict_var_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); g = gp.grad_log_likelihood(y); res["grad_s"] = time.perf_counter() - t0
    # the optimiser objective (gp.py:470-480) as ONE fused device call per iterate: two warm-up iterates
    # (work arrays exist, the handle comes back from the pool untrimmed), then the best of three at
    # DIFFERENT parameter vectors (nothing can be served from a cache)
    p = gp.get_parameter_vector()
    gp.grad_nll(p, y)                                        # (switches nll to the eager-gradient form)
    fused = []
    for it in range(5):
        t0 = time.perf_counter(); v, gg = gp.nll_and_grad(p + 1e-3 * (it + 1), y); fused.append(time.perf_counter() - t0)
    res["fused_nll_and_grad_s"] = min(fused[2:])
    res["fused_nll_and_grad_all_s"] = fused
    res["fused_not_slower_than_separate_calls"] = bool(
        res["fused_nll_and_grad_s"] <= 1.05 * (res["compute_loglike_s"] + res["grad_s"]))
    f1, f3 = n ** 3 / 3.0 + 2.0 * n ** 2, 2.0 * n ** 3 / 3.0
    f2_run = float(n) * n * m + 2.0 * n * m                  # what gh_chol_predict executes: V = L^-1 K*^T (N^2 M), colsum(V^2), V^T z
    f2_ref = 2.0 * n ** 2 * m                                # the reference's algorithm: cho_solve(K*^T) = two sweeps (gp.py:536)
    res.update({
        "workload": "N=%d 3-D sorted-by-x0 uniform, Matern52(0.5)+Constant(0.1), yerr=0.1: compute()+log_likelihood(), "
                    "predict(mean+var, M=%d), grad_log_likelihood(); NumPy in / NumPy out" % (n, m),
        "N": n, "M": m, "log_likelihood": float(ll), "grad": [float(v_) for v_ in g],
        "tflops": {"compute_loglike (N^3/3+2N^2)": f1 / res["compute_loglike_s"] * 1e-12,
                   "predict (N^2 M + 2 N M: the forward sweep that is run)": f2_run / res["predict_var_s"] * 1e-12,
                   "grad (2 N^3/3: K^-1 = L^-T L^-1)": f3 / res["grad_s"] * 1e-12,
                   "fused nll+grad (N^3)": (f1 + f3) / res["fused_nll_and_grad_s"] * 1e-12},
        "predict_effective_vs_reference_algorithm": {
            "flops_model": "2 N^2 M (the reference solves both triangular systems, gp.py:536)",
            "tflops_equivalent": f2_ref / res["predict_var_s"] * 1e-12,
            "note": "not a rate of this GPU: half of the reference's arithmetic is not done (var = k** - ||L^-1 k*||^2)"},
        "roofline": {"kernel": "gemm_f64_mfma_dma family (factor + triangular inverse + K^-1 product)", "bound": "mfma",
                     "achieved": (f1 + f3) / res["fused_nll_and_grad_s"] * 1e-12, "peak": PEAK_FP64_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": (f1 + f3) / res["fused_nll_and_grad_s"] * 1e-12 / PEAK_FP64_MFMA_TFLOPS,
                     "traffic": None, "scope": "whole fused objective call, host clock"}})
    over = [k for k, v_ in res["tflops"].items() if v_ > PEAK_FP64_MFMA_TFLOPS]
    if over:
        res["flop_model_error"] = "rates above the fp64 matrix peak: %r" % over
    del gp
    return res


def f1_kernel(kernels):
    """docs/tutorials/hyper.rst:91-95 (the survey's f1 workload): k1 + k2 * ExpSine2 + k3 + k4, 11 kernel parameters"""
    k1 = 66.0 ** 2 * kernels.ExpSquaredKernel(metric=67.0 ** 2)
    k2 = 2.4 ** 2 * kernels.ExpSquaredKernel(90.0 ** 2) * kernels.ExpSine2Kernel(gamma=2.0 / 1.3 ** 2, log_period=0.0)
    k3 = 0.66 ** 2 * kernels.RationalQuadraticKernel(log_alpha=np.log(0.78), metric=1.2 ** 2)
    k4 = 0.18 ** 2 * kernels.ExpSquaredKernel(1.6 ** 2)
    return k1 + k2 + k3 + k4


def f1_data(n, seed=7):
    """n 'monthly CO2' samples over 1958 .. 2010 shaped like the tutorial's Mauna Loa series (the data set itself is a download)"""
    rng = np.random.RandomState(seed)
    t = np.sort(1958.0 + 52.0 * (np.arange(n) + rng.uniform(0.0, 1.0, n)) / n)
    y = 315.0 + 0.9 * (t - 1958.0) + 0.012 * (t - 1958.0) ** 2 + 3.0 * np.sin(2 * np.pi * t) + 0.8 * np.sin(4 * np.pi * t + 0.4)
    return t, y + 0.25 * rng.randn(n)


def f1_report(local_rank, sizes=(2048, 8192, 16384), cpu_n=2048):
    """SURVEY 8(f).1 on the kernel it exists for (hyper.rst:91-152): the 13-parameter model (11 kernel parameters + white noise +
    mean) -- compute()+log_likelihood(), grad_log_likelihood() and the fused nll_and_grad() per optimiser iterate, the kernel-matrix
    build alone (HIP events inside compute()), and the reference CPU path (its C++ evaluator + LAPACK) at the smallest size."""
    from george_amd import GP, kernels, BasicSolver
    res = {"kernel": "66^2 ES(67^2) + 2.4^2 ES(90^2) ExpSine2(2/1.3^2, 0) + 0.66^2 RQ(log 0.78, 1.2^2) + 0.18^2 ES(1.6^2); "
                     "mean and white noise fitted: 13 parameters (docs/tutorials/hyper.rst:91-104)", "sizes": {}}
    for n in sizes:
        t, y = f1_data(n)
        gp = GP(f1_kernel(kernels), mean=float(np.mean(y)), fit_mean=True, white_noise=np.log(0.19 ** 2), fit_white_noise=True,
                solver=BasicSolver, device=local_rank, profile=True)
        r = {}
        for rep in range(3):                                 # third pass is the measurement
            t0 = time.perf_counter(); gp.compute(t); ll = gp.log_likelihood(y); r["compute_loglike_ms"] = (time.perf_counter() - t0) * 1e3
            r["build_ms"] = float(gp.solver.profile()["ms_build"])
            t0 = time.perf_counter(); g = gp.grad_log_likelihood(y); r["grad_ms"] = (time.perf_counter() - t0) * 1e3
        p = gp.get_parameter_vector()
        gp.grad_nll(p, y)
        fused = []
        for it in range(5):
            t0 = time.perf_counter(); v, gg = gp.nll_and_grad(p + 1e-4 * (it + 1), y); fused.append((time.perf_counter() - t0) * 1e3)
        r["fused_nll_and_grad_ms"] = min(fused[2:])
        npd = -(-n // 128) * 128
        tiles = (npd // 128) * (npd // 128 + 1) // 2
        bytes_alg = tiles * 128 * 128 * 8 + 16 * n
        r["build_GBs"] = bytes_alg / (r["build_ms"] * 1e-3) * 1e-9
        r["build_frac_of_hbm"] = r["build_GBs"] / PEAK_HBM_GBS
        r["build_elements_per_s"] = tiles * 128 * 128 / (r["build_ms"] * 1e-3)
        r["log_likelihood"] = float(ll)
        r["grad"] = [float(v_) for v_ in g]
        res["sizes"]["N%d" % n] = r
        del gp
    if cpu_n:
        # the reference's CPU path for the same iterate at the smallest size: its C++ evaluator (value + 11-parameter gradient) + LAPACK
        from oracle import solver_np
        t, y = f1_data(cpu_n)
        kernel = f1_kernel(kernels)
        mean, wn = float(np.mean(y)), np.log(0.19 ** 2)
        d = solver_np.DenseOracle(kernel)
        t0 = time.perf_counter()
        ll_ref = solver_np.gp_log_likelihood(d, t[:, None], 0.0, y, mean=mean, white_noise=wn)
        t_ll = time.perf_counter() - t0
        t0 = time.perf_counter()
        kg, A = solver_np.gp_grad_log_likelihood(d, kernel, t[:, None], y, mean=mean)
        t_g = time.perf_counter() - t0
        alpha = d.apply_inverse(y - mean)
        gref = np.concatenate([[alpha.sum()], [0.5 * np.exp(wn) * np.trace(A)], kg])       # gp.py:443-461 for a constant mean / white noise
        mine = res["sizes"].get("N%d" % cpu_n)
        res["cpu_reference"] = {"n": cpu_n, "kind": solver_np.evaluator_kind(), "compute_loglike_s": t_ll, "grad_s": t_g,
                                "log_likelihood": float(ll_ref)}
        if mine:
            res["cpu_reference"]["rel_ll"] = abs(mine["log_likelihood"] - ll_ref) / abs(ll_ref)
            res["cpu_reference"]["grad_rel_max"] = float(np.max(np.abs(np.asarray(mine["grad"]) - gref) / np.maximum(np.abs(gref), 1e-3 * np.abs(gref).max())))
            res["cpu_reference"]["gpu_over_cpu_iterate"] = (t_ll + t_g) / (mine["fused_nll_and_grad_ms"] * 1e-3)
    return res


def mgpu_abi_report(local_rank, n=32768):
    """The sharded solver behind the C ABI (gh_mgpu_*: host threads + RCCL) as a world of ONE on the leased GPU:
    communicator creation, the all-reduce self-check and the block-cyclic driver end to end, against the
    single-GPU solver on the same inputs.  (More than one physical device is the driver's 8-GPU tier.)"""
    from george_amd import BasicSolver, MultiGPUSolver
    x, yerr, y = make_inputs(n)
    kernel = make_kernel("expsquared", np.var(y))
    X, sig = x[:, None], np.sqrt(yerr ** 2 + 1.25e-12)
    d = BasicSolver(kernel, device=local_rank)
    d.compute(X, sig)
    ll0 = -0.5 * (n * np.log(2 * np.pi) + d.log_determinant) - 0.5 * d.dot_solve(y)
    s = MultiGPUSolver(kernel, devices=[local_rank], transport="rccl")
    s.compute(X, sig)                                        # warm-up (buffers, communicator)
    t0 = time.perf_counter()
    s.compute(X, sig)
    q = s.dot_solve(y)
    sec = time.perf_counter() - t0
    ll = -0.5 * (n * np.log(2 * np.pi) + s.log_determinant) - 0.5 * q
    pr, pc, nb = s.grid_shape()
    del s, d
    return {"workload": "N=%d 1-D ExpSquared through gh_mgpu_create/compute/dot_solve, n_dev=1, transport RCCL" % n,
            "seconds_per_step": sec, "value_tflops": flops_alg(n) / sec * 1e-12, "grid": "%dx%d" % (pr, pc), "nb": nb,
            "log_likelihood": ll, "parity": {"n": n, "ll_gpu": ll, "ll_ref": ll0, "rel": abs(ll - ll0) / abs(ll0),
                                             "ref": "single-GPU gh_chol_* on the same inputs (itself reference-pinned)"}}


def multi_device_probe(timeout_s=240):
    """scripts/abi_multi_device_probe.py in a child process with a time limit: the two several-devices-in-one-process
    forms of the C ABI (gh_mgpu_*, gh_hodlr_mgpu_*) on every device this box shows -- real peer traffic when there is
    more than one MI355X, the HODLR split on one GPU listed twice otherwise.  Never part of the timed region or of the
    line's parity verdict: a failure here is reported in place, the headline stands."""
    import subprocess
    if os.environ.get("GEORGE_AMD_BENCH_NO_MULTI_PROBE"):
        return {"skipped": "GEORGE_AMD_BENCH_NO_MULTI_PROBE"}
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "abi_multi_device_probe.py")]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "no answer within %d s" % timeout_s}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "exit %d: %s" % (r.returncode, r.stderr.strip()[-400:])}
    return json.loads(lines[-1])


def abi_form_report(args, world):
    """N > 1, rank 0, after the process group is gone: the OTHER multi-GPU form -- gh_mgpu_* behind the C ABI, ONE
    process driving all `world` devices with host threads and grouped RCCL send/recv (george_amd.MultiGPUSolver) --
    on the same workload, in a child process with a hard time limit, so that one driver run returns both forms.
    Never part of `value`; a failure or a time-out is reported in place."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "abi_multi_device_probe.py"), "--dense-n", str(args.n),
           "--dense-kernel", args.kernel, "--devices", str(world), "--no-hodlr", "--no-single", "--reps", "2"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.abi_timeout, text=True)
    except subprocess.TimeoutExpired:
        return {"error": "no answer within %d s" % args.abi_timeout}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "exit %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
    d = json.loads(lines[-1]).get("dense_rccl", {})
    if "error" in d:
        return {"error": d["error"][:300]}
    gold = golden_ll(args.n, KERNEL_NAMES[args.kernel])
    res = {"what": "gh_mgpu_create/compute/dot_solve over RCCL, one process, %d devices" % world, "seconds_per_step": d["sharded_s"],
           "value_tflops": flops_alg(args.n) / d["sharded_s"] * 1e-12, "grid": d["grid"], "nb": d["nb"],
           "log_likelihood": d["ll_sharded"], "wall_s": time.perf_counter() - t0}
    if gold is not None:
        res["rel"] = abs(d["ll_sharded"] - gold[0]) / abs(gold[0])
    return res


def public_api_report(n, local_rank, steps=2):
    """The headline work through the public facade: NumPy x, yerr, y -> GP.compute -> log_likelihood,
    host->device of the inputs included (SURVEY.md 8d's statement of the metric)."""
    import torch
    from george_amd import GP, kernels
    x, yerr, y = make_inputs(n)
    gp = GP(float(np.var(y)) * kernels.ExpSquaredKernel(1.0), device=local_rank)
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)                                # warm-up (buffers, streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gp.compute(x, yerr)
        ll = gp.log_likelihood(y)
    sec = (time.perf_counter() - t0) / steps
    del gp
    return {"seconds_per_step": sec, "value_tflops": flops_alg(n) / sec * 1e-12, "steps": steps, "log_likelihood": ll,
            "note": "GP.compute(x, yerr); GP.log_likelihood(y) on NumPy arrays: 3*8*N bytes over PCIe + the Python facade"}


# the ONLY numbers the reference publishes for this path -- a log-log plot
# (docs/tutorials/scaling.rst:156-176,224, scaling_files/scaling_16_0.png), read off by eye (+-20 %), unstated CPU of ~2018
REF_CURVE_SECONDS = {"BasicSolver": {1000: 2.0e-2, 5000: 1.2, 10000: 7.0},
                     "HODLRSolver(default tol=0.1, seed=42)": {1000: 2.5e-3, 10000: 4.5e-2, 50000: 0.4}}


def curve_report(local_rank):
    """The reference's own benchmark loop (docs/tutorials/scaling.rst:56-59,67,84,156-176) at the sizes BASELINE.md
    quotes from its plot: best-of-K ``gp.compute(x[:n], yerr[:n]); gp.log_likelihood(y[:n])`` on the first n of the
    50000 sorted points, NumPy in, through the GP facade, for BasicSolver and for HODLRSolver at its DEFAULT
    tol = 0.1 (seed = 42).  The HODLR log-likelihood is checked against the dense one with the tutorial's own
    criterion at that tolerance (tests/test_tutorial.py:39-43: allclose)."""
    from george_amd import GP, kernels, HODLRSolver
    x, yerr, y = make_inputs(50000)
    kernel = float(np.var(y)) * kernels.ExpSquaredKernel(1.0)
    gps = {"BasicSolver": GP(kernel, device=local_rank),
           "HODLRSolver(default tol=0.1, seed=42)": GP(kernel, solver=HODLRSolver, seed=42, device=local_rank)}
    rows, dense_ll = [], {}
    for name in ("BasicSolver", "HODLRSolver(default tol=0.1, seed=42)"):
        gp = gps[name]
        for n, ref_s in sorted(REF_CURVE_SECONDS[name].items()):
            best, ll = np.inf, None
            for it in range(max(3, min(20, 100000 // n)) + 1):
                t0 = time.perf_counter()
                gp.compute(x[:n], yerr[:n])
                ll = gp.log_likelihood(y[:n])
                dt = time.perf_counter() - t0
                if it > 0:
                    best = min(best, dt)
            row = {"solver": name, "n": n, "seconds": best, "reference_plot_seconds": ref_s, "speedup_vs_plot": ref_s / best,
                   "log_likelihood": float(ll)}
            if name == "BasicSolver":
                dense_ll[n] = ll
            else:
                if n not in dense_ll:                        # N = 50000: the dense answer from the device solver
                    g0 = gps["BasicSolver"]
                    g0.compute(x[:n], yerr[:n])
                    dense_ll[n] = g0.log_likelihood(y[:n])
                row["rel_vs_dense"] = abs(ll - dense_ll[n]) / abs(dense_ll[n])
            rows.append(row)
    return {"what": "compute(x[:n], yerr[:n]) + log_likelihood(y[:n]), best of K, NumPy in (scaling.rst:156-176); "
                    "reference_plot_seconds = BASELINE.md section 1, read off the reference's plot by eye (+-20 %), unstated CPU",
            "rows": rows}


def _r(v, nd=4):
    """round floats for the printed line (the detail file keeps full precision)"""
    if isinstance(v, float):
        return float("%.*g" % (nd + 2, v))
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def compact_line(out):
    """The ONE printed line: the contract's keys + every secondary config in a few numbers each, < 6 kB, so that the
    driver's stored stdout tail holds all of it.  The full record (texts, per-phase splits, every parity operand)
    goes to the detail file named in the line."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "log_likelihood", "frac_of_fp64_mfma_peak"]
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "N", "kernel", "solver", "parallelism", "flops_model", "grid", "nb", "panel_widths") if k in cfg}
    if "value_public_api" in out:
        line["value_public_api"] = out["value_public_api"]
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: rf[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                               "traffic_algorithmic_bytes", "launches", "avg_launch_ms",
                                               "algorithmic_flops_per_launch", "peak_measured", "frac_of_measured",
                                               "scope") if k in rf}
        sa = rf.get("standalone")
        if sa and "K1024" in sa:
            line["roofline"]["standalone_M%d" % sa["M"]] = {k: [_r(sa[k]["tflops"]), _r(sa[k]["frac"])] for k in ("K1024", "K2048")}
        line["roofline"]["kernel"] = line["roofline"]["kernel"][:96]
        w = rf.get("with_overlapped_block_column_launches")
        if w:
            line["roofline"]["with_overlapped_block_column_launches"] = {k: w[k] for k in ("achieved", "frac", "busy_ms_per_step")}
    kb = out.get("roofline_kernel_build")
    if kb:
        line["roofline_kernel_build"] = {"kernel": "kmat_interior_kernel", "bound": "hbm", "achieved": kb["achieved"], "peak": kb["peak"],
                                         "unit": kb["unit"], "frac": kb["frac"], "ms": kb["ms"], "traffic": kb.get("traffic")}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "n", "seconds") if k in cb}
        line["cpu_baseline"]["sample"] = cb.get("sample", "")[:150]
        if cb.get("second_point"):
            sp = cb["second_point"]
            line["cpu_baseline"]["second_point"] = {k: sp[k] for k in ("n", "seconds", "value", "gpu_seconds", "gpu_over_cpu", "rel") if k in sp}
        if cb.get("full_size_run"):
            line["cpu_baseline"]["full_size_other_box_s"] = cb["full_size_run"].get("seconds_total")
    if "phases_ms" in out:
        line["phases_ms"] = out["phases_ms"]
    also = {}
    a = cfg.get("also_configs1_N16384")
    if a:
        also["configs1_N16384"] = {"ms": a["seconds_per_step"] * 1e3, "tflops": a["value_tflops"], "frac": a["frac_of_fp64_mfma_peak"]}
        if "per_step_s" in a:
            also["configs1_N16384"]["ms_min_med_max"] = [a["per_step_s"][k] * 1e3 for k in ("min", "median", "max")]
    a = cfg.get("also_C4")
    if a and "seconds_per_step" in a:
        also["C4_hodlr_N262144"] = {"ms": a["seconds_per_step"] * 1e3,
                                    "ms_min_med_max": [a["per_step_s"][k] * 1e3 for k in ("min", "median", "max")] if "per_step_s" in a else None,
                                    "rank_per_level": a.get("rank_per_level"),
                                    "roofline": {k: a["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                                   "algorithmic_bytes", "streamed_bytes", "achieved_streamed")
                                                 if k in a["roofline"]},
                                    "cpu_ref_s": (a.get("cpu_baseline") or {}).get("value"),
                                    "cpu_ref_full_size_s": (a.get("cpu_baseline") or {}).get("full_size_seconds_build_container")}
    a = cfg.get("also_C5")
    if a and "compute_loglike_s" in a:
        also["C5_N32768_3d"] = {"compute_loglike_s": a["compute_loglike_s"], "predict_var_s": a["predict_var_s"], "grad_s": a["grad_s"],
                                "fused_nll_and_grad_s": a["fused_nll_and_grad_s"], "fused_frac": a["roofline"]["frac"]}
    a = cfg.get("also_f1")
    if a and "sizes" in a:
        also["f1_hyper_kernel_13p"] = {k: {q: _r(v[q]) for q in ("compute_loglike_ms", "build_ms", "build_frac_of_hbm", "grad_ms", "fused_nll_and_grad_ms")}
                                       for k, v in a["sizes"].items()}
        cr = a.get("cpu_reference")
        if cr:
            also["f1_hyper_kernel_13p"]["cpu_ref_N%d_s" % cr["n"]] = [_r(cr["compute_loglike_s"]), _r(cr["grad_s"])]
            if "gpu_over_cpu_iterate" in cr:
                also["f1_hyper_kernel_13p"]["gpu_over_cpu_iterate"] = _r(cr["gpu_over_cpu_iterate"])
    elif a:
        also["f1_hyper_kernel_13p"] = a
    a = cfg.get("also_abi_multi_gpu_world_of_one")
    if a:
        also["abi_mgpu_world1_N32768"] = {"s": a.get("seconds_per_step"), "tflops": a.get("value_tflops")} if "error" not in a else {"error": a["error"][:120]}
    a = cfg.get("also_abi_multi_device")
    if a:
        d = {"devices": a.get("devices_visible")}
        if "dense_rccl" in a:
            d["dense_rccl"] = {k: a["dense_rccl"].get(k) for k in ("n", "grid", "nb", "single_gpu_s", "sharded_s", "speedup", "rel", "error") if k in a["dense_rccl"]}
        if "hodlr_split" in a:
            d["hodlr_split"] = [{k: c.get(k) for k in ("n", "single_gpu_s", "split_s", "rel", "error") if k in c} for c in a["hodlr_split"].get("cases", [])]
        if "error" in a:
            d["error"] = a["error"][:160]
        also["abi_multi_device"] = d
    a = cfg.get("also_C3_matern32")
    if a:
        also["C3_matern32"] = {"s": a["seconds_per_step"], "tflops": a["value_tflops"]}
    a = cfg.get("also_other_grid")
    if a:
        also["other_grid"] = {k: a[k] for k in ("grid", "seconds_per_step", "value_tflops", "error") if k in a}
    a = cfg.get("also_curve")
    if a:
        also["reference_plot_sizes"] = [[r["solver"].split("(")[0], r["n"], r["seconds"], r["reference_plot_seconds"]] for r in a["rows"]]
    if also:
        line["also"] = also
    par = out.get("parity")
    if isinstance(par, dict):
        line["parity"] = {"bound": par.get("bound"), "ok": par.get("ok"),
                          "rel": {k: v["rel"] for k, v in par.items() if isinstance(v, dict) and "rel" in v}}
    for k in ("rccl_ranks_seen", "timeline_rank0_ms", "abi_form", "links"):
        if k in out:
            line[k] = out[k]
    if "rccl" in out:
        line["rccl"] = {k: out["rccl"][k] for k in ("backend", "ranks_seen", "distinct_devices")}
    if "detail" in out:
        line["detail"] = out["detail"]
    return _r(line)


def golden_ll(n, kernel_name="ExpSquared"):
    """reference log-likelihood committed under tests/golden/large.json for the headline inputs, or None"""
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "large.json")))
        key = {(16384, "ExpSquared"): "C2", (65536, "ExpSquared"): "NS", (65536, "Matern32"): "C3",
               (20480, "Matern32"): "M32_20k"}.get((n, kernel_name))
        return (g[key]["loglike"], key) if key in g else None
    except Exception:
        return None


def hodlr_main(args, local_rank):
    """python bench.py --workload hodlr [--n 262144]: secondary report for config C4 (one GPU)."""
    n = args.n if args.n != 65536 else 262144
    job = HodlrJob(n, local_rank)
    elapsed, ll = run_timed(job, args.steps, args.warmup, lambda: None)
    ranks = job.ranks()
    out = {"metric": "gp_hodlr_compute_plus_log_likelihood_seconds", "value": elapsed / args.steps, "unit": "s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": False, "scaling": "replicas", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "N=%d 1-D ExpSquared, HODLRSolver(tol=1e-10, min_size=100, seed=42): "
                                  "compute()+log_likelihood()" % n, "N": n},
           "log_likelihood": ll, "max_rank": max(ranks) if ranks else 0, "n_internal_nodes": len(ranks)}
    job.close()
    if not args.no_cpu:
        from oracle import hodlr_np, solver_np
        import george_amd.kernels as K
        nc = min(n, 32768)
        x, yerr, y = make_inputs(nc)
        kernel = float(np.var(y)) * K.ExpSquaredKernel(1.0)
        t0 = time.perf_counter()
        llc = solver_np.gp_log_likelihood(hodlr_np.HODLROracle(kernel, tol=1e-10), x[:, None], yerr, y)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": dt, "unit": "s", "cores": 1, "kind": "port",
                               "sample": "NumPy restatement of hodlr.h at N=%d (the reference's Eigen extension "
                                         "cannot be built here); loglike=%.10g" % (nc, llc)}
    emit(out)


def emit(out, detail_path=None):
    """Write the full record to the detail file, print THE (compact) line.  RCCL writes a version banner to C stdout at communicator creation ("RCCL version : ...",
    five lines); through a pipe that stdio buffer is flushed at exit, i.e. AFTER Python's line.  Flush it first so
    that the JSON line is the last thing on stdout."""
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            out["detail"] = os.path.relpath(detail_path, ROOT)
        except OSError as e:
            out["detail"] = "not written: %r" % (e,)
    line = json.dumps(compact_line(out))
    if len(line) > 6000:                                     # the driver keeps the last 8 kB of stdout
        sys.stderr.write("bench.py: the printed line is %d bytes (> 6000)\n" % len(line))
    print(line)
    sys.stdout.flush()


def relaunch(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this very script under
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and hand its exit status back.  Fails
    loudly -- before launching anything -- if the box has fewer than N GPUs."""
    import socket
    import subprocess
    if args.tile_ops != "numpy" and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d visible: refusing to run (an N > 1 line is only ever "
                             "printed by N ranks on N GPUs)\n" % (args.gpus, have))
            return 2
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    fwd = []
    for a in sys.argv[1:]:                                       # torch.distributed.run's parser trips over "--n"
        fwd.append("--size" if a == "--n" else ("--size=" + a[4:] if a.startswith("--n=") else a))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + fwd
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def run_timed(job, steps, warmup, barrier):
    import torch
    ll = None
    for _ in range(warmup):
        ll = job.step()
    if hasattr(job, "reset_profile"):
        job.reset_profile()
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)   # (no GPU: --tile-ops numpy self-test)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ll = job.step()
    sync()
    barrier()
    return time.perf_counter() - t0, ll


def run_steps(job, steps, warmup):
    """The secondary legs of the line: every step timed on its own (synchronised on both sides), so that the line can carry
    min / median / max and a slow box or a first-touch step shows as such.  -> (list of seconds, last value)"""
    import torch
    v = None
    for _ in range(warmup):
        v = job.step()
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = job.step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return ts, v


def spread(ts):
    return {"min": float(np.min(ts)), "median": float(np.median(ts)), "max": float(np.max(ts)), "n": len(ts)}


def link_probe(dist, device, world, rank, panel_bytes):
    """Before the timed steps of an N > 1 run: what the links give, so that the line itself checks the assumptions of
    profiles/r04/scale_model.md -- (i) one-way rate of ONE busy link (rank 0 -> each peer in turn), (ii) per-link rate with all
    N - 1 links of every rank busy (each rank sends to every other rank at once), (iii) bus bandwidth of the bulk all-gather at
    the first step's panel size.  Works on any backend (the launcher self-test runs it over gloo)."""
    import torch
    nbytes = int(min(64 << 20, max(1 << 20, panel_bytes)))
    n = nbytes // 8
    sync = (lambda: torch.cuda.synchronize()) if device != "cpu" else (lambda: None)
    buf = torch.ones(n, dtype=torch.float64, device=device)
    rx = [torch.empty(n, dtype=torch.float64, device=device) for _ in range(world)]
    out = {"bytes": nbytes}

    def timed(fn, reps=3):
        fn()
        sync(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync(); dist.barrier()
        return (time.perf_counter() - t0) / reps

    one = []
    for peer in range(1, world):
        def f(peer=peer):
            if rank == 0:
                dist.send(buf, dst=peer)
            elif rank == peer:
                dist.recv(rx[0], src=0)
        one.append(nbytes / timed(f) * 1e-9)
    out["one_link_GBs"] = {"min": min(one), "median": float(np.median(one)), "max": max(one)}

    def all_links():
        ops = []
        for d in range(1, world):
            ops.append(dist.P2POp(dist.isend, buf, (rank + d) % world))
            ops.append(dist.P2POp(dist.irecv, rx[d], (rank - d) % world))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    t = timed(all_links)
    out["all_links_busy_GBs_per_link"] = nbytes / t * 1e-9
    gat = torch.empty(n * world, dtype=torch.float64, device=device)
    t = timed(lambda: dist.all_gather_into_tensor(gat, buf))
    out["allgather_busbw_GBs"] = nbytes * (world - 1) / t * 1e-9
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--size", dest="n", type=int, default=65536,
                    help="N (north-star target config: 65536, 1-D ExpSquared); spell it --size under "
                         "torch.distributed.run, whose own parser trips over the abbreviation --n")
    ap.add_argument("--nb", type=int, default=0, help="outer panel width (0 = library default)")
    ap.add_argument("--cpu-n", type=int, default=20480, help="size of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-n2", type=int, default=32768, help="size of the second, larger same-box CPU point (0 = skip; ~40 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--grid", default="", help="N > 1: process grid PrxPc (default: N x 1, whole tile rows per rank in snake order; "
                    "'square' = 1x2 / 2x2 / 2x4)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the FULL record goes (the printed line is its < 6 kB digest); '' = nowhere")
    ap.add_argument("--abi-timeout", type=int, default=420,
                    help="N > 1: time limit in seconds for the second multi-GPU form (gh_mgpu_* behind the C ABI, one process) "
                         "that rank 0 runs in a child process after the ranks have left; 0 = skip")
    ap.add_argument("--no-lookahead", action="store_true", help="single-stream factorisation (profiling aid)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary N=16384 (configs[1]) measurement")
    ap.add_argument("--workload", default="dense", choices=["dense", "hodlr"],
                    help="dense = the headline metric; hodlr = secondary report for BASELINE config C4")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl == RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: every rank uses device 0 (with --backend gloo: exercises the N>1 path on a 1-GPU box)")
    ap.add_argument("--dump-intervals", default="", help="write the per-launch (start, end, flops) list behind the roofline "
                    "object to this JSON file (committed under profiles/: the union-based fraction is reproducible from it)")
    ap.add_argument("--kernel", default="expsquared", choices=sorted(KERNEL_NAMES),
                    help="expsquared = north-star target / configs[1]; matern32 = configs[2] (C3, checked against "
                         "tests/golden/large.json[C3] at N=65536)")
    ap.add_argument("--tile-ops", default="hip", choices=["hip", "numpy"],
                    help="numpy: launcher self-test of the N > 1 path on a box without a GPU (tests/np_tile_ops.py; "
                         "the line says so in 'data' and is not a measurement)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))                                 # N ranks under torch.distributed.run, same arguments

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch exactly one rank per GPU asked for "
                         "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d), or plain "
                         "'python bench.py --gpus %d', which launches them itself\n"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
        sys.exit(2)
    selftest = args.tile_ops == "numpy"
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if not selftest:
        if local_rank >= torch.cuda.device_count():
            sys.stderr.write("bench.py: rank %d wants GPU %d, but this box has %d (use --share-gpu --backend gloo to "
                             "debug the N > 1 path on one GPU)\n" % (rank, local_rank, torch.cuda.device_count()))
            sys.exit(2)
        torch.cuda.set_device(local_rank)
    if args.workload == "hodlr":
        if rank == 0:
            hodlr_main(args, local_rank)
        return

    rccl = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            from george_amd.distributed import nccl_options
            try:
                dist.init_process_group("nccl", pg_options=nccl_options())
            except TypeError:
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        # who is in the group: one (rank, device index, device name, PCI bus id) per rank, gathered
        me = {"rank": rank, "device": local_rank}
        if not selftest:
            pr_ = torch.cuda.get_device_properties(local_rank)
            me.update({"name": pr_.name, "pci_bus_id": getattr(pr_, "pci_bus_id", None), "uuid": str(getattr(pr_, "uuid", ""))})
        seen = [None] * world
        dist.all_gather_object(seen, me)
        rccl = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "members": seen,
                "distinct_devices": len(set((m.get("uuid") or m.get("pci_bus_id") or m["device"]) for m in seen))}
        if args.backend == "nccl" and not args.share_gpu and rccl["distinct_devices"] != world:
            raise SystemExit("bench.py: %d ranks on %d distinct GPUs" % (world, rccl["distinct_devices"]))
        from george_amd.distributed import DistributedDenseJob
        ops = None
        if selftest:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from np_tile_ops import NumpyTileOps
            amp = float(np.var(make_inputs(args.n)[2]))
            ops = NumpyTileOps(make_kernel(args.kernel, amp))
        job = DistributedDenseJob(args.n, args.nb, local_rank, make_inputs, kernel=make_kernel, kernel_name=args.kernel, ops=ops,
                                  grid=args.grid or None)
        barrier = dist.barrier
    else:
        job = DenseJob(args.n, args.nb, local_rank, profile=True, lookahead=not args.no_lookahead, kernel=args.kernel)
        barrier = lambda: None

    links = None
    if world > 1 and not args.no_extra:
        import torch.distributed as dist
        try:
            links = link_probe(dist, "cpu" if selftest else "cuda", world, rank, args.n * job.nb * 8 // world)
        except Exception as e:                                    # (a backend without all_gather_into_tensor / batched p2p: say so)
            links = {"error": repr(e)[:200]}
    elapsed, ll = run_timed(job, args.steps, args.warmup, barrier)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if selftest else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold the same log-likelihood (replicated scalars)
        lls = torch.tensor([ll, -ll], dtype=torch.float64, device=t.device)
        dist.all_reduce(lls, op=dist.ReduceOp.MAX)
        ll_spread = float(lls[0].item() + lls[1].item())             # max - min over ranks
        upd_prof = job.chol.update_profile()     # this rank's trailing updates over the timed steps (HIP events, main stream)
        try:
            tline = job.chol.timeline()
            tline.pop("per_step", None)
        except Exception as e:
            tline = {"error": repr(e)}
        # configs[2] (C3: N=65536 Matern32, block-cyclic) on the same workspace: one warm-up + one timed step
        c3 = None
        if not args.no_extra and args.kernel != "matern32" and (args.n == 65536 or selftest):
            job.set_kernel("matern32")
            job.step()
            e3, ll3 = run_timed(job, 1, 0, barrier)
            t3 = torch.tensor([e3], dtype=torch.float64, device=t.device)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            c3 = (float(t3.item()), ll3)
        # the OTHER grid on the same ranks: the default is N x 1 (whole tile rows per rank, chosen on the evidence of a replay
        # model, profiles/r04/scale_model.md), north_star words it 2-D block-cyclic -- one warm-up + one timed step of
        # 'square' (1x2 / 2x2 / 2x4), or of N x 1 when --grid asked for something else, so that ONE hardware run adjudicates
        other = None
        if not args.no_extra and world in (2, 4, 8):
            og = "square" if not args.grid else ""
            try:
                job.set_kernel(args.kernel)
                j2 = DistributedDenseJob(args.n, args.nb, local_rank, make_inputs, kernel=make_kernel, kernel_name=args.kernel, ops=ops,
                                         grid=og or None)
                j2.step()
                eo, llo = run_timed(j2, 1, 0, barrier)
                to = torch.tensor([eo], dtype=torch.float64, device=t.device)
                dist.all_reduce(to, op=dist.ReduceOp.MAX)
                other = {"grid": "%dx%d" % j2.chol_grid(), "seconds_per_step": float(to.item()),
                         "value_tflops": flops_alg(args.n) / float(to.item()) * 1e-12, "log_likelihood": llo, "steps": 1}
                del j2
            except Exception as e:
                other = {"error": repr(e)[:200]}

    if world > 1:
        # every rank's RCCL banner (C stdio) out BEFORE rank 0 prints the line, so that the line is last
        import torch.distributed as dist
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        dist.barrier()
    if rank == 0:
        sec = elapsed / args.steps
        value = flops_alg(args.n) / sec * 1e-12
        out = {
            "metric": "gp_compute_plus_log_likelihood_effective_tflops",
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec * 1e3, "seconds_per_step": sec, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not selftest else "launcher-selftest (NumPy tile stand-in on CPU: NOT a measurement)",
            "config": {
                "workload": "N=%d 1-D sorted uniform(0,10) x, var(y)*%sKernel(1.0), yerr=0.1, "
                            "dense Cholesky fp64 (BasicSolver path): compute()+log_likelihood()" % (args.n, KERNEL_NAMES[args.kernel]),
                "N": args.n, "ndim": 1, "kernel": KERNEL_NAMES[args.kernel], "solver": "dense-cholesky",
                "parallelism": "1gpu" if world == 1 else "block-cyclic-%d" % world,
                "flops_model": "N^3/3 + 2 N^2",
            },
            "log_likelihood": ll,
            "frac_of_fp64_mfma_peak": value / (PEAK_FP64_MFMA_TFLOPS * world),
        }
        if world == 1:
            out["config"]["panel_widths"] = ("adaptive: 2048 while > 25600 trailing columns, then 1024 (gh_chol.hip panel_starts)"
                                             if args.nb == 0 else "uniform %d" % args.nb)
        if world > 1:
            out["rccl"] = rccl
            out["rccl_ranks_seen"] = rccl["ranks_seen"]
            out["config"]["grid"] = "%dx%d" % job.chol_grid()
            out["config"]["nb"] = job.nb
            parity = {"ranks_agree": {"spread": ll_spread, "rel": abs(ll_spread) / max(abs(ll), 1e-300)}}
            gold = golden_ll(args.n, KERNEL_NAMES[args.kernel])
            if gold is not None:
                parity["headline"] = {"n": args.n, "ll_gpu": ll, "ll_ref": gold[0], "rel": abs(ll - gold[0]) / abs(gold[0]),
                                      "ref": "tests/golden/large.json[%s]: reference C++ evaluator + LAPACK in the build "
                                             "container (oracle/gen_golden_large.py)" % gold[1]}
            elif not selftest:
                # no committed scalar at this size: the single-GPU solver (itself reference-pinned) on rank 0's GPU
                try:
                    j1 = DenseJob(args.n, 0, local_rank, profile=False, kernel=args.kernel)
                    ll1 = j1.step()
                    j1.close()
                    parity["headline"] = {"n": args.n, "ll_gpu": ll, "ll_ref": ll1, "rel": abs(ll - ll1) / abs(ll1),
                                          "ref": "single-GPU gh_chol_* on rank 0 (no committed reference scalar at this size)"}
                except Exception as e:
                    parity["headline_error"] = repr(e)
            else:
                from oracle import solver_np
                xs, es, ys = make_inputs(args.n)
                llr = solver_np.gp_log_likelihood(solver_np.DenseOracle(make_kernel(args.kernel, np.var(ys))), xs[:, None], es, ys)
                parity["headline"] = {"n": args.n, "ll_gpu": ll, "ll_ref": float(llr), "rel": abs(ll - llr) / abs(llr),
                                      "ref": "oracle/solver_np (self-test)"}
            if c3 is not None:
                g3 = golden_ll(args.n, "Matern32")
                out["config"]["also_C3_matern32"] = {
                    "workload": "N=%d 1-D Matern32, block-cyclic over %d ranks (BASELINE configs[2])" % (args.n, world),
                    "seconds_per_step": c3[0], "value_tflops": flops_alg(args.n) / c3[0] * 1e-12, "steps": 1, "log_likelihood": c3[1]}
                if g3 is not None:
                    parity["C3_matern32"] = {"n": args.n, "ll_gpu": c3[1], "ll_ref": g3[0], "rel": abs(c3[1] - g3[0]) / abs(g3[0]),
                                             "ref": "tests/golden/large.json[%s]" % g3[1]}
            if other is not None:
                out["config"]["also_other_grid"] = other
                if "log_likelihood" in other:
                    parity["other_grid"] = {"n": args.n, "ll_gpu": other["log_likelihood"], "ll_ref": ll,
                                            "rel": abs(other["log_likelihood"] - ll) / abs(ll), "ref": "the default grid, same ranks"}
            if links is not None:
                out["links"] = links
            out["parity"] = parity
            out["parity"]["bound"] = 1e-6
            out["parity"]["ok"] = all(v["rel"] <= 1e-6 for v in parity.values() if isinstance(v, dict))
            ms, fl, calls = upd_prof                           # rank 0's trailing updates, HIP events on its main stream
            if ms > 0:
                ach = fl / (ms * 1e-3) * 1e-12
                out["roofline"] = {
                    "kernel": "gemm_f64_mfma_dma_sp<full> (k-major x k-major; rank 0's trailing tile updates; one event pair per "
                              "update sweep, its tile-column GEMMs fanned over 3 streams)",
                    "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": None, "launches": calls,
                    "avg_launch_ms": ms / max(calls, 1), "algorithmic_flops_per_launch": fl / max(calls, 1),
                    "scope": "per GPU (rank 0)"}
        if world == 1:
            p = job.profile()
            if p.n_trailing > 0 and p.ms_trailing > 0:
                ach_syrk = p.trailing_flops / (p.ms_trailing * 1e-3) * 1e-12
                nl = int(p.n_trailing)
                # THE roofline object: the dominant kernel alone -- algorithmic flops of a wide lower-triangular SYRK
                # launch / that launch's own start-to-end time (HIP events on the stream it is launched on), averaged
                # over the timed steps.  rocprofv3's average duration of the same kernel must agree
                # (profiles/r04/kernel_trace_N65536_*.md).
                out["roofline"] = {
                    "kernel": "gemm_f64_mfma_dma_sp<lower> (k-major x k-major; trailing SYRK A22 -= L21 L21^T, one wide launch per panel)",
                    "bound": "mfma", "achieved": ach_syrk, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach_syrk / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                    "launches": nl, "avg_launch_ms": p.ms_trailing / nl,
                    "algorithmic_flops_per_launch": p.trailing_flops / nl}
                union = getattr(p, "ms_update_union", 0.0)
                if union > 0 and p.update_flops > 0:
                    # With the depth-1 look-ahead the block-column launch U(j, j+1) of the chain stream runs BESIDE the
                    # wide launch and borrows part of the chip from it: flops of all update launches / the time during
                    # which any of them ran (union of their HIP-event intervals) says what the chip delivered meanwhile.
                    ach = p.update_flops / (union * 1e-3) * 1e-12
                    out["roofline"]["with_overlapped_block_column_launches"] = {
                        "achieved": ach, "frac": ach / PEAK_FP64_MFMA_TFLOPS, "busy_ms_per_step": union,
                        "algorithmic_flops_per_step": p.update_flops,
                        "note": "union of the HIP-event intervals of the wide SYRK launches and the block-column launches "
                                "that overlap them; per-launch list: --dump-intervals"}
                out["roofline"].update(pmc_traffic(args.n))
                try:                                         # the measured instruction ceiling (include/george_amd_debug.h), after the timed region
                    mo = (C.c_double * 8)()
                    job.N.check(job.N.lib.gh_microbench_mfma_f64_ceiling(mo, 8))
                    out["roofline"]["peak_measured"] = float(mo[3])
                    out["roofline"]["frac_of_measured"] = ach_syrk / float(mo[3])
                    out["roofline"]["peak_measured_note"] = ("bare v_mfma_f64_16x16x4_f64 issue loop, 64x more workgroups than slots, "
                                                             "%.1f / %.1f / %.1f TFLOP/s at 1 / 2 / 4 wavefronts per SIMD" % (mo[0], mo[1], mo[2]))
                except Exception as e:
                    out["roofline"]["peak_measured_error"] = repr(e)
                try:
                    # the same kernel ALONE on the chip (after the timed region): one SYRK-shaped launch per panel width of the
                    # factorisation, M = 32768.  `achieved` above is the kernel in place -- beside the panel chain of the next
                    # panel, which with 2048-column panels (round 6) holds its share of the CUs twice as long
                    out["roofline"]["standalone"] = syrk_standalone(job, local_rank)
                except Exception as e:
                    out["roofline"]["standalone"] = {"error": repr(e)[:160]}
                # the launches behind `achieved`, so that the union can be re-derived from the line itself (and from
                # profiles/<round>/update_intervals_N<n>.json, written by --dump-intervals)
                iv = job.update_intervals()
                if len(iv):
                    order = np.argsort(iv[:, 0])
                    tot, lo, hi = 0.0, iv[order[0], 0], iv[order[0], 1]
                    for q in order[1:]:
                        if iv[q, 0] > hi:
                            tot, lo, hi = tot + hi - lo, iv[q, 0], iv[q, 1]
                        elif iv[q, 1] > hi:
                            hi = iv[q, 1]
                    tot += hi - lo
                    out["roofline"]["launch_intervals"] = {"n": int(len(iv)), "union_ms_recomputed": tot,
                                                           "sum_gflop": float(iv[:, 2].sum() * 1e-9)}
                    if args.dump_intervals:
                        with open(args.dump_intervals, "w") as f:
                            json.dump({"n": args.n, "nb": args.nb, "ms_update_union": union, "update_flops": p.update_flops,
                                       "achieved_tflops": p.update_flops / (union * 1e-3) * 1e-12 if union > 0 else None,
                                       "launches_start_end_ms_flops": iv.tolist()}, f)
            out["phases_ms"] = {"total_compute": p.ms_total, "kernel_matrix_build": p.ms_build,
                                "panel_factor_trsm": p.ms_panel, "trailing_syrk": p.ms_trailing,
                                "forward_solve": p.ms_solve}
            out["timing_note"] = ("the headline steps run with profile=1: one hipEvent pair around the build, every "
                                  "panel and every trailing launch INSIDE the timed region (~130 event records per step)")
            if p.ms_build > 0:
                npad = -(-args.n // 128) * 128
                tiles = (npad // 128) * (npad // 128 + 1) // 2
                bytes_alg = tiles * 128.0 * 128.0 * 8.0 + 2.0 * 8.0 * args.n          # lower 128-tiles written + x, yerr read
                out["roofline_kernel_build"] = {
                    "kernel": "kmat_interior_kernel<ExpSquared, 1> (lower 128-tiles of K(x,x) + diag(yerr^2); edge and diagonal tiles through the general tile code of the same launch)",
                    "bound": "hbm", "achieved": bytes_alg / (p.ms_build * 1e-3) * 1e-9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": bytes_alg / (p.ms_build * 1e-3) * 1e-9 / PEAK_HBM_GBS, "traffic": None,
                    "algorithmic_bytes": bytes_alg, "ms": p.ms_build}
            job.close()
            parity = {}
            gold = golden_ll(args.n)
            if gold is not None:
                parity["headline"] = {"n": args.n, "ll_gpu": ll, "ll_ref": gold[0], "rel": abs(ll - gold[0]) / abs(gold[0]),
                                      "ref": "tests/golden/large.json[%s]: reference C++ evaluator + LAPACK in the build "
                                             "container (oracle/gen_golden_large.py)" % gold[1]}
            if not args.no_extra:
                out["public_api"] = public_api_report(args.n, local_rank)
                out["value_public_api"] = {
                    "value": out["public_api"]["value_tflops"], "ms_per_step": out["public_api"]["seconds_per_step"] * 1e3,
                    "rel_to_value": out["public_api"]["value_tflops"] / value,
                    "note": "the metric as SURVEY 8(d) words it: GP.compute(x, yerr); GP.log_likelihood(y) on NumPy arrays, host->device "
                            "of x / yerr / y and the Python facade inside; `value` itself has the inputs resident in HBM"}
                if args.n != 16384:
                    j2 = DenseJob(16384, args.nb, local_rank, profile=False)
                    ts2, ll2 = run_steps(j2, 7, 2)
                    e2 = float(np.mean(ts2)) * 5
                    out["config"]["also_configs1_N16384"] = {
                        "seconds_per_step": e2 / 5, "value_tflops": flops_alg(16384) / (e2 / 5) * 1e-12,
                        "frac_of_fp64_mfma_peak": flops_alg(16384) / (e2 / 5) * 1e-12 / PEAK_FP64_MFMA_TFLOPS,
                        "log_likelihood": ll2, "per_step_s": spread(ts2)}
                    j2.close()
                    g2 = golden_ll(16384)
                    if g2 is not None:
                        parity["configs1"] = {"n": 16384, "ll_gpu": ll2, "ll_ref": g2[0], "rel": abs(ll2 - g2[0]) / abs(g2[0])}
                out["config"]["also_C4"] = hodlr_report(262144, local_rank, cpu_n=0 if args.no_cpu else 32768)
                if "parity" in out["config"]["also_C4"]:
                    parity["C4_hodlr"] = out["config"]["also_C4"]["parity"]
                if "parity_cpu_sample" in out["config"]["also_C4"]:
                    parity["C4_hodlr_cpu_sample"] = out["config"]["also_C4"]["parity_cpu_sample"]
                out["config"]["also_C5"] = c5_report(local_rank)
                try:
                    out["config"]["also_f1"] = f1_report(local_rank, cpu_n=0 if args.no_cpu else 2048)
                    cr = out["config"]["also_f1"].get("cpu_reference")
                    if cr and "rel_ll" in cr:
                        parity["f1_hyper_kernel_N2048"] = {"rel": cr["rel_ll"], "ll_ref": cr["log_likelihood"], "grad_rel_max": cr["grad_rel_max"]}
                except Exception as e:
                    out["config"]["also_f1"] = {"error": repr(e)[:200]}
                try:
                    out["config"]["also_abi_multi_gpu_world_of_one"] = mgpu_abi_report(local_rank)
                    parity["abi_multi_gpu_world_of_one"] = out["config"]["also_abi_multi_gpu_world_of_one"]["parity"]
                except Exception as e:                               # (RCCL missing on the box: say so, keep the line)
                    out["config"]["also_abi_multi_gpu_world_of_one"] = {"error": repr(e)}
                out["config"]["also_abi_multi_device"] = multi_device_probe()
                try:
                    out["config"]["also_curve"] = curve_report(local_rank)
                except Exception as e:
                    out["config"]["also_curve_error"] = repr(e)
            if not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(args.cpu_n)
                jp = DenseJob(args.cpu_n, args.nb, local_rank, profile=False)       # the GPU at the SAME N as the CPU sample
                llp = jp.step()
                jp.close()
                llr = out["cpu_baseline"]["log_likelihood"]
                parity["cpu_sample"] = {"n": args.cpu_n, "ll_gpu": llp, "ll_ref": llr, "rel": abs(llp - llr) / abs(llr)}
                if args.cpu_n2 > 0 and not args.no_extra:
                    try:
                        sp = cpu_baseline_second_point(args.cpu_n2, local_rank)
                        out["cpu_baseline"]["second_point"] = sp
                        parity["cpu_second_point"] = {"n": sp["n"], "rel": sp["rel"]}
                    except Exception as e:
                        out["cpu_baseline"]["second_point_error"] = repr(e)
            if parity:
                out["parity"] = parity
                out["parity"]["bound"] = 1e-6
                out["parity"]["ok"] = all(v["rel"] <= 1e-6 for v in parity.values() if isinstance(v, dict))
        if world > 1:
            out["timeline_rank0_ms"] = tline                     # panel / exchange / gather chain over the timed steps
    if world > 1:
        # the ranks leave (and give their GPUs back) BEFORE rank 0 runs the second multi-GPU form and prints the line
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        job = None
        if not selftest:
            torch.cuda.empty_cache()
            if args.abi_timeout > 0 and not args.share_gpu:
                out["abi_form"] = abi_form_report(args, world)
    if rank == 0:
        emit(out, args.detail or None)
        c5 = out.get("config", {}).get("also_C5", {})
        if c5 and (not c5.get("fused_not_slower_than_separate_calls", True) or "flop_model_error" in c5):
            sys.stderr.write("bench.py: C5 CHECK FAILED: fused objective slower than the separate calls, or a rate above "
                             "peak: %r\n" % ({k: c5.get(k) for k in ("compute_loglike_s", "grad_s", "fused_nll_and_grad_all_s",
                                                                    "flop_model_error")},))
            sys.exit(4)
        if isinstance(out.get("parity"), dict) and not out["parity"].get("ok", True):
            sys.stderr.write("bench.py: PARITY FAILURE (relative log-likelihood difference above 1e-6): %r\n" % (out["parity"],))
            sys.exit(3)


if __name__ == "__main__":
    main()
