// gh_hodlr.hip -- level-batched HODLR solver on one MI355X.
//
// Replaces the reference's recursive, single-threaded HODLR factorisation
// (include/george/hodlr.h: Node ctor :29-66, low_rank_approx :136-221, compute :75-103,
// factorize :223-235, apply_inverse :237-254, solve :107-114; driver src/george/solvers/_hodlr.cpp
// :55-94).  The algebra is the same -- off-diagonal blocks compressed by partially-pivoted ACA with
// randomly chosen rows, leaves factored exactly, one Woodbury step per internal node -- but the
// recursion is turned inside out so that every kernel launch works on ALL nodes of a tree level:
//
//   * the binary tree (hodlr.h:48: internal iff size/2 >= min_size) is built on the host;
//   * ACA for all nodes of a level runs as one launch, one workgroup per node, evaluating kernel
//     rows/columns on the fly (gh_eval.h) -- the N x N matrix is never formed;
//   * (layout note) V is only ever read one level at a time, so VA is stored LEVEL-MAJOR: level l is a
//     contiguous N x R_l row-major block at element offset N * off_l; U is needed both ways -- all
//     shallower levels at once during the factorisation sweep (row-major N x Rtot, UA) and one level
//     at a time in every solve (a level-major copy UL made once at the end of compute()).  With both
//     in the row-major form every level pass of a solve touched all 5 cache lines of a 75-column row
//     for the 3-15 columns it needed.
//   * U and V of every level live in two N x Rtot row-major arrays (UA, VA): column block l holds
//     level l, row i the point i (rows [start, start+half) of a node hold its U_[0] / V_[0], the
//     rest U_[1] / V_[1], zero-padded to the level's max rank).  "Apply the inverse of level l to
//     the U's of all its ancestors" (hodlr.h:95-102) is then ONE multi-column solve on the
//     contiguous column range [0, off_l) of UA;
//   * leaves and the 2r x 2r Woodbury cores S are inverted explicitly once (batched Gauss-Jordan
//     with partial pivoting; log|det| = sum log|pivot| as hodlr.h:87-93), so that every apply is a
//     batched small dense product: reduce (V^T x per 128-row chunk) -> sum -> S^-1 -> update.
//
// Differences from the reference that stay inside its own tolerance criterion (tests compare to
// the dense answer with allclose): one RNG stream per node (the reference threads ONE mt19937
// through the pre-order construction, so the row choices differ); partial-pivot LU / Gauss-Jordan
// instead of Eigen FullPivLU / LDLT; a block whose residual rows have ALL dropped under the 1e-14
// pivot threshold keeps its low-rank factors where the reference switches to the exact block
// (hodlr.h:160-176; same block to 1e-14 per entry, rank r instead of min(rows, cols)).  Ranks grow
// as far as the tolerance asks, up to RANK_CAP = 1024 (the scratch starts at 256 columns and a
// level is redone with twice as many when a block is cut short); a block that would need more, or
// more than a caller-given opts.max_rank, is an ERROR (GH_ERR_BAD_ARG), never a silent truncation.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "gh_common.h"
#include "gh_threads.h"
#include "gh_gemm_tile.h"
#include "gh_spin.h"

// gh_potf2.hip: batched 128x128 Cholesky + inverse of the factor (block b at A + b*stride_a)
int gh_launch_potf2_batched(double* A, int64_t lda, int64_t stride_a, double* dinv, int64_t stride_d, long long* info,
                            int nbatch, hipStream_t st);
// ... and the leaf form: block b -> K_b^-1 (full symmetric, in place) and logdet[b] = log|K_b|, nothing else written
int gh_launch_potf2_kinv_batched(double* A, int64_t lda, int64_t stride_a, double* logdet, long long* info, int nbatch, hipStream_t st);
int gh_launch_potf2_kinv_kernel_batched(double* A, int64_t lda, int64_t stride_a, double* logdet, long long* info, int nbatch,
                                        const GhFast& fast, const double* x, const double* yerr, int nd, const void* leaves, hipStream_t st);

#define HCH 128          // rows per reduce/update chunk
#define CPASS 256        // columns handled per pass of an apply
#define RANK_CAP 1024    // hard ceiling on a block's ACA rank (scratch n x rank, 2 rank x 2 rank cores)

// ------------------------------------------------------------------ device structs
struct LvlNode { int start, half, size, pad; };      // pad: added to the node's index where the ACA seeds its generator
struct Chunk { int node, half, row0, nrows; };
struct MMJob { long a_off; int b_row, o_row, m, kd; };
struct LeafDesc { int start, size; long off; };

__device__ __forceinline__ double hw_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block-wide sum broadcast to all threads; `sh` needs blockDim/64 doubles
__device__ __forceinline__ double hw_block_sum(double v, double* sh) {
  v = hw_wave_sum(v);
  const int nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += sh[w];
  return t;
}
// block-wide argmax of (val, idx): largest val, smallest idx on ties (Eigen maxCoeff order)
__device__ __forceinline__ void hw_block_argmax(double& val, int& idx, double* shv, int* shi) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(val, off, 64);
    const int oi = __shfl_down(idx, off, 64);
    if (ov > val || (ov == val && oi >= 0 && (idx < 0 || oi < idx))) { val = ov; idx = oi; }
  }
  const int nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = val; shi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  double bv = shv[0];
  int bi = shi[0];
  for (int w = 1; w < nw; ++w)
    if (shv[w] > bv || (shv[w] == bv && shi[w] >= 0 && (bi < 0 || shi[w] < bi))) { bv = shv[w]; bi = shi[w]; }
  val = bv; idx = bi;
}

// =========================================================================== ACA
// hodlr.h:136-221 for every internal node of one level.  Tcm is column-major scratch
// (Tcm[k*N + i]): for a node, entries at its first-half rows hold V(:,k) (the block's columns),
// at its second-half rows U(:,k).
#define ACA_THREADS 512
#ifndef ACA_WAVES_PER_EU
#define ACA_WAVES_PER_EU 4     // one-workgroup nodes: <= 128 registers per lane -- two of these workgroups, or one and 256 registers of other kernels, per SIMD
#endif
#ifndef ACA_WAVES_PER_EU_CL
#define ACA_WAVES_PER_EU_CL 2  // the cooperative launch: the critical chain of phase 1 keeps the registers it wants (no spills)
#endif
#define ACA_MAXR 2048          // coefficient slots in LDS: rank <= 2048 (one-workgroup nodes) / 1024 (clusters)
#define ACA_NC 64              // candidate rows tested per search pass once the search has started failing
#define ACA_LIDX 4096          // row permutations of one-workgroup nodes live in LDS up to this many rows
#ifndef ACA_CAPD
#define ACA_CAPD 2048          // doubles of U and of V a one-workgroup node mirrors in LDS (its first CAPD / n rows of each factor)
#endif
#define ACA_XC 512             // ... and its coordinates when ndim == 1 (rows, then columns)
#define ACA_DYN_BYTES (ACA_CAPD > 0 ? (2 * ACA_CAPD + 2 * ACA_XC) * 8 : 0)
// One node is worked on by a CLUSTER of G workgroups (blockIdx.x = node * G + g): the top levels
// have 1, 2, 4, ... nodes with blocks of N/2, N/4, ... rows, and one workgroup per node left the
// single workgroup of level 0 with 70 % of the whole HODLR compute() at N = 262144.  Workgroup g
// owns the columns and rows  t = g * 512 + tid (+ G * 512 ...)  of the block; the cluster meets
// at three barriers per ACA step (row chosen / pivot search / norms), each a monotonic counter in
// HBM, and exchanges its partial results (arg-max candidates, partial sums) through `part`.
// Every workgroup reduces the SAME partials in the SAME order, so all of them take identical
// decisions without a broadcast.  Data written by another workgroup of the cluster is read with
// agent-scope atomic loads (a plain load may hit a stale line of this CU's L1).  G = 1 is the
// old one-workgroup-per-node kernel (no counters touched).  The launch keeps nodes * G <= 256 so
// that the whole grid is resident (a spinning cluster member never waits for an unscheduled one);
// a spin that outlasts ~2 s raises `*fail` and bails out instead of hanging the GPU.
struct AcaShared {
  double shd[8];
  int shi[8];
  double coef[ACA_MAXR];
  int s_i;
  // batched candidate search (one-workgroup nodes)
  int cand_k[ACA_NC], cand_i[ACA_NC], cand_tail[ACA_NC], cand_bestn[ACA_NC];
  double cand_best[ACA_NC];
  unsigned long long cand_st[ACA_NC];
  unsigned short lidx[ACA_LIDX];
  double pivv;
};
// stores of values that another workgroup of the cluster will read: agent-scope atomics (write-through,
// visible to the other XCDs' atomic loads once s_waitcnt has seen them complete) -- no release fence
__device__ __forceinline__ void aca_st(double* p, double v, bool shared_w) {
  if (shared_w) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ double aca_ld(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int aca_ldi(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// cluster barrier number `epoch` (0, 1, 2, ...) on counter `bar`; returns false on time-out
__device__ __forceinline__ bool aca_barrier(unsigned* bar, int G, unsigned& epoch, int* fail, int fence) {
  if (G == 1) { __syncthreads(); return true; }
  __shared__ int ok;
  // Release side without a fence: everything this workgroup wrote for the others went out as
  // agent-scope atomic stores (aca_st), and s_waitcnt makes every lane's stores complete before the
  // arrival is counted.  __threadfence() here writes this XCD's L2 back at every barrier -- three
  // per ACA step, 128 workgroups: 60 us per barrier, 2.8 of the 7 ms of ACA time at N = 262144.
  // (The `fence` argument restores it; its environment switch went in round 4.)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fence) __threadfence();
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (unsigned)G * (epoch + 1u);
    GhSpin spin(fail);                                      // (gh_spin.h: the 2-s give-up and the abort word)
    int good = 1;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (!spin.keep_waiting(0u)) { good = 0; break; }
    }
    if (!good) atomicExch(fail, 1);
    // (no acquire fence: everything another cluster member wrote is read with agent-scope atomic
    //  loads, which go past this XCD's caches; a fence here would invalidate the L2 at every barrier)
    ok = good;
  }
  __syncthreads();
  ++epoch;
  return ok != 0;
}

// Several levels in ONE launch: workgroups [wg0, wg0 + nwg) work on the level described by a segment
// (the per-level arguments of the kernel are then taken from it).  The clustered levels of a tree
// are launched this way, 256 workgroups in all, so that every cluster is resident whatever the
// others do; launched one after the other they were 3 of the 9 ms of a C4 compute().
struct AcaSeg {
  const LvlNode* nodes; double* Tcm; int* idx; int* ranks; unsigned* bars; double* part; int* sel; int* fail; int* trunc;
  int level, G, wg0, nwg;
  // one-workgroup segments: dur[node] <- how long the node took (10-ns ticks); order != nullptr: workgroup q of the segment takes
  // node order[q] (the host's longest-first order from the previous compute() of the handle)
  int* dur; const int* order;
};
// CL: the CLUSTER instantiation (G > 1: the cooperative launch of the top levels) without the one-workgroup-only machinery --
// batched candidate search, LDS mirrors, LDS row permutation; !CL: one workgroup per node (G == 1) without the cluster protocol.
// One kernel for both needed 225 registers per lane: the cooperative launch -- one 512-thread workgroup on every CU for 1.4 ms,
// two wavefronts per SIMD -- then held 464 of each SIMD's 512 registers, and nothing else of phase 1 (one-workgroup nodes 225,
// leaf Cholesky 256, leaf build 127) could share a SIMD with it (profiles/r06/hodlr_phase1_registers.md).
template <bool FAST, bool CL>
__global__ __launch_bounds__(ACA_THREADS, CL ? ACA_WAVES_PER_EU_CL : ACA_WAVES_PER_EU) void hodlr_aca_kernel(
    const GhNode* __restrict__ prog, int n_prog, GhFast fast, int nd, const double* x, const LvlNode* nodes, double* Tcm, long N,
    int rcap, int* idx, int* ranks, double tol, unsigned long long seed, int level,
    int G_, unsigned* bars, double* part, int pstride, int* sel, int* fail, int multi, int fence, int* trunc,
    const AcaSeg* segs, int nseg, int capd_) {
  int G = CL ? G_ : 1;
  const int capd = CL ? 0 : capd_;
#ifdef ACA_CL_SETPRIO
  if (CL) __builtin_amdgcn_s_setprio(ACA_CL_SETPRIO);       // the critical chain of phase 1 first at every SIMD's arbiter
#endif
  __shared__ AcaShared sh;
  extern __shared__ double aca_dyn[];                  // capd > 0: U mirror | V mirror | coordinates (ACA_DYN_BYTES)
  int bid = blockIdx.x;
  int* dur = nullptr;
  const int* order = nullptr;
  if (segs) {
    int q = 0;
    while (q + 1 < nseg && bid >= segs[q].wg0 + segs[q].nwg) ++q;
    const AcaSeg sg = segs[q];
    nodes = sg.nodes; Tcm = sg.Tcm; idx = sg.idx; ranks = sg.ranks; bars = sg.bars; part = sg.part; sel = sg.sel;
    fail = sg.fail; trunc = sg.trunc; level = sg.level; G = CL ? sg.G : 1;
    bid -= sg.wg0;
    if (!CL) { dur = sg.dur; order = sg.order; }
  }
  const long long t_begin = dur ? wall_clock64() : 0;
  const int node = order ? order[bid] : bid / G, g = order ? 0 : bid % G;
  const LvlNode nodev = nodes[node];
  const int col0 = nodev.start, n_cols = nodev.half;
  const int row0 = nodev.start + nodev.half, n_rows = nodev.size - nodev.half;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int t0 = g * nt + tid, ts = G * nt;            // this thread's first column/row and its stride
  const bool sw = G > 1;                               // values other workgroups read go out as atomics
  unsigned* bar = bars + node;
  double* mypart = part + ((long)node * G + g) * pstride;
  const double* allpart = part + (long)node * G * pstride;
  unsigned epoch = 0;
  const int full_rank = n_rows < n_cols ? n_rows : n_cols;
  int max_rank = full_rank;
  if (max_rank > rcap) max_rank = rcap;
  // one-workgroup nodes keep the row permutation in LDS (the candidate draws are a serial chain of
  // dependent reads and writes: ~1 us each through HBM, 64 of them per search pass)
  const bool lperm = !CL && (G == 1) && n_rows <= ACA_LIDX;
  // Small one-workgroup nodes (levels 8-10 of C4: 1792 of its 2047 blocks) are a chain of ~10 dependent global round trips
  // per ACA step -- 13 us per step for 128-entry vectors.  They mirror the first `kcap` rows of U and V (and, in 1-D, their
  // coordinates) in LDS: every read of a factor entry below comes from the mirror when its row is there, every write goes to
  // both.  Same values, same order of operations: the factors and ranks do not change by a bit.
  const int kcap = (lperm && capd > 0 && n_rows <= capd && n_cols <= capd) ? min(capd / n_rows, capd / n_cols) : 0;
  double* const uc = aca_dyn;
  double* const vc = aca_dyn + capd;
  double* const xs = aca_dyn + 2 * capd;
  const bool xlds = kcap > 0 && nd == 1 && n_rows <= ACA_XC && n_cols <= ACA_XC;
  if (xlds) {
    for (int t = tid; t < n_rows; t += nt) xs[t] = x[row0 + t];
    for (int t = tid; t < n_cols; t += nt) xs[ACA_XC + t] = x[col0 + t];
  }
  auto xrow = [&](int m) -> const double* { return xlds ? (const double*)(xs + m) : x + (long)(row0 + m) * nd; };
  auto xcol = [&](int n) -> const double* { return xlds ? (const double*)(xs + ACA_XC + n) : x + (long)(col0 + n) * nd; };
  if (lperm) { for (int t = tid; t < n_rows; t += nt) sh.lidx[t] = (unsigned short)t; }
  else if (g == 0) { for (int t = tid; t < n_rows; t += nt) idx[row0 + t] = t; }
  int remaining = n_rows, rank = 0;
  int batch = 8;                                       // candidates per search pass: 8, then ACA_NC once a pass has failed
  double norm = 0.0;
  const double tol2 = tol * tol;
  bool converged = false;
  // (nodev.pad: index of the launch's first node in its tree level -- non-zero only for the sub-trees of a split tree)
  unsigned long long st = seed ^ ((unsigned long long)(level + 1) << 40) ^ ((unsigned long long)(node + nodev.pad) * 0x9E3779B97F4A7C15ull);
  __syncthreads();
#ifdef GH_ACA_TIMES
  const long long dbg_t0 = wall_clock64();
  int dbg_passes = 0;
#endif
  bool have_sel = false;                               // clusters: the next candidate row has been drawn and published already
  auto draw_row = [&]() {                              // (workgroup 0, thread 0 of the cluster) hodlr.h:159-176: a random unused row
    st += 0x9E3779B97F4A7C15ull;
    unsigned long long z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const int k = (int)(z % (unsigned long long)remaining);
    int pick;
    if (lperm) { pick = sh.lidx[k]; sh.lidx[k] = sh.lidx[remaining - 1]; }
    else { pick = idx[row0 + k]; idx[row0 + k] = idx[row0 + remaining - 1]; }
    if (sw) __hip_atomic_store(sel + node, pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else sel[node] = pick;
  };
  while (rank < max_rank) {
    // ---- choose a random unused row with a non-negligible residual (hodlr.h:159-191)
    bool got = false;
    int j = -1;
    double pivot = 0.0;
    // One-workgroup nodes test a BATCH of candidate rows per pass, wavefront w the candidates
    // w, w + 8, ...  Towards the end of a node's ACA every remaining row is below the 1e-14
    // threshold and the search walks through all of them before giving up (hodlr.h:159-191 does
    // too): one row per pass made levels 8 and 9 of C4 cost 2.6 ms for rank-3 blocks, eight per pass
    // 0.75 + 0.6 ms; after the first pass without a hit the batch grows to 64.  Same result as the
    // one-by-one search: the candidates are drawn in the same order, the first that passes wins,
    // and the draws after it are undone (row permutation and generator state restored).
    while (!CL && (multi & 1) && lperm && remaining > 0 && rank <= 32) {
      int NC = remaining < batch ? remaining : batch;
      if (rank + NC > rcap) NC = rcap - rank;
      if (NC < 1) break;
#ifdef GH_ACA_TIMES
      ++dbg_passes;
#endif
      if (tid == 0) {
        for (int c = 0; c < NC; ++c) {
          st += 0x9E3779B97F4A7C15ull;
          unsigned long long z = st;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          z ^= z >> 31;
          const int k = (int)(z % (unsigned long long)(remaining - c));
          sh.cand_k[c] = k;
          sh.cand_i[c] = sh.lidx[k];
          sh.cand_tail[c] = sh.lidx[remaining - c - 1];
          sh.lidx[k] = (unsigned short)sh.cand_tail[c];
          sh.cand_st[c] = st;
        }
      }
      __syncthreads();
      const int lane = tid & 63, wave = tid >> 6;
      for (int c = wave; c < NC; c += (nt >> 6)) {
        const int i = sh.cand_i[c];
        double* cw = sh.coef + wave * 32;
        __builtin_amdgcn_wave_barrier();              // (the previous candidate's reads of cw are done)
        for (int k = lane; k < rank; k += 64) cw[k] = k < kcap ? uc[k * n_rows + i] : Tcm[(long)k * N + row0 + i];
        __builtin_amdgcn_s_waitcnt(0);                // (own wavefront's LDS writes, read back below)
        __builtin_amdgcn_wave_barrier();
        double best = -1.0;
        int bestn = -1;
        const double* xi = xrow(i);
        const int kv = rank < kcap ? rank : kcap;
        // (only the candidate's largest entry is kept: storing 8-64 residual rows per pass to keep one was most of this
        //  kernel's write traffic; the chosen row is formed again below, by the whole workgroup)
        for (int n = lane; n < n_cols; n += 64) {
          double v = FAST ? gh_fast_value(fast, xi, xcol(n)) : gh_eval_value(prog, n_prog, xi, xcol(n));
          for (int k = 0; k < kv; ++k) v -= cw[k] * vc[k * n_cols + n];
          for (int k = kv; k < rank; ++k) v -= cw[k] * Tcm[(long)k * N + col0 + n];
          const double a = fabs(v);
          if (a > best) { best = a; bestn = n; }
        }
        for (int off = 32; off > 0; off >>= 1) {          // largest value, smallest column on ties
          const double ov = __shfl_down(best, off, 64);
          const int oi = __shfl_down(bestn, off, 64);
          if (ov > best || (ov == best && oi >= 0 && (bestn < 0 || oi < bestn))) { best = ov; bestn = oi; }
        }
        if (lane == 0) { sh.cand_best[c] = best; sh.cand_bestn[c] = bestn; }
      }
      __syncthreads();
      int chosen = -1;
      for (int c = 0; c < NC; ++c)
        if (sh.cand_best[c] >= 1e-14) { chosen = c; break; }                           // hodlr.h:191
      if (chosen < 0) {
        remaining -= NC;
        // (round 6) The first pass without a hit: before walking the rest of the rows 64 at a time, SCREEN them all -- thread t the
        // rows lidx[t], lidx[t + 512], ..., a whole row each (no cross-lane reduction, V entries and column points as LDS
        // broadcasts), leaving as soon as anybody has found an entry >= 1e-14.  If nobody has, every remaining row would fail
        // its test: the search ends as it would after the walk (rows exhausted, same rank, same factors).  The walk of the one
        // such node of C4's level 8 took 806 us -- twelve passes -- and was the tail of phase 1; its screen is ~50 us.
        if (batch != ACA_NC && remaining > 0 && rank <= 8 && (long)n_rows * n_cols <= 512L * 512L) {    // (a thread walks whole rows: 512 entries here; 0.7 ms for a 2048 x 2048 block)
          if (tid == 0) sh.s_i = 0;
          __syncthreads();
          const int kv = rank < kcap ? rank : kcap;
          for (int q = tid; q < remaining && !*(volatile int*)&sh.s_i; q += nt) {
            const int i = sh.lidx[q];
            double cu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) cu[k] = k < rank ? (k < kcap ? uc[k * n_rows + i] : Tcm[(long)k * N + row0 + i]) : 0.0;
            const double* xi = xrow(i);
            bool hit = false;
            for (int n = 0; n < n_cols && !hit; ++n) {
              double v = FAST ? gh_fast_value(fast, xi, xcol(n)) : gh_eval_value(prog, n_prog, xi, xcol(n));
#pragma unroll
              for (int k = 0; k < 8; ++k) if (k < rank) v -= cu[k] * (k < kv ? vc[k * n_cols + n] : Tcm[(long)k * N + col0 + n]);
              hit = fabs(v) >= 1e-14;
              if ((n & 31) == 31 && *(volatile int*)&sh.s_i) break;
            }
            if (hit) *(volatile int*)&sh.s_i = 1;
          }
          __syncthreads();
          if (!*(volatile int*)&sh.s_i) remaining = 0;
          __syncthreads();
        }
        batch = ACA_NC;
        __syncthreads();
        continue;
      }
      if (tid == 0) {                                 // undo the draws after the chosen one, last first
        for (int c = NC - 1; c > chosen; --c) {
          sh.lidx[sh.cand_k[c]] = (unsigned short)sh.cand_i[c];
          sh.lidx[remaining - c - 1] = (unsigned short)sh.cand_tail[c];
        }
      }
      st = sh.cand_st[chosen];                        // (every thread keeps the generator state in step)
      remaining -= chosen + 1;
      j = sh.cand_bestn[chosen];
      {
        // the chosen candidate's residual row into row `rank` of the scratch: same expression and order of k as in the
        // search, so the same bits
        const int i = sh.cand_i[chosen];
        for (int k = tid; k < rank; k += nt) sh.coef[k] = k < kcap ? uc[k * n_rows + i] : Tcm[(long)k * N + row0 + i];
        __syncthreads();
        const double* xi = xrow(i);
        const int kv = rank < kcap ? rank : kcap;
        for (int n = tid; n < n_cols; n += nt) {
          double v = FAST ? gh_fast_value(fast, xi, xcol(n)) : gh_eval_value(prog, n_prog, xi, xcol(n));
          for (int k = 0; k < kv; ++k) v -= sh.coef[k] * vc[k * n_cols + n];
          for (int k = kv; k < rank; ++k) v -= sh.coef[k] * Tcm[(long)k * N + col0 + n];
          Tcm[(long)rank * N + col0 + n] = v;
          if (rank < kcap) vc[rank * n_cols + n] = v;
        }
      }
      __syncthreads();
      pivot = rank < kcap ? vc[rank * n_cols + j] : Tcm[(long)rank * N + col0 + j];
      got = true;
      break;
    }
    while (!got && (have_sel || remaining > 0)) {
      if (!have_sel) {
        if (g == 0 && tid == 0) draw_row();
        --remaining;
        if (!aca_barrier(bar, G, epoch, fail, fence)) return;                         // B1: row chosen
      }
      have_sel = false;
      const int i = (G == 1) ? sel[node] : aca_ldi(sel + node);
      for (int k = tid; k < rank; k += nt) sh.coef[k] = aca_ld(Tcm + (long)k * N + row0 + i);   // U(i, 0:rank)
      __syncthreads();
      double best = -1.0, bestv = 0.0;
      int bestn = -1;
      const double* xi = x + (long)(row0 + i) * nd;
      for (int n = t0; n < n_cols; n += ts) {
        double v = FAST ? gh_fast_value(fast, xi, x + (long)(col0 + n) * nd)
                        : gh_eval_value(prog, n_prog, xi, x + (long)(col0 + n) * nd);
        for (int k = 0; k < rank; ++k) v -= sh.coef[k] * Tcm[(long)k * N + col0 + n];
        Tcm[(long)rank * N + col0 + n] = v;           // (rewritten after the pivot is known: owner-only so far)
        if (rank < kcap) vc[rank * n_cols + n] = v;
        const double a = fabs(v);
        if (a > best) { best = a; bestn = n; }
      }
      hw_block_argmax(best, bestn, sh.shd, sh.shi);
      if (G > 1) {
        if (tid == 0) {
          aca_st(mypart + 0, best, true);
          aca_st(mypart + 1, (double)bestn, true);
          aca_st(mypart + 2, bestn >= 0 ? Tcm[(long)rank * N + col0 + bestn] : 0.0, true);   // (this workgroup wrote it)
        }
        if (!aca_barrier(bar, G, epoch, fail, fence)) return;                         // B2: pivot search
        // every workgroup reduces the same G candidates with the same tree (thread q takes member q's):
        // largest value, smallest column on ties.  (A serial loop of 3 G device-scope loads in EVERY
        // thread was most of the 185 us an ACA step of the root node took.)
        double cv = -1.0, cvv = 0.0;
        int cn = -1;
        if (tid < G) {
          cv = aca_ld(allpart + (long)tid * pstride);
          cn = (int)aca_ld(allpart + (long)tid * pstride + 1);
          cvv = aca_ld(allpart + (long)tid * pstride + 2);
          if (cn < 0) cv = -1.0;
        }
        best = cv; bestn = cn;
        hw_block_argmax(best, bestn, sh.shd, sh.shi);
        if (tid < G && cn >= 0 && cn == bestn) sh.pivv = cvv;      // (columns are owned by one workgroup: a unique writer)
        __syncthreads();
        bestv = bestn >= 0 ? sh.pivv : 0.0;
      } else {
        bestv = bestn >= 0 ? Tcm[(long)rank * N + col0 + bestn] : 0.0;
      }
      if (best >= 1e-14) { got = true; j = bestn; pivot = bestv; break; }              // hodlr.h:191
    }
    // rows exhausted: every residual row tested below 1e-14 in absolute value -- keep the factors we
    // have (the reference returns the exact block as a rank-min(rows, cols) "trivial factorisation",
    // hodlr.h:160-176; the two represent the same block to 1e-14 per entry)
    if (!got) { converged = true; break; }
    // ---- normalise the row by its pivot, build the column (hodlr.h:194-199)
    __syncthreads();
    double vn2 = 0.0;
    for (int n = t0; n < n_cols; n += ts) {
      const double v = (rank < kcap ? vc[rank * n_cols + n] : Tcm[(long)rank * N + col0 + n]) / pivot;
      aca_st(Tcm + (long)rank * N + col0 + n, v, sw);
      if (rank < kcap) vc[rank * n_cols + n] = v;
      vn2 += v * v;
    }
    for (int k = tid; k < rank; k += nt) sh.coef[k] = k < kcap ? vc[k * n_cols + j] : aca_ld(Tcm + (long)k * N + col0 + j);    // V(j, 0:rank)
    __syncthreads();
    double un2 = 0.0;
    const double* xj = xcol(j);
    const int kvu = rank < kcap ? rank : kcap;
    for (int m = t0; m < n_rows; m += ts) {
      double u = FAST ? gh_fast_value(fast, xrow(m), xj) : gh_eval_value(prog, n_prog, xrow(m), xj);
      for (int k = 0; k < kvu; ++k) u -= sh.coef[k] * uc[k * n_rows + m];
      for (int k = kvu; k < rank; ++k) u -= sh.coef[k] * Tcm[(long)k * N + row0 + m];
      aca_st(Tcm + (long)rank * N + row0 + m, u, sw);
      if (rank < kcap) uc[rank * n_rows + m] = u;
      un2 += u * u;
    }
    ++rank;
    if (rank >= full_rank) { converged = true; break; }                                // hodlr.h:203
    if (rank >= max_rank) break;                                                       // rank cap: NOT converged
    // cross terms |u_new . u_k|, |v_new . v_k|, k < rank-1, of the norm estimate (hodlr.h:210-214):
    // this workgroup's share of each dot product, four at a time
    const double* ul = Tcm + (long)(rank - 1) * N + row0;
    const double* vl = Tcm + (long)(rank - 1) * N + col0;
    double maxu = 0.0, maxv = 0.0;
    // (round 6) ONE workgroup barrier for all the block sums of a step instead of two per sum: the
    // wavefronts' partial sums of the two squared norms and of the 2 (rank - 1) cross terms go to LDS (sh.coef is free between the
    // column build and the next step), thread k then adds the eight wavefront sums of term k in hw_block_sum's order -- the same
    // bits -- and publishes it.  The root of C4 (rank ~20) went through ~460 block sums, two barriers each.
#ifndef GH_ACA_BATCH_ONES
#define GH_ACA_BATCH_ONES 1
#endif
    const bool batched = (CL || GH_ACA_BATCH_ONES) && rank <= 120;
    if (batched) {
      const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
      double* const ws = sh.coef;                      // [(term * 2 + side) * 8 + wavefront]; term rank - 1 = the squared norms
      __syncthreads();                                 // (every thread is done with the column build's coefficients)
      {
        const double a = hw_wave_sum(un2), b = hw_wave_sum(vn2);
        if (lane == 0) { ws[((rank - 1) * 2) * 8 + wave] = a; ws[((rank - 1) * 2 + 1) * 8 + wave] = b; }
      }
      for (int k0 = 0; k0 < rank - 1; k0 += 4) {
        double du[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
        for (int m = t0; m < n_rows; m += ts) {
          const double u = rank - 1 < kcap ? uc[(rank - 1) * n_rows + m] : ul[m];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (k0 + q < rank - 1) du[q] += (k0 + q < kcap ? uc[(k0 + q) * n_rows + m] : Tcm[(long)(k0 + q) * N + row0 + m]) * u;
        }
        for (int n = t0; n < n_cols; n += ts) {
          const double v = rank - 1 < kcap ? vc[(rank - 1) * n_cols + n] : vl[n];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (k0 + q < rank - 1) dv[q] += (k0 + q < kcap ? vc[(k0 + q) * n_cols + n] : Tcm[(long)(k0 + q) * N + col0 + n]) * v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double a = hw_wave_sum(du[q]), b = hw_wave_sum(dv[q]);
          if (lane == 0 && k0 + q < rank - 1) { ws[((k0 + q) * 2) * 8 + wave] = a; ws[((k0 + q) * 2 + 1) * 8 + wave] = b; }
        }
      }
      __syncthreads();
      double ta = 0.0, tb = 0.0;
      if (tid < rank)
        for (int w = 0; w < nw; ++w) { ta += ws[(tid * 2) * 8 + w]; tb += ws[(tid * 2 + 1) * 8 + w]; }
      if (G > 1) {
        if (tid < rank - 1) { aca_st(mypart + 6 + 2 * tid, ta, true); aca_st(mypart + 7 + 2 * tid, tb, true); }
        if (tid == rank - 1) { sh.shd[0] = ta; sh.shd[1] = tb; }      // (published behind the next row's draw, below)
        __syncthreads();
        un2 = sh.shd[0]; vn2 = sh.shd[1];
      } else {
        __syncthreads();                               // (all partial sums read)
        if (tid < rank) { ws[tid] = tid < rank - 1 ? fabs(ta) : ta; ws[1024 + tid] = tid < rank - 1 ? fabs(tb) : tb; }
        __syncthreads();
        for (int k = 0; k < rank - 1; ++k) {
          if (ws[k] > maxu) maxu = ws[k];
          if (ws[1024 + k] > maxv) maxv = ws[1024 + k];
        }
        un2 = ws[rank - 1]; vn2 = ws[1024 + rank - 1];
        __syncthreads();                               // (coef is written again at the next step)
      }
    } else {
    un2 = hw_block_sum(un2, sh.shd);
    vn2 = hw_block_sum(vn2, sh.shd);
    for (int k0 = 0; k0 < rank - 1; k0 += 4) {
      double du[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
      for (int m = t0; m < n_rows; m += ts) {
        const double u = rank - 1 < kcap ? uc[(rank - 1) * n_rows + m] : ul[m];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (k0 + q < rank - 1) du[q] += (k0 + q < kcap ? uc[(k0 + q) * n_rows + m] : Tcm[(long)(k0 + q) * N + row0 + m]) * u;
      }
      for (int n = t0; n < n_cols; n += ts) {
        const double v = rank - 1 < kcap ? vc[(rank - 1) * n_cols + n] : vl[n];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (k0 + q < rank - 1) dv[q] += (k0 + q < kcap ? vc[(k0 + q) * n_cols + n] : Tcm[(long)(k0 + q) * N + col0 + n]) * v;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double a = hw_block_sum(du[q], sh.shd);
        const double b = hw_block_sum(dv[q], sh.shd);
        if (G > 1) {
          if (tid == 0 && k0 + q < rank - 1) { aca_st(mypart + 6 + 2 * (k0 + q), a, true); aca_st(mypart + 7 + 2 * (k0 + q), b, true); }
        } else {
          if (fabs(a) > maxu) maxu = fabs(a);
          if (fabs(b) > maxv) maxv = fabs(b);
        }
      }
    }
    }
    if (G > 1) {
      // (round 5) the NEXT step's first candidate row is drawn here and published with this barrier: the draw depends on the
      // generator and the row permutation alone, not on the norms, every member has read the current `sel` before it arrived at
      // B2, and a draw made in vain (the block converges below) changes nothing that is read again -- same draws in the same
      // order, one cluster barrier per step fewer (two instead of three)
      if ((multi & 2) && remaining > 0) {
        if (g == 0 && tid == 0) draw_row();
        --remaining;
        have_sel = true;
      }
      if (tid == 0) { aca_st(mypart + 3, un2, true); aca_st(mypart + 4, vn2, true); }     // (slots 0-2 may still be read by a slow member)
      if (!aca_barrier(bar, G, epoch, fail, fence)) return;                           // B3: norms (+ the next row)
      {
        double pu = 0.0, pv = 0.0;
        if (tid < G) { pu = aca_ld(allpart + (long)tid * pstride + 3); pv = aca_ld(allpart + (long)tid * pstride + 4); }
        un2 = hw_block_sum(pu, sh.shd);
        vn2 = hw_block_sum(pv, sh.shd);
      }
      {
        const int lane = tid & 63, wave = tid >> 6;          // wavefront w sums the G shares of the dot products k = w, w + 8, ...
        for (int k = wave; k < rank - 1; k += (nt >> 6)) {
          double a = 0.0, b = 0.0;
          for (int q = lane; q < G; q += 64) { a += aca_ld(allpart + (long)q * pstride + 6 + 2 * k); b += aca_ld(allpart + (long)q * pstride + 7 + 2 * k); }
          a = hw_wave_sum(a);
          b = hw_wave_sum(b);
          if (lane == 0) { sh.coef[k] = fabs(a); sh.coef[ACA_MAXR / 2 + k] = fabs(b); }   // (coef is free here: reloaded at the next step)
        }
      }
      __syncthreads();
      for (int k = 0; k < rank - 1; ++k) {
        if (sh.coef[k] > maxu) maxu = sh.coef[k];
        if (sh.coef[ACA_MAXR / 2 + k] > maxv) maxv = sh.coef[ACA_MAXR / 2 + k];
      }
      __syncthreads();
    }
    const double rowcol = un2 * vn2;
    if (rowcol < tol2 * norm) { converged = true; break; }                             // hodlr.h:206-207
    norm += rowcol;
    if (rank > 1) norm += 2.0 * maxu + 2.0 * maxv;
  }
  if (g == 0 && tid == 0) {
    ranks[node] = rank;
    if (dur) dur[node] = (int)(wall_clock64() - t_begin);
    if (!converged && rank < full_rank) atomicExch(trunc, 1);     // stopped by the cap, not by the tolerance
#ifdef GH_ACA_TIMES                                                // (build-time debugging aid: per-node durations of one-workgroup nodes)
    if (G == 1) { mypart[0] = (double)(wall_clock64() - dbg_t0); mypart[1] = (double)rank; mypart[2] = (double)remaining; mypart[3] = (double)dbg_passes; mypart[4] = (double)dbg_t0; }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ACA with ONE WAVEFRONT per node, for the deep levels whose blocks have at most 64 E rows and columns, E = 1, 2, 4 (levels
// 9 and 10 of C4 -- 256 x 256 and 128 x 128 blocks, 1536 of its 2047 nodes; the workgroup kernel above spent 124 of its
// 198 ms of workgroup time on them: a 512-thread workgroup, ~10 workgroup barriers and six block-wide reductions per ACA
// step for vectors of 128 or 256 entries, and one thread drawing up to 64 candidate rows per search pass).  Four nodes per
// 256-thread workgroup, no workgroup barrier anywhere.  Lane l owns rows and columns l + 64 e, e < E, and keeps ITS entries
// of the factors found so far (rank <= AW_RW) in registers: U[k][e], V[k][e].  What another lane's entry is needed for --
// the candidate row's coefficients U(i, :), the pivot column's V(j, :) -- travels by one shuffle per k.  LDS holds only the
// node's coordinates (1-D) and its row permutation: 4.5 KiB per node at E = 4, so these workgroups find room beside the
// cooperative launch, the leaf Cholesky and the one-workgroup nodes that share the chip with them in phase 1.
//
// SAME BITS as hodlr_aca_kernel with G = 1: the same generator and draws, candidates tested in drawing order (the first
// whose largest residual entry reaches 1e-14 wins -- what the workgroup kernel's batched search returns), residuals formed by
// the same expression with k ascending, and every sum reduced by the tree the workgroup kernel uses for <= 512 entries: there
// thread t holds element t alone, wavefront q reduces elements 64 q .. 64 q + 63 with the shfl_down tree and the wavefront
// results are added in order; here lane l holds elements l + 64 e, "virtual wavefront" e reduced by the same tree, then
// added in order.  Ranks, factors, log-determinants do not change by a bit (tests/test_gpu_hodlr.py).
// A node that needs more than AW_RW_OF(E) columns is NOT cut short: the launch raises the level's `trunc` word to 2 and the host
// redoes the level with the workgroup kernel (and remembers it for the handle's next compute()).
// rank capacity of the register mirrors: 8 columns, 6 where a lane holds four entries of each (256 x 256 blocks: 96 instead of 128
// registers of mirrors -- with 8 the kernel spilled 560 bytes per lane)
#define AW_RW_OF(E) ((E) >= 4 ? 6 : 8)
#define AW_NODES 4              // nodes (wavefronts) per workgroup
// (the lane's E entries of a vector as a clang extended vector, not an array: a run-time element index -- the candidate row's
//  slot -- is then an extractelement the backend lowers to selects; on arrays, however the selects were spelt, the optimiser
//  folded them back into a run-time array index and moved U and V to scratch memory)
template <int E> struct AwVec { typedef double type __attribute__((ext_vector_type(E))); };
template <> struct AwVec<1> { typedef double type __attribute__((ext_vector_type(2))); };      // (one entry used)
template <int E>
__device__ __forceinline__ double aw_sum(const typename AwVec<E>::type& x) {
  // (hw_block_sum of the workgroup kernel for <= 64 E entries: t = 0 + wave0 + wave1 + ...)
  double t = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) t += __shfl(hw_wave_sum(x[e]), 0, 64);
  return t;
}
template <int E>
__device__ __forceinline__ double aw_pick(const typename AwVec<E>::type& x, int slot) {      // x[slot], slot wave-uniform
  return E == 1 ? x[0] : x[slot];
}
template <bool FAST, int E>
__global__ __launch_bounds__(64 * AW_NODES) void hodlr_aca_wave_kernel(
    const GhNode* __restrict__ prog, int n_prog, GhFast fast, int nd, const double* x, const LvlNode* nodes, int n_nodes, double* Tcm, long N,
    int rcap, int* ranks, double tol, unsigned long long seed, int level, int* trunc) {
  constexpr int MR = 64 * E;                            // rows / columns capacity
  constexpr int AW_RW = AW_RW_OF(E);
  __shared__ double xs_all[AW_NODES][2 * MR];           // 1-D: [0, MR) row coordinates, [MR, 2 MR) column coordinates
  __shared__ unsigned short lidx_all[AW_NODES][MR];     // row permutation
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node = blockIdx.x * AW_NODES + wave;
  if (node >= n_nodes) return;                          // (no workgroup barrier in this kernel)
  double* const xs = xs_all[wave];
  unsigned short* const lidx = lidx_all[wave];
  const LvlNode nodev = nodes[node];
  const int col0 = nodev.start, n_cols = nodev.half;
  const int row0 = nodev.start + nodev.half, n_rows = nodev.size - nodev.half;
  const bool xlds = nd == 1;
  bool cm[E], rm[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { cm[e] = lane + 64 * e < n_cols; rm[e] = lane + 64 * e < n_rows; }
  if (xlds) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (rm[e]) xs[lane + 64 * e] = x[row0 + lane + 64 * e];
      if (cm[e]) xs[MR + lane + 64 * e] = x[col0 + lane + 64 * e];
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) if (rm[e]) lidx[lane + 64 * e] = (unsigned short)(lane + 64 * e);
  auto xrow = [&](int m) -> const double* { return xlds ? (const double*)(xs + m) : x + (long)(row0 + m) * nd; };
  auto xcol = [&](int n) -> const double* { return xlds ? (const double*)(xs + MR + n) : x + (long)(col0 + n) * nd; };
  auto kval = [&](const double* a, const double* b) -> double { return FAST ? gh_fast_value(fast, a, b) : gh_eval_value(prog, n_prog, a, b); };
  const int full_rank = n_rows < n_cols ? n_rows : n_cols;
  int max_rank = full_rank;
  if (max_rank > rcap) max_rank = rcap;
  int remaining = n_rows, rank = 0;
  double norm = 0.0;
  const double tol2 = tol * tol;
  bool converged = false;
  unsigned long long st = seed ^ ((unsigned long long)(level + 1) << 40) ^ ((unsigned long long)(node + nodev.pad) * 0x9E3779B97F4A7C15ull);
  typedef typename AwVec<E>::type VE;
  VE U[AW_RW], V[AW_RW];
#pragma unroll
  for (int k = 0; k < AW_RW; ++k) { U[k] = (VE)(0.0); V[k] = (VE)(0.0); }
  __builtin_amdgcn_s_waitcnt(0);                        // (the wavefront's own LDS writes above)
  __builtin_amdgcn_wave_barrier();
  while (rank < max_rank) {
    if (rank >= AW_RW) {                                // more columns than the mirrors hold: the level goes to the workgroup kernel
      if (lane == 0) atomicMax(trunc, 2);
      return;
    }
    // ---- a random unused row with a non-negligible residual (hodlr.h:159-191): candidates one by one, in drawing order
    bool got = false;
    VE v = (VE)(0.0);
    while (remaining > 0) {
      st += 0x9E3779B97F4A7C15ull;
      unsigned long long z = st;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      const int kk = (int)(z % (unsigned long long)remaining);
      const int i = __builtin_amdgcn_readfirstlane((int)lidx[kk]);
      __builtin_amdgcn_wave_barrier();                  // (every lane has read lidx[kk] before lane 0 overwrites it)
      if (lane == 0) lidx[kk] = lidx[remaining - 1];
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      --remaining;
      const int si = i >> 6, li = i & 63;
      double cw[AW_RW];
#pragma unroll
      for (int k = 0; k < AW_RW; ++k) cw[k] = (k < rank) ? __shfl(aw_pick<E>(U[k], si), li, 64) : 0.0;       // U(i, k)
      const double* xi = xrow(i);
      bool hit = false;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        double t = cm[e] ? kval(xi, xcol(lane + 64 * e)) : 0.0;
#pragma unroll
        for (int k = 0; k < AW_RW; ++k) if (k < rank) t -= cw[k] * V[k][e];
        v[e] = t;
        hit = hit || (cm[e] && fabs(t) >= 1e-14);       // hodlr.h:191 on the row's largest entry
      }
      if (__any(hit)) { got = true; break; }
    }
    if (!got) { converged = true; break; }              // rows exhausted (see hodlr_aca_kernel)
    // ---- pivot: largest |entry|, smallest column on ties
    int j;
    double pivot;
    {
      double best = -1.0;
      int bestn = -1;
#pragma unroll
      for (int e = 0; e < E; ++e) { const double a = cm[e] ? fabs(v[e]) : -1.0; if (cm[e] && a > best) { best = a; bestn = lane + 64 * e; } }
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bestn, off, 64);
        if (ov > best || (ov == best && oi >= 0 && (bestn < 0 || oi < bestn))) { best = ov; bestn = oi; }
      }
      j = __builtin_amdgcn_readfirstlane(bestn);
      pivot = __shfl(aw_pick<E>(v, j >> 6), j & 63, 64);
    }
    // ---- normalise the row by its pivot, build the column (hodlr.h:194-199)
    VE vn = (VE)(0.0), un = (VE)(0.0), u = (VE)(0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (cm[e]) { v[e] = v[e] / pivot; Tcm[(long)rank * N + col0 + lane + 64 * e] = v[e]; vn[e] = v[e] * v[e]; }
    }
    double cv[AW_RW];
#pragma unroll
    for (int k = 0; k < AW_RW; ++k) cv[k] = (k < rank) ? __shfl(aw_pick<E>(V[k], j >> 6), j & 63, 64) : 0.0;   // V(j, k)
    const double* xj = xcol(j);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      double t = rm[e] ? kval(xrow(lane + 64 * e), xj) : 0.0;
#pragma unroll
      for (int k = 0; k < AW_RW; ++k) if (k < rank) t -= cv[k] * U[k][e];
      u[e] = t;
      if (rm[e]) { Tcm[(long)rank * N + row0 + lane + 64 * e] = t; un[e] = t * t; }
    }
    // (column `rank` of the mirrors: selects with constant register indices -- written as `if (k == rank) V[k][e] = ...` the
    //  compiler turned the chain back into V[rank][e], a run-time index, and moved both arrays to scratch memory)
#pragma unroll
    for (int k = 0; k < AW_RW; ++k) {
      const bool here = (k == rank);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        V[k][e] = here ? (cm[e] ? v[e] : 0.0) : V[k][e];
        U[k][e] = here ? (rm[e] ? u[e] : 0.0) : U[k][e];
      }
    }
    ++rank;
    if (rank >= full_rank) { converged = true; break; }                                // hodlr.h:203
    if (rank >= max_rank) break;                                                       // rank cap: NOT converged
    const double un2 = aw_sum<E>(un), vn2 = aw_sum<E>(vn);
    // cross terms |u_new . u_k|, |v_new . v_k|, k < rank - 1 (hodlr.h:210-214)
    double maxu = 0.0, maxv = 0.0;
#pragma unroll
    for (int k = 0; k < AW_RW - 1; ++k)
      if (k < rank - 1) {
        VE pu = (VE)(0.0), pv = (VE)(0.0);
#pragma unroll
        for (int e = 0; e < E; ++e) { pu[e] = rm[e] ? U[k][e] * u[e] : 0.0; pv[e] = cm[e] ? V[k][e] * v[e] : 0.0; }
        const double a = aw_sum<E>(pu), b = aw_sum<E>(pv);
        if (fabs(a) > maxu) maxu = fabs(a);
        if (fabs(b) > maxv) maxv = fabs(b);
      }
    const double rowcol = un2 * vn2;
    if (rowcol < tol2 * norm) { converged = true; break; }                             // hodlr.h:206-207
    norm += rowcol;
    if (rank > 1) norm += 2.0 * maxu + 2.0 * maxv;
  }
  if (lane == 0) {
    ranks[node] = rank;
    if (!converged && rank < full_rank) atomicMax(trunc, 1);     // stopped by the caller's cap, not by the tolerance
  }
}

// 1 (default): the deep levels whose blocks have <= 256 rows and columns through hodlr_aca_wave_kernel; 0: every level through the
// workgroup kernel (A/B and the same-bits test)
static int g_hodlr_leaf_fused = 1;      // 128-row leaves of fast-form kernels: evaluated inside the factorisation kernel (0: a build launch first)
extern "C" int gh_debug_set_hodlr_leaf_fused(int on) {
  const int prev = g_hodlr_leaf_fused;
  g_hodlr_leaf_fused = on ? 1 : 0;
  return prev;
}
static int g_hodlr_coop_singles = 1;    // clusterable levels that end up with one workgroup per node ride at the end of the cooperative launch
extern "C" int gh_debug_set_hodlr_coop_singles(int on) {
  const int prev = g_hodlr_coop_singles;
  g_hodlr_coop_singles = on ? 1 : 0;
  return prev;
}
// The clusters BELOW the first clustered level get 1 / this of the workgroups the even-load rule deals them (never fewer than two).
// Even load per thread makes every cluster as fast as the root's -- but only the root's chain of ~20 ACA steps is the critical path
// of phase 1; the clusters below it finish earlier whatever they get, and every workgroup of a cluster holds its CU (registers:
// nothing else fits beside it) mostly waiting at cluster barriers.  Half as wide they take longer, still end before the root,
// and the CUs go to the one-workgroup nodes and the leaves: C4 3.48 -> 3.40 ms, 1 048 576 17.9 -> 17.4, never slower
// (profiles/r06/hodlr_coop_lower_ab.md; a quarter: 4.02 ms -- then they outlast the root).
static int g_hodlr_coop_lower = 2;
extern "C" int gh_debug_set_hodlr_coop_lower(int div) {
  const int prev = g_hodlr_coop_lower;
  g_hodlr_coop_lower = div < 1 ? 2 : div;
  return prev;
}
static int g_hodlr_u_from_v = 1;        // the factorisation's leaf product reads the level-major V and writes U for the first time (no U from the compaction)
extern "C" int gh_debug_set_hodlr_u_from_v(int on) {
  const int prev = g_hodlr_u_from_v;
  g_hodlr_u_from_v = on ? 1 : 0;
  return prev;
}
static int g_hodlr_lpt = 1;             // the one-workgroup ACA launch takes a level's nodes longest first (durations of the handle's previous compute())
extern "C" int gh_debug_set_hodlr_lpt(int on) {
  const int prev = g_hodlr_lpt;
  g_hodlr_lpt = on ? 1 : 0;
  return prev;
}
static int g_hodlr_coop_wgs = 256;      // workgroups of the cooperative ACA launch (<= CUs: every cluster resident)
extern "C" int gh_debug_set_hodlr_coop_wgs(int n) {
  const int prev = g_hodlr_coop_wgs;
  g_hodlr_coop_wgs = n < 32 ? 32 : (n > 256 ? 256 : n);
  return prev;
}
static int g_hodlr_wave_aca = 1;
extern "C" int gh_debug_set_hodlr_wave_aca(int on) {
  const int prev = g_hodlr_wave_aca;
  g_hodlr_wave_aca = on ? 1 : 0;
  return prev;
}
// (static + dynamic LDS of a launch with the mirrors is 67 KiB: above the 64 KiB a kernel gets without asking; per device)
static int aca_lds_attr() {
  static thread_local unsigned long long done = 0;
  int dev = 0;
  GH_HIP(hipGetDevice(&dev));
  if (dev < 64 && (done >> dev & 1ull)) return GH_OK;
  GH_HIP(hipFuncSetAttribute((const void*)hodlr_aca_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ACA_DYN_BYTES));
  GH_HIP(hipFuncSetAttribute((const void*)hodlr_aca_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ACA_DYN_BYTES));
  if (dev < 64) done |= 1ull << dev;
  return GH_OK;
}

// B (rows of this level's nodes x R, row-major, ld = R) <- first rank columns of Tcm, zero padded
// All levels in ONE launch (eleven launches of 10-38 us each at C4): workgroups [b0, b0 + nn * ny) belong to the
// level described by a segment; inside it, workgroup (node, y) as in hodlr_compact_kernel below.
struct CompactSeg { const double* Tcm; const LvlNode* nodes; const int* ranks; int R, ny; long off, offv, ldv; int b0, nblk; };
// (the segment table rides in the kernel arguments: uploading it was a copy from pageable memory -- staged and waited for -- between
//  the ranks' arrival on the host and this launch, on the critical path of every compute())
struct CompactSegs { int n; CompactSeg s[24]; };
__global__ void hodlr_compact_all_kernel(const CompactSegs segs, long N, double* UA, long ld, double* VA) {
  int q = 0;
  while (q + 1 < segs.n && (int)blockIdx.x >= segs.s[q].b0 + segs.s[q].nblk) ++q;
  const CompactSeg sg = segs.s[q];
  const int local = (int)blockIdx.x - sg.b0, node = local / sg.ny, by = local % sg.ny;
  const LvlNode nd = sg.nodes[node];
  const int rk = sg.ranks[node], R = sg.R;
  const long tot = (long)nd.size * R;
  for (long e = (long)by * blockDim.x + threadIdx.x; e < tot; e += (long)sg.ny * blockDim.x) {
    const int r = (int)(e / R), k = (int)(e % R);
    const long i = nd.start + r;
    const double v = (k < rk) ? sg.Tcm[(long)k * N + i] : 0.0;
    if (UA) UA[i * ld + sg.off + k] = v;          // (nullptr: the leaf product reads the level-major copy and writes U itself -- LeafSrc)
    VA[i * sg.ldv + sg.offv + k] = v;
  }
}
__global__ void hodlr_compact_kernel(const double* Tcm, long N, const LvlNode* nodes, const int* ranks,
                                     int R, double* UA, long ld, long off, double* VA, long ldv, long offv) {
  const LvlNode nd = nodes[blockIdx.x];
  const int rk = ranks[blockIdx.x];
  const long tot = (long)nd.size * R;
  // (element index split as (row, k), k fastest: the WRITES of a row's R columns are what must be
  //  coalesced -- with one thread per row and coalesced reads of the column-major Tcm the eleven
  //  launches took 558 instead of 185 us)
  for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.y * blockDim.x) {
    const int r = (int)(e / R), k = (int)(e % R);
    const long i = nd.start + r;
    const double v = (k < rk) ? Tcm[(long)k * N + i] : 0.0;
    UA[i * ld + off + k] = v;
    if (VA) VA[i * ldv + offv + k] = v;
  }
}
// UL (level-major) <- UA (row-major n x Rtot): column c of row i goes to UL[colbase[c] + i * colld[c]]
__global__ void hodlr_relayout_kernel(const double* UA, long n, int Rtot, const long* colbase, const int* colld, double* UL) {
  const long tot = n * Rtot;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const long i = e / Rtot;
    const int c = (int)(e % Rtot);
    UL[colbase[c] + i * colld[c]] = UA[e];
  }
}

// ======================================================================== leaves
// pitch == 0: leaf b is stored size x size at leaves[b].off; pitch > 0: in a pitch x pitch slot at
// b * pitch^2, identity-padded (the batched Cholesky path below wants 128 x 128 blocks)
__global__ void hodlr_leaf_build_kernel(const GhNode* __restrict__ prog, int n_prog, GhFast fast, int nd, const double* x,
                                        const double* yerr, const LeafDesc* leaves, double* Lf, int pitch) {
  const LeafDesc lf = leaves[blockIdx.x];
  if (pitch > 0) {
    // (round 5) the leaf's coordinates and noise through LDS first: with both points of every element fetched from global
    // memory inside the loop a thread's eight elements were eight dependent round trips -- 750-860 us for the 2048 leaves of
    // C4 (45 G elements/s; the dense build evaluates 700 G/s), the longest piece of the leaf chain and, through it, of the
    // whole first phase of a step.  Same evaluator, same ordered arguments: same bits.
    __shared__ double xs[256 * 8];
    __shared__ double es[256];
    const bool in_lds = nd <= 8 && lf.size <= 256;
    if (in_lds) {
      for (int t = threadIdx.x; t < lf.size * nd; t += blockDim.x) xs[t] = x[(long)lf.start * nd + t];
      for (int t = threadIdx.x; t < lf.size; t += blockDim.x) es[t] = yerr[lf.start + t];
      __syncthreads();
    }
    const double* const xb = in_lds ? (const double*)xs : x + (long)lf.start * nd;
    double* slot = Lf + (long)blockIdx.x * pitch * pitch;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < pitch * pitch; e += gridDim.y * blockDim.x) {
      const int r = e / pitch, c = e % pitch;
      double v = (r == c) ? 1.0 : 0.0;
      if (r < lf.size && c < lf.size) {
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        const double* pa = xb + (long)lo * nd;
        const double* pb = xb + (long)hi * nd;
        v = fast.ok ? gh_fast_value(fast, pa, pb) : gh_eval_value(prog, n_prog, pa, pb);
        if (r == c) { const double e2 = in_lds ? es[r] : yerr[lf.start + r]; v += e2 * e2; }
      }
      slot[e] = v;
    }
    return;
  }
  const long tot = (long)lf.size * lf.size;
  for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.y * blockDim.x) {
    const int r = (int)(e / lf.size), c = (int)(e % lf.size);
    const int lo = r < c ? r : c, hi = r < c ? c : r;        // ordered arguments: exactly symmetric
    const double* pa = x + (long)(lf.start + lo) * nd;
    const double* pb = x + (long)(lf.start + hi) * nd;
    double v = fast.ok ? gh_fast_value(fast, pa, pb) : gh_eval_value(prog, n_prog, pa, pb);
    if (r == c) { const double e2 = yerr[lf.start + r]; v += e2 * e2; }              // hodlr.h:125, _hodlr.cpp:76
    Lf[lf.off + e] = v;
  }
}

// out[b] = 2 * sum_i log L_ii of the b-th 128 x 128 factored slot (identity padding adds 0)
__global__ __launch_bounds__(128) void hodlr_leaf_logdet_kernel(const double* Lf, double* out) {
  __shared__ double sh[8];
  const double* slot = Lf + (long)blockIdx.x * 128 * 128;
  const double v = hw_block_sum(log(slot[threadIdx.x * 129]), sh);
  if (threadIdx.x == 0) out[blockIdx.x] = 2.0 * v;
}

// the same for a 256 x 256 slot factored as 2 x 2 blocks of 128 (both diagonal blocks hold their factors)
__global__ __launch_bounds__(256) void hodlr_leaf_logdet256_kernel(const double* Lf, double* out) {
  __shared__ double sh[8];
  const double* slot = Lf + (long)blockIdx.x * 256 * 256;
  const double v = hw_block_sum(log(slot[threadIdx.x * 257]), sh);
  if (threadIdx.x == 0) out[blockIdx.x] = 2.0 * v;
}

// ---- batched 128 x 128 products of the leaf stage on the dense solver's tile function (round 5)
// C_b (-)= A_b B_b^T for b < gridDim.x: A_b, B_b are 128 x K with k contiguous, one workgroup per product, the LDS-DMA /
// v_mfma_f64_16x16x4 tile of gh_gemm_tile.h.  The leaf stage's products went through hodlr_mm_kernel (a generic 32 x 64 x 32
// tile with register staging: 8-13 TFLOP/s -- 0.67 ms for the 2048 products K^-1 = L^-T L^-1 of C4, 1.1-2.2 ms for each of the
// nine products of 4096 leaves of 171 rows at N = 700000).
template <bool ACC>
__global__ __launch_bounds__(256, 2) void hodlr_bmm_nt_kernel(double* C, long ldc, long sc, const double* A, long lda, long sa,
                                                              const double* B, long ldb, long sb, long K) {
  __shared__ __attribute__((aligned(1024))) double sm[4 * BM * BK];
  const long b = blockIdx.x;
  gh_tile128_nt_sp<ACC>(sm, C + b * sc, ldc, A + b * sa, lda, B + b * sb, ldb, K);
}
// dst_b = src_b^T (128 x 128 each): blockIdx.y = one of the sixteen 32 x 32 tiles, through a padded LDS tile (8.4 KiB: the
// first form staged the whole block -- 132 KiB, one workgroup per CU, 633 us for the 2048 leaves of C4)
__global__ __launch_bounds__(256) void hodlr_transpose128_kernel(const double* src, long lds_, long ss, double* dst, long ldd, long sd) {
  __shared__ double t[32 * 33];
  const double* s = src + (long)blockIdx.x * ss;
  double* d = dst + (long)blockIdx.x * sd;
  const int tr = (blockIdx.y >> 2) * 32, tc = (blockIdx.y & 3) * 32;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) t[(r0 + 8 * q) * 33 + c] = s[(long)(tr + r0 + 8 * q) * lds_ + tc + c];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) d[(long)(tc + r0 + 8 * q) * ldd + tr + c] = t[c * 33 + r0 + 8 * q];
}

// Batched in-place inverse by Gauss-Jordan with partial (row) pivoting; one workgroup per matrix
// (row-major n x n at base + offs[b]).  logdet[b] = sum log|pivot|.  scratch: n doubles + n ints
// per matrix at sc_off[b].  When the launch provides dynamic LDS (`lds_doubles` >= n * (n|1) + n) the
// matrix is worked on in LDS with an odd row pitch and written back once: the n rank-1 updates
// are then n^3 LDS accesses instead of n^3 L2 round trips (2048 leaves of 128 x 128 at C4:
// 12.9 ms -> well under 1 ms).  Larger matrices (leaves can reach 2 min_size - 1 rows) fall back
// to working in place in HBM.
extern __shared__ double gj_lds[];
__global__ __launch_bounds__(256) void gj_inverse_kernel(double* base, const long* offs, const int* sizes,
                                                         double* scratch_d, int* scratch_i, const long* sc_off,
                                                         double* logdet, int* fail, int lds_doubles) {
  __shared__ double shd[4];
  __shared__ int shi[4];
  const int b = blockIdx.x, n = sizes[b], tid = threadIdx.x;
  double* M = base + offs[b];
  int* piv = scratch_i + sc_off[b];
  const int lane = tid & 63, wave = tid >> 6;
  const bool in_lds = (long)n * (n | 1) + n <= (long)lds_doubles;
  double* W = in_lds ? gj_lds : M;
  double* fcol = in_lds ? gj_lds + (long)n * (n | 1) : scratch_d + sc_off[b];
  const int P = in_lds ? (n | 1) : n;
  if (in_lds) {
    for (int i = wave; i < n; i += 4)
      for (int c = lane; c < n; c += 64) W[(long)i * P + c] = M[(long)i * n + c];
    __syncthreads();
  }
  double ld = 0.0;
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    double best = -1.0;
    int bi = -1;
    for (int i = k + tid; i < n; i += 256) {
      const double a = fabs(W[(long)i * P + k]);
      if (a > best) { best = a; bi = i; }
    }
    hw_block_argmax(best, bi, shd, shi);
    const int p = bi;
    if (!(best > 0.0) || p < 0) { bad = true; break; }     // singular or NaN (uniform)
    if (tid == 0) piv[k] = p;
    if (p != k)
      for (int c = tid; c < n; c += 256) {
        const double t = W[(long)k * P + c];
        W[(long)k * P + c] = W[(long)p * P + c];
        W[(long)p * P + c] = t;
      }
    __syncthreads();
    const double pv = W[(long)k * P + k];
    ld += log(fabs(pv));
    for (int i = tid; i < n; i += 256) fcol[i] = W[(long)i * P + k];
    __syncthreads();
    for (int c = tid; c < n; c += 256) W[(long)k * P + c] = ((c == k) ? 1.0 : W[(long)k * P + c]) / pv;
    for (int i = tid; i < n; i += 256) if (i != k) W[(long)i * P + k] = 0.0;
    __syncthreads();
    // rank-1 update, one wavefront per row (no index divisions; 512-byte row segments per access),
    // the pivot row held in registers
    for (int c0 = 0; c0 < n; c0 += 256) {
      const int c_lo = c0 + lane;
      double rk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rk[q] = (c_lo + 64 * q < n) ? W[(long)k * P + c_lo + 64 * q] : 0.0;
      for (int i = wave; i < n; i += 4) {
        if (i == k) continue;
        const double fi = fcol[i];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c_lo + 64 * q < n) W[(long)i * P + c_lo + 64 * q] -= fi * rk[q];
      }
    }
    __syncthreads();
  }
  if (bad) {
    if (tid == 0) { atomicExch(fail, b + 1); logdet[b] = 0.0; }
    return;
  }
  for (int k = n - 1; k >= 0; --k) {                        // undo the row swaps on the columns
    const int p = piv[k];
    if (p != k)
      for (int r = tid; r < n; r += 256) {
        const double t = W[(long)r * P + k];
        W[(long)r * P + k] = W[(long)r * P + p];
        W[(long)r * P + p] = t;
      }
    __syncthreads();
  }
  if (in_lds)
    for (int i = wave; i < n; i += 4)
      for (int c = lane; c < n; c += 64) M[(long)i * n + c] = W[(long)i * P + c];
  if (tid == 0) logdet[b] = ld;
}

// ---- the same inverse for SMALL matrices (n <= NMAX <= 32: the 2r x 2r Woodbury cores), one WAVEFRONT per
// matrix, no barriers, no LDS: lane i holds row i in registers.  gj_inverse_kernel above is built for leaves
// of up to 256 rows -- five workgroup barriers, a block-wide arg-max through LDS and a global write per
// pivot: 3 us per pivot step, 91 us for the 30 x 30 core of the root node, 0.42 ms over the eleven levels
// of C4.  Here a pivot step is a DPP row reduction for the pivot search (on the high words of |a|: as
// unsigned integers they order like the doubles), 2 n v_readlane for the pivot row and n FMAs.
// Rows are not swapped: pivot k is found among the rows not yet used and stays where it is (row p_k plays
// row k of P A); (P A)^-1 = A^-1 P^T, so lane p_r ends with A^-1[r][p_c] in column register c.
__device__ __forceinline__ double gj_bcast(double v, int src_lane) {         // src_lane: wave-uniform, not a constant
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned gj_row_max_u32(unsigned v) {            // max over the lane's 16-lane row, in every lane
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}
template <int NMAX>
// tsum != nullptr: the matrix is not read from `base` but built on the way in as the Woodbury core
// S = [[I, V1^T U1], [V0^T U0, I]] (hodlr.h:229-232) from rows [2 R b, 2 R b + 2 R) x columns [0, R) of tsum
// (pitch Cp) -- what hodlr_sbuild_kernel did in a launch of its own, eleven times per compute().
__global__ __launch_bounds__(256) void gj_small_kernel(double* base, const long* offs, const int* sizes, int nb,
                                                       double* logdet, int* fail, const double* tsum, long Cp, int R) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;                            // (the whole wavefront)
  const int n = __builtin_amdgcn_readfirstlane(sizes[b]);
  double* const M = base + offs[b];
  const bool row = lane < n;
  double m[NMAX];
  if (tsum) {
    const double* const tr = tsum + ((long)b * n + lane) * Cp;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      double v = (lane == c) ? 1.0 : 0.0;
      if (row && c < n) {
        if (lane < R && c >= R) v = tr[c - R];
        else if (lane >= R && c < R) v = tr[c];
      }
      m[c] = (row && c < n) ? v : 0.0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < NMAX; ++c) m[c] = (row && c < n) ? M[(long)lane * n + c] : 0.0;
  }
  bool used = false, bad = false;
  int myk = 0, pc[NMAX];
  // (round 6) log|det| = sum_k log|pivot_k| in pivot order -- but not INSIDE the pivot loop, where every lane evaluated the same
  // f64 logarithm (~150 instructions) on the critical path of each of the n dependent steps: lane k keeps pivot k, all lanes take
  // their logarithm at once after the loop, lane 0 adds them in the same order (the same bits)
  double mypiv = 1.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < n && !bad) {                          // (uniform)
      const bool cand = row && !used;
      const unsigned key = cand ? (unsigned)__double2hiint(fabs(m[k])) : 0u;
      const unsigned rm = gj_row_max_u32(key);
      const unsigned mx = max((unsigned)__builtin_amdgcn_readlane((int)rm, 0), (unsigned)__builtin_amdgcn_readlane((int)rm, 16));   // n <= 32: two rows
      const unsigned long long who = __builtin_amdgcn_ballot_w64(cand && key == mx);
      const int p = (int)__builtin_ctzll(who | (1ull << 63));
      const double pv = gj_bcast(m[k], p);
      if (who == 0ull || !(fabs(pv) > 0.0)) { bad = true; }
      else {
        pc[k] = p;
        if (lane == p) { used = true; myk = k; }
        if (lane == k) mypiv = pv;
        const double rinv = 1.0 / pv;
        const double f = m[k];
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {
          if (c != k && c < n) {
            const double pr = gj_bcast(m[c], p) * rinv;
            m[c] = (lane == p) ? pr : fma(-f, pr, m[c]);
          }
        }
        m[k] = (lane == p) ? rinv : -f * rinv;
      }
    }
  }
  if (bad) {
    if (lane == 0) { atomicExch(fail, b + 1); logdet[b] = 0.0; }
    return;
  }
  if (row) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c < n) M[(long)myk * n + pc[c]] = m[c];
  }
  {
    const double lg = log(fabs(mypiv));
    double ld = 0.0;
    for (int k = 0; k < n; ++k) ld += gj_bcast(lg, k);
    if (lane == 0) logdet[b] = ld;
  }
}

// The same inverse with ONE WORKGROUP per matrix, its columns dealt over the four wavefronts (column c on wavefront c & 3): for
// the levels with few nodes, where gj_small_kernel leaves one wavefront to walk all n pivots x n columns alone -- 42-46 us for each
// of the three 24..30-row cores at the top of C4's tree, 13-17 us for the 10..14-row ones below them.  The owner of column k
// finds the pivot and puts the column and the pivot row's index in LDS (double-buffered: one barrier per pivot); every
// wavefront then updates its own columns with exactly gj_small_kernel's operations -- a column's arithmetic does not depend
// on which wavefront holds it: the same bits.
template <int NMAX>
__global__ __launch_bounds__(256) void gj_small4_kernel(double* base, const long* offs, const int* sizes, int nb,
                                                        double* logdet, int* fail, const double* tsum, long Cp, int R) {
  constexpr int CW = NMAX / 4;
  __shared__ double fcol[2][64];
  __shared__ int sp[2];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x;
  const int n = __builtin_amdgcn_readfirstlane(sizes[b]);
  double* const M = base + offs[b];
  const bool row = lane < n;
  double m[CW];
  if (tsum) {
    const double* const tr = tsum + ((long)b * n + lane) * Cp;
#pragma unroll
    for (int q = 0; q < CW; ++q) {
      const int c = 4 * q + w;
      double v = (lane == c) ? 1.0 : 0.0;
      if (row && c < n) {
        if (lane < R && c >= R) v = tr[c - R];
        else if (lane >= R && c < R) v = tr[c];
      }
      m[q] = (row && c < n) ? v : 0.0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < CW; ++q) { const int c = 4 * q + w; m[q] = (row && c < n) ? M[(long)lane * n + c] : 0.0; }
  }
  bool used = false, bad = false;
  int myk = 0, pcw[CW];
  double mypiv = 1.0;                             // (lane k keeps pivot k: the logarithms are taken after the loop -- see gj_small_kernel)
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < n && !bad) {                          // (uniform over the workgroup: `bad` is set by all wavefronts together)
      const int buf = k & 1, own = k & 3, qk = k >> 2;
      if (w == own) {
        const bool cand = row && !used;
        const unsigned key = cand ? (unsigned)__double2hiint(fabs(m[qk])) : 0u;
        const unsigned rm = gj_row_max_u32(key);
        const unsigned mx = max((unsigned)__builtin_amdgcn_readlane((int)rm, 0), (unsigned)__builtin_amdgcn_readlane((int)rm, 16));
        const unsigned long long who = __builtin_amdgcn_ballot_w64(cand && key == mx);
        const int p = (int)__builtin_ctzll(who | (1ull << 63));
        const double pv = gj_bcast(m[qk], p);
        fcol[buf][lane] = m[qk];
        if (lane == 0) sp[buf] = (who == 0ull || !(fabs(pv) > 0.0)) ? -1 : p;
      }
      __syncthreads();
      const int p = __builtin_amdgcn_readfirstlane(sp[buf]);
      if (p < 0) { bad = true; }
      else {
        const double f = fcol[buf][lane];
        const double pv = fcol[buf][p];
        if (lane == p) { used = true; myk = k; }
        if (lane == k) mypiv = pv;
        const double rinv = 1.0 / pv;
#pragma unroll
        for (int q = 0; q < CW; ++q) {
          const int c = 4 * q + w;
          if (c == k) pcw[q] = p;
          if (c != k && c < n) {
            const double pr = gj_bcast(m[q], p) * rinv;
            m[q] = (lane == p) ? pr : fma(-f, pr, m[q]);
          }
        }
        if (w == own) m[qk] = (lane == p) ? rinv : -f * rinv;
      }
    }
  }
  if (bad) {
    if (threadIdx.x == 0) { atomicExch(fail, b + 1); logdet[b] = 0.0; }
    return;
  }
  if (row) {
#pragma unroll
    for (int q = 0; q < CW; ++q) {
      const int c = 4 * q + w;
      if (c < n) M[(long)myk * n + pcw[q]] = m[q];
    }
  }
  if (w == 0) {
    const double lg = log(fabs(mypiv));
    double ld = 0.0;
    for (int k = 0; k < n; ++k) ld += gj_bcast(lg, k);
    if (lane == 0) logdet[b] = ld;
  }
}

// =============================================================== batched small dense products
// O(job rows, 0:C) (=|-=) A_job (m x kd) * B(job rows, 0:C); A element (r, k) at
// A[a_off + r*a_rs + k*a_cs]; B row b_row+k at B[(b_row+k)*ldb + b_col0 + c].
struct MMArgs {
  const MMJob* jobs;
  const double* A; long a_rs, a_cs;
  const double* B; long ldb, b_col0;
  double* O; long ldo, o_col0;
  int C, subtract, mtiles;
};
__global__ __launch_bounds__(256) void hodlr_mm_kernel(MMArgs a) {
  __shared__ double As[32 * 33];
  __shared__ double Bs[32 * 64];
  const MMJob job = a.jobs[blockIdx.x];
  const int c0 = blockIdx.z * 64, tid = threadIdx.x;
  // 32 x 64 tile on the matrix pipe: wavefront w takes the 16-row block w & 1 and the two 16-column
  // blocks 2 (w >> 1), 2 (w >> 1) + 1; operands are staged in LDS exactly as for the VALU loop this
  // replaced (8 FMAs per staged element and lane -> 2 MFMAs per 4 k)
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int bi = wave & 1, bj = 2 * (wave >> 1);
  typedef double mm_v4d __attribute__((ext_vector_type(4)));
  const bool rfast = (a.a_rs == 1);
  for (int mt = 0; mt < a.mtiles; ++mt) {
    const int m0 = (blockIdx.y * a.mtiles + mt) * 32;
    if (m0 >= job.m) break;                                   // (uniform)
    mm_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < job.kd; k0 += 32) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const int r = rfast ? (e & 31) : (e >> 5), k = rfast ? (e >> 5) : (e & 31);
        double v = 0.0;
        if (m0 + r < job.m && k0 + k < job.kd) v = a.A[job.a_off + (long)(m0 + r) * a.a_rs + (long)(k0 + k) * a.a_cs];
        As[r * 33 + k] = v;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = tid + 256 * q;
        const int k = e >> 6, cc = e & 63;
        double v = 0.0;
        if (k0 + k < job.kd && c0 + cc < a.C) v = a.B[(long)(job.b_row + k0 + k) * a.ldb + a.b_col0 + c0 + cc];
        Bs[k * 64 + cc] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const double av = As[(16 * bi + fr) * 33 + 4 * kk + fk];
        const double b0 = Bs[(4 * kk + fk) * 64 + 16 * bj + fr];
        const double b1 = Bs[(4 * kk + fk) * 64 + 16 * bj + 16 + fr];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc1, 0, 0, 0);
      }
      __syncthreads();
    }
    // f64 MFMA C/D map: row = (lane >> 4) + 4 reg, col = lane & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * bi + fk + 4 * r;
      if (row >= job.m) continue;
      double* o = a.O + (long)(job.o_row + row) * a.ldo + a.o_col0 + c0 + 16 * bj + fr;
      if (c0 + 16 * bj + fr < a.C) o[0] = a.subtract ? (o[0] - acc0[r]) : acc0[r];
      if (c0 + 16 * bj + 16 + fr < a.C) o[16] = a.subtract ? (o[16] - acc1[r]) : acc1[r];
    }
  }
}
// ------------------------------------------------------------ narrow right-hand sides (C <= 8)
// The tile kernel above does a full 32 x 64 x 32 block of matrix-pipe work per staged slab whatever
// the real extents: for ONE right-hand side (every log-likelihood evaluation) 63 of its 64 columns
// are padding, and a level pass of a solve cost ~50 us for a few MFLOP.  These do the same three
// steps with plain FMAs on exactly the data there is.
#define MV_C 8
// P[(o_row + r) * Cp + c] = sum_k V(r, k) X(b_row + k, c);  V(r, k) at A[a_off + r + k * R]  (level-major V block)
__global__ __launch_bounds__(256) void hodlr_mv_reduce_kernel(const MMJob* jobs, const double* A, int R, const double* X, long ldx,
                                                              long xcol0, double* P, long Cp, int C) {
  __shared__ double part[8][32][MV_C];
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, r = tid & 31, pt = tid >> 5;
  for (int r0 = 0; r0 < R; r0 += 32) {
    double acc[MV_C];
#pragma unroll
    for (int c = 0; c < MV_C; ++c) acc[c] = 0.0;
    if (r0 + r < R) {
      for (int k = pt; k < job.kd; k += 8) {
        const double v = A[job.a_off + (long)k * R + r0 + r];
        const double* xr = X + (long)(job.b_row + k) * ldx + xcol0;
#pragma unroll
        for (int c = 0; c < MV_C; ++c) if (c < C) acc[c] += v * xr[c];
      }
    }
#pragma unroll
    for (int c = 0; c < MV_C; ++c) part[pt][r][c] = acc[c];
    __syncthreads();
    if (pt == 0 && r0 + r < R) {
      for (int c = 0; c < C; ++c) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += part[q][r][c];
        P[(long)(job.o_row + r0 + r) * Cp + c] = v;
      }
    }
    __syncthreads();
  }
}
// X(o_row + i, c) -= sum_k U(i, k) T(b_row + k, c);  U(i, k) at A[a_off + i * a_rs + k], i < m, k < kd <= 32
__global__ __launch_bounds__(128) void hodlr_mv_update_kernel(const MMJob* jobs, const double* A, long a_rs, const double* T, long Cp,
                                                              double* X, long ldx, long xcol0, int C) {
  __shared__ double ts[32 * MV_C];
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  for (int e = tid; e < job.kd * C; e += 128) ts[(e / C) * MV_C + (e % C)] = T[(long)(job.b_row + e / C) * Cp + (e % C)];
  __syncthreads();
  if (tid >= job.m) return;
  double acc[MV_C];
#pragma unroll
  for (int c = 0; c < MV_C; ++c) acc[c] = 0.0;
  const double* ur = A + job.a_off + (long)tid * a_rs;
  for (int k = 0; k < job.kd; ++k) {
    const double u = ur[k];
#pragma unroll
    for (int c = 0; c < MV_C; ++c) acc[c] += u * ts[k * MV_C + c];
  }
  double* xr = X + (long)(job.o_row + tid) * ldx + xcol0;
#pragma unroll
  for (int c = 0; c < MV_C; ++c) if (c < C) xr[c] -= acc[c];
}
// X rows of leaf b <- Kinv_b X rows (in place: the leaf's rows are staged in LDS first); leaf size <= 256
__global__ __launch_bounds__(256) void hodlr_mv_leaf_kernel(const MMJob* jobs, const double* Kinv, long pitch, double* X, long ldx,
                                                            long xcol0, int C) {
  __shared__ double xs[256 * MV_C];
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = job.m;
  for (int e = tid; e < n * C; e += 256) xs[(e / C) * MV_C + (e % C)] = X[(long)(job.b_row + e / C) * ldx + xcol0 + (e % C)];
  __syncthreads();
  for (int i = wave; i < n; i += 4) {
    const double* row = Kinv + job.a_off + (long)i * pitch;
    double acc[MV_C];
#pragma unroll
    for (int c = 0; c < MV_C; ++c) acc[c] = 0.0;
    for (int k = lane; k < n; k += 64) {
      const double a = row[k];
#pragma unroll
      for (int c = 0; c < MV_C; ++c) acc[c] += a * xs[k * MV_C + c];
    }
#pragma unroll
    for (int c = 0; c < MV_C; ++c) {
      if (c < C) {
        const double v = hw_wave_sum(acc[c]);
        if (lane == 0) X[(long)(job.o_row + i) * ldx + xcol0 + c] = v;
      }
    }
  }
}
// Tsum for narrow right-hand sides: 32 threads per column each add a contiguous slice of the chunks,
// one thread then adds the 32 slice sums in order (the serial walk over up to N/256 chunks by a
// single active lane took 55-60 us at the top levels)
__global__ __launch_bounds__(256) void hodlr_sum_narrow_kernel(const double* P, const int* crange, int R, long Cp, int C, double* Tsum) {
  __shared__ double sl[32][MV_C];
  const int node = blockIdx.x, row = blockIdx.y;
  const int half = row < R ? 1 : 0, k = row < R ? row : row - R;
  const int cb = crange[(node * 2 + half) * 2], ce = crange[(node * 2 + half) * 2 + 1];
  const int c = threadIdx.x & 7, sidx = threadIdx.x >> 3;
  const int per = (ce - cb + 31) / 32;
  const int lo = cb + sidx * per, hi = lo + per < ce ? lo + per : ce;
  double v = 0.0;
  if (c < C) for (int ch = lo; ch < hi; ++ch) v += P[((long)ch * R + k) * Cp + c];
  sl[sidx][c] = v;
  __syncthreads();
  if (threadIdx.x < C) {
    double t = 0.0;
    for (int q = 0; q < 32; ++q) t += sl[q][threadIdx.x];
    Tsum[((long)node * 2 * R + row) * Cp + threadIdx.x] = t;
  }
}
// ---- round 5: the narrow solve (C <= 8 right-hand sides: every log-likelihood) in fewer, better-shaped launches.
// Round 4's solve of C4 (N = 262144, one right-hand side) was 45 launches, 0.57 ms: the leaf kernel walked 32 rows per
// wavefront with one exposed HBM round trip each (110 us for 268 MB), the per-chunk reduce kept R of every 32 lanes busy in a
// serial walk over the chunk's rows (21 us per level for 8 MB), and each level paid four dispatches.
//
// X rows of a leaf <- K_leaf^-1 X rows, thread = OUTPUT ROW: K^-1 is symmetric, so row i of the product is the sum over k of
// column i of row k -- every load of a wavefront is one contiguous 512-byte piece of row k, no reduction across lanes, and
// eight rows' loads are in flight per thread.  The rows k are split over 256 / (padded leaf size) thread groups whose partial
// sums are added in a fixed order.  (K^-1 = L^-T L^-1 is symmetric up to rounding: its (i, k) and (k, i) entries may differ
// in the last bit, as may the sum order from the row form -- the results agree to rounding, tests/test_gpu_hodlr.py.)
__global__ __launch_bounds__(256) void hodlr_mv_leaf_sym_kernel(const MMJob* jobs, const double* Kinv, long pitch, double* X, long ldx,
                                                                long xcol0, int C) {
  __shared__ double xs[256 * MV_C];
  __shared__ double part[256 * MV_C];
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, n = job.m;
  for (int e = tid; e < n * C; e += 256) xs[(e / C) * MV_C + (e % C)] = X[(long)(job.b_row + e / C) * ldx + xcol0 + (e % C)];
  __syncthreads();
  const int S = n <= 64 ? 4 : (n <= 128 ? 2 : 1), per_row = 256 / S;
  const int i = tid % per_row, sidx = tid / per_row;
  const int kper = (n + S - 1) / S, k_lo = sidx * kper, k_hi = min(n, k_lo + kper);
  double acc[MV_C];
#pragma unroll
  for (int c = 0; c < MV_C; ++c) acc[c] = 0.0;
  if (i < n) {
    const double* col = Kinv + job.a_off + i;
    int k = k_lo;
    for (; k + 8 <= k_hi; k += 8) {
      double a[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = col[(long)(k + q) * pitch];
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int c = 0; c < MV_C; ++c) if (c < C) acc[c] += a[q] * xs[(k + q) * MV_C + c];
    }
    for (; k < k_hi; ++k) {
      const double a = col[(long)k * pitch];
#pragma unroll
      for (int c = 0; c < MV_C; ++c) if (c < C) acc[c] += a * xs[k * MV_C + c];
    }
  }
#pragma unroll
  for (int c = 0; c < MV_C; ++c) part[tid * MV_C + c] = acc[c];
  __syncthreads();
  if (sidx == 0 && i < n) {
    for (int c = 0; c < C; ++c) {
      double v = part[i * MV_C + c];
      for (int q = 1; q < S; ++q) v += part[(q * per_row + i) * MV_C + c];
      X[(long)(job.o_row + i) * ldx + xcol0 + c] = v;
    }
  }
}
// One pass over the rows of a chunk for TWO neighbouring levels of the sweep (either part may be absent):
//   update (level l):   X(rows, c) -= sum_k U_l(row, k) T_l(b_row + k, c)                  [hodlr_mv_update_kernel]
//   reduce (level l'):  P[(o_row + r) Cp + c] = sum_rows V_l'(row, r) X(row, c)             [hodlr_mv_reduce_kernel]
// l' is the next shallower level with a positive rank and the SAME chunks (HLevel::chunk_geom): what the reduce reads is what
// the update has just written, kept in LDS.  Reduce: wavefront w takes the columns r = w, w + 4, ... of V, lane = row (and
// row + 64), one wavefront sum per (r, c) -- all lanes busy whatever R is, every load issued before the first sum.
__global__ __launch_bounds__(256) void hodlr_mv_updred_kernel(const MMJob* ujobs, const double* U, long u_rs, const double* T,
                                                              const MMJob* rjobs, const double* V, int R2, double* P,
                                                              long Cp, double* X, long ldx, long xcol0, int C) {
  __shared__ double ts[32 * MV_C];
  __shared__ double xs[128 * MV_C];
  extern __shared__ double us[];                     // [128][up]: the chunk's rows of U_l; up = 17 or 33 (the launch sizes it: 17 KiB
                                                     // instead of 33 lets all 2048 workgroups of a C4 level be resident at once)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int row0, m;
  // the reduce's V values first (they do not depend on the update): wavefront w, columns r = w + 4 q, rows lane and lane + 64
  double v0[8], v1[8];
  MMJob rj = {0, 0, 0, 0, 0};
  if (rjobs) {
    rj = rjobs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = wave + 4 * q;
      v0[q] = (r < R2 && lane < rj.kd) ? V[rj.a_off + (long)lane * R2 + r] : 0.0;
      v1[q] = (r < R2 && lane + 64 < rj.kd) ? V[rj.a_off + (long)(lane + 64) * R2 + r] : 0.0;
    }
  }
  if (ujobs) {
    const MMJob job = ujobs[blockIdx.x];
    row0 = job.o_row; m = job.m;
    double xold[MV_C];                                // (requested with everything else: one round trip for the whole update)
    if (tid < m) {
#pragma unroll
      for (int c = 0; c < MV_C; ++c) xold[c] = c < C ? X[(long)(row0 + tid) * ldx + xcol0 + c] : 0.0;
    }
    for (int e = tid; e < job.kd * C; e += 256) ts[(e / C) * MV_C + (e % C)] = T[(long)(job.b_row + e / C) * Cp + (e % C)];
    // the chunk's rows of U_l through LDS, all 256 threads, element e = (row, k) with k fastest: consecutive lanes read the kd
    // contiguous doubles of a row, then the next row (U is row-major with pitch u_rs here: one thread per row reading its kd
    // values in turn was 64 scattered 8-byte requests per load instruction)
    const int kd = job.kd, up = kd <= 16 ? 17 : 33;
    for (int e0 = tid; e0 < m * kd; e0 += 8 * 256) {           // eight loads in flight per thread (a rolled loop waits for each in turn)
      double uv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + 256 * q, r_ = e / kd, k_ = e - r_ * kd;
        uv[q] = e < m * kd ? U[job.a_off + (long)r_ * u_rs + k_] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + 256 * q, r_ = e / kd, k_ = e - r_ * kd;
        if (e < m * kd) us[r_ * up + k_] = uv[q];
      }
    }
    __syncthreads();
    if (tid < m) {
      double acc[MV_C];
#pragma unroll
      for (int c = 0; c < MV_C; ++c) acc[c] = 0.0;
      for (int k = 0; k < kd; ++k) {
        const double u = us[tid * up + k];
#pragma unroll
        for (int c = 0; c < MV_C; ++c) acc[c] += u * ts[k * MV_C + c];
      }
      double* xr = X + (long)(row0 + tid) * ldx + xcol0;
#pragma unroll
      for (int c = 0; c < MV_C; ++c)
        if (c < C) { const double v = xold[c] - acc[c]; xr[c] = v; xs[tid * MV_C + c] = v; }
    }
  } else {
    row0 = rj.b_row; m = rj.kd;
    for (int e = tid; e < m * C; e += 256) xs[(e / C) * MV_C + (e % C)] = X[(long)(row0 + e / C) * ldx + xcol0 + (e % C)];
  }
  if (!rjobs) return;
  __syncthreads();
  const bool k0 = lane < m, k1 = lane + 64 < m;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = wave + 4 * q;
    if (r >= R2) break;                              // (uniform)
    for (int c = 0; c < C; ++c) {
      double t = v0[q] * (k0 ? xs[lane * MV_C + c] : 0.0);
      t += v1[q] * (k1 ? xs[(lane + 64) * MV_C + c] : 0.0);
      t = hw_wave_sum(t);
      if (lane == 0) P[(long)(rj.o_row + r) * Cp + c] = t;
    }
  }
}
// Per node: Tsum = the chunk partials of each half added up (as hodlr_sum_narrow_kernel: 32 slices, then the slice sums in
// order), then Tout = S^-1 Tsum, the 2R x 2R core inverse times 2R x C, in the same workgroup (hodlr.h:247-252).
__global__ __launch_bounds__(256) void hodlr_mv_summm_kernel(const double* P, const int* crange, int R, long Cp, int C, const double* Sinv,
                                                             double* Tout) {
  __shared__ double sl[32][MV_C];
  __shared__ double tsum[64 * MV_C];
  const int node = blockIdx.x, n2 = 2 * R;
  const int c = threadIdx.x & 7, sidx = threadIdx.x >> 3;
  for (int row = 0; row < n2; ++row) {
    const int half = row < R ? 1 : 0, k = row < R ? row : row - R;
    const int cb = crange[(node * 2 + half) * 2], ce = crange[(node * 2 + half) * 2 + 1];
    const int per = (ce - cb + 31) / 32;
    const int lo = cb + sidx * per, hi = lo + per < ce ? lo + per : ce;
    double v = 0.0;
    if (c < C) for (int ch = lo; ch < hi; ++ch) v += P[((long)ch * R + k) * Cp + c];
    sl[sidx][c] = v;
    __syncthreads();
    if (threadIdx.x < MV_C) {
      double t = 0.0;
      for (int q = 0; q < 32; ++q) t += sl[q][threadIdx.x];
      tsum[row * MV_C + threadIdx.x] = threadIdx.x < C ? t : 0.0;
    }
    __syncthreads();
  }
  const double* S = Sinv + (long)node * n2 * n2;
  for (int e = threadIdx.x; e < n2 * MV_C; e += 256) {
    const int i = e >> 3, cc = e & 7;
    if (cc >= C) continue;
    double t = 0.0;
    for (int k = 0; k < n2; ++k) t += S[i * n2 + k] * tsum[k * MV_C + cc];
    Tout[((long)node * n2 + i) * Cp + cc] = t;
  }
}
// Tsum[node][0:R] = sum of the partials of its half-1 chunks, [R:2R] = half-0 chunks (hodlr.h:247-249)
// blockDim = 64 x NS: NS threads per column each add a contiguous slice of the node's chunks, the first
// then adds the NS slice sums in order (fixed order: reproducible).  The top levels have up to N/256
// chunks per half: as one thread per column this walk took 55-60 us per launch.
#define SUM_NS 8
__global__ __launch_bounds__(64 * SUM_NS) void hodlr_sum_kernel(const double* P, const int* crange /* [node][half][2] */, int R, long Cp, int C, double* Tsum) {
  __shared__ double sl[SUM_NS][64];
  const int node = blockIdx.x, row = blockIdx.y;          // row in [0, 2R)
  const int half = row < R ? 1 : 0, k = row < R ? row : row - R;
  const int cb = crange[(node * 2 + half) * 2], ce = crange[(node * 2 + half) * 2 + 1];
  const int lane = threadIdx.x & 63, sidx = threadIdx.x >> 6;
  const int per = (ce - cb + SUM_NS - 1) / SUM_NS;
  const int lo = cb + sidx * per, hi = lo + per < ce ? lo + per : ce;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    double v = 0.0;
    if (c < C) {
#pragma unroll 8                                  // (loads of 8 chunks in flight; the sum stays in chunk order)
      for (int ch = lo; ch < hi; ++ch) v += P[((long)ch * R + k) * Cp + c];
    }
    sl[sidx][lane] = v;
    __syncthreads();
    if (sidx == 0 && c < C) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < SUM_NS; ++q) t += sl[q][lane];
      Tsum[((long)node * 2 * R + row) * Cp + c] = t;
    }
    __syncthreads();
  }
}
// S = [[I, V1^T U1], [V0^T U0, I]]  (hodlr.h:229-232) from Tsum (C == R)
__global__ void hodlr_sbuild_kernel(const double* Tsum, long Cp, int R, double* S) {
  const int node = blockIdx.x, n2 = 2 * R;
  double* s = S + (long)node * n2 * n2;
  for (int e = threadIdx.x; e < n2 * n2; e += blockDim.x) {
    const int r = e / n2, c = e % n2;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < R && c >= R) v = Tsum[((long)node * n2 + r) * Cp + (c - R)];
    else if (r >= R && c < R) v = Tsum[((long)node * n2 + r) * Cp + c];
    s[e] = v;
  }
}
__global__ void hodlr_copyrows_kernel(const double* Y, long ldy, double* X, long ldx, long x_col0, long n, int C) {
  const long tot = n * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const long i = e / C;
    const int c = (int)(e % C);
    X[i * ldx + x_col0 + c] = Y[i * ldy + c];
  }
}
// out[blockIdx.x] = sum over this workgroup's contiguous slice of a[i] * b[i] (b == nullptr: of a[i]);
// called twice: 256 slices, then one workgroup over the 256 partials -- fixed order, reproducible
__global__ __launch_bounds__(256) void hodlr_dot_kernel(const double* a, const double* b, long n, double* out) {
  __shared__ double sh[4];
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) v += b ? a[i] * b[i] : a[i];
  v = hw_block_sum(v, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ void hodlr_eye_kernel(double* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i * n + i] = 1.0;
}

// ================================================================================ host side
struct HNode { int start, size, half, level, is_leaf; };
struct HLevel {
  int top_level = -1;
  bool top = false;                 // pseudo-level of a sub-tree handle (HSub below): one node, the ancestor cut down to the local rows
  std::vector<int> node_ids;
  int R = 0, off = 0, nchunks = 0;
  GhPooledBuf d_nodes, d_chunks, d_crange, d_red_jobs, d_upd_jobs, d_smul_jobs, d_ranks, sinv;   // (one stream: h->st)
  GhPooledBuf d_gj_offs, d_gj_sizes, d_gj_sc, d_updl_jobs;
  std::vector<int> ranks;
  // The job tables depend on the tree and on (R, off, Rtot) only: inside an optimiser loop neither
  // changes from one compute() to the next, and re-uploading them (~9 small copies per level, each a
  // host round trip) was ~1 ms of the 12 ms of a C4 compute.
  std::vector<int> chunk_geom;      // (row0, rows) of every chunk in order: two levels with the same list can share a pass over the rows
  bool nodes_up = false;
  std::vector<int> aca_dur;         // per node: ticks its one-workgroup ACA took in the last compute() (empty: unknown)
  GhPooledBuf d_order;              // the node order the next one-workgroup launch takes them in (longest first)
  int tab_R = -1, tab_off = -1, gj_R = -1;
  long tab_Rtot = -1;
};

// A handle can be ONE SUB-TREE of a tree that is split over several devices (gh_hodlr_mgpu, end of this file): its
// rows are rows [row0, row0 + n) of the whole problem, its tree is the sub-tree rooted at global level `depth`, and the
// `depth` levels above it appear here as PSEUDO-LEVELS of one node each -- the ancestor at that level, cut down to the
// local rows (which all lie in ONE of its halves).  Their low-rank factors are not computed here (the ACA of a top node
// runs on one device; T[l] holds the local rows of its result), and whenever such a level is applied, the 2R x C sums
// V^T X are completed over the devices below that ancestor (`allreduce`) between "sum" and "core product".  Everything
// else -- tables, kernels, the order of the sweep -- is the single-device code.
struct HSub {
  int depth = 0;                    // 0: an ordinary handle
  std::vector<int> half, R;         // [depth] the half of the level-l ancestor the local rows are in; the rank of global level l
  std::vector<const double*> T;     // [depth] column-major n x R[l]: local rows of the ancestor's ACA factors (this device)
  std::vector<int> seed_off;        // per local level: index of this sub-tree's first internal node in the global level
  void* ctx = nullptr;
  int (*allreduce)(void* ctx, int level, double* dT, int rows, int cols, long pitch, hipStream_t st) = nullptr;
  int (*local_done)(void* ctx) = nullptr;      // the part of compute() that needs no other device has been enqueued and has finished
  std::vector<double> ld_top;       // out: log|det| of the core of the level-l ancestor (every device below it computes the same)
  std::vector<int> sig() const { std::vector<int> v{depth}; v.insert(v.end(), half.begin(), half.end()); v.insert(v.end(), seed_off.begin(), seed_off.end()); return v; }
};

struct gh_hodlr {
  gh_hodlr_opts opts;
  HSub sub;
  std::vector<int> tree_sub;        // sub.sig() the tree was built for
  hipStream_t st = nullptr;
  GhBuf d_gather;                   // (fetch of every level's ranks / flags in one copy)
  int* h_gather = nullptr; size_t h_gather_cap = 0;      // pinned
  bool shared_streams = false;   // st, st_b, st_c belong to the process (gh_shared_streams, gh_common.h): not destroyed here
  hipStream_t st_b = nullptr;    // second stream: ACA of the one-workgroup-per-node levels beside the clustered ones
  hipEvent_t ev_b = nullptr;
  hipStream_t st_d = nullptr;    // fourth queue (the process-wide chain stream): one more independent ACA chain at a time
  hipEvent_t ev_d = nullptr;
  hipStream_t st_c = nullptr;    // third stream: the leaf stage, beside both ACA streams
  hipEvent_t ev_c = nullptr;
  std::vector<hipEvent_t> aca_ev;        // timing stamps of the side items of the last compute(), two per item
  size_t aca_ev_used = 0;
  std::vector<int> aca_items;            // level per item (-1: leaf stage), in stamp order
  hipEvent_t aca_fused_ev[2] = {nullptr, nullptr};
  bool aca_timed = false;
  std::vector<double> aca_ms;            // measured milliseconds: [0..nlev) levels, [nlev] fused launch, [nlev+1] leaf stage
  std::vector<char> wave_bad;            // per level: a block needed more columns than the wavefront-per-node ACA holds (hodlr_aca_wave_kernel): the workgroup kernel from then on
  int64_t n = 0;
  int ndim = 0;
  bool computed = false;
  double logdet = 0.0;
  std::vector<HNode> nodes;
  std::vector<HLevel*> levels;
  std::vector<LeafDesc> leaves;
  int Rtot = 0, max_leaf = 0, max_chunks = 0, maxR = 0;
  int leaf_pitch = 0;            // row pitch of the stored leaf inverses
  int cpass = CPASS;             // columns per apply pass = row pitch of P / Tsum / Tout / Y (>= the largest level rank)
  GhBuf x, yerr, UA, VA, leaf_inv, d_leaves, d_leaf_jobs, P, Tsum, Tout, Y, rhs, scal, work, dotp;
  GhBuf d_leaf_prod;
  GhBuf d_aca_segs, d_aca_segs1;
  double* pin = nullptr;         // pinned host block for the results compute() brings back (log|det| of every block, flags)
  size_t pin_doubles = 0;
  GhBuf UL, d_colbase, d_colld;  // level-major copy of the final U (wide solves: ensure_ul) and its column map
  bool ul_valid = false;
  long col_Rtot = -1;
  std::vector<int> col_sig;
  GhBuf ld_all, flags;           // log|det| of every factored block of a compute(); [0] Gauss-Jordan failure, [2..3] leaf info
  ~gh_hodlr() {
    for (auto* l : levels) delete l;
    if (h_gather) (void)hipHostFree(h_gather);
    if (ev_b) (void)hipEventDestroy(ev_b);
    if (st_b && !shared_streams) (void)hipStreamDestroy(st_b);
    if (ev_c) (void)hipEventDestroy(ev_c);
    if (ev_d) (void)hipEventDestroy(ev_d);
    if (st_c && !shared_streams) (void)hipStreamDestroy(st_c);
    for (auto& e : aca_ev) (void)hipEventDestroy(e);
    for (auto& e : aca_fused_ev) if (e) (void)hipEventDestroy(e);
    if (st && !shared_streams) (void)hipStreamDestroy(st);
  }
  int64_t tree_n = -1;
  int tree_min = -1;
  bool leaf_tab_up = false;
  void reset_tree() { for (auto* l : levels) delete l; levels.clear(); nodes.clear(); leaves.clear(); tree_n = -1; leaf_tab_up = false; col_Rtot = -1; col_sig.clear(); aca_ms.clear(); wave_bad.clear(); }
};

extern "C" int gh_hodlr_create(const gh_hodlr_opts* opts, gh_hodlr** out) {
  if (!out) { gh_set_error("null output"); return GH_ERR_BAD_ARG; }
  if (gh_device_count() <= 0) { gh_set_error("no HIP device available: the george_amd HODLR solver needs an MI355X"); return GH_ERR_HIP; }
  gh_hodlr* h = new gh_hodlr();
  memset(&h->opts, 0, sizeof(h->opts));
  if (opts) h->opts = *opts;
  else { h->opts.min_size = 100; h->opts.tol = 0.1; h->opts.seed = 42; }
  if (h->opts.min_size < 1) h->opts.min_size = 1;
  if (h->opts.max_rank < 0) h->opts.max_rank = 0;              // 0: as much as the tolerance asks for, up to RANK_CAP
  if (h->opts.max_rank > RANK_CAP) h->opts.max_rank = RANK_CAP;
  if (hipSetDevice(h->opts.device) != hipSuccess) { delete h; gh_set_error("cannot initialise HIP device %d", opts ? opts->device : 0); return GH_ERR_HIP; }
  hipStream_t shq[4] = {nullptr, nullptr, nullptr, nullptr};
  if (gh_shared_streams(h->opts.device, shq) && shq[2] && shq[3]) {
    h->shared_streams = true;                 // the process-wide streams: main, and two side streams for compute()
    h->st = shq[0];
  } else if ((gh_prime_device(h->opts.device), hipStreamCreate(&h->st)) != hipSuccess) {
    delete h; gh_set_error("cannot initialise HIP device %d", opts ? opts->device : 0); return GH_ERR_HIP;
  }
  *out = h;
  return GH_OK;
}
extern "C" void gh_hodlr_destroy(gh_hodlr* h) {
  if (!h) return;
  (void)hipSetDevice(h->opts.device);
  // pooled blocks may be re-acquired by another handle on another stream: nothing of this handle's may
  // still be queued when they are released
  if (h->st) (void)hipStreamSynchronize(h->st);
  if (h->st_b) (void)hipStreamSynchronize(h->st_b);
  if (h->st_c) (void)hipStreamSynchronize(h->st_c);
  if (h->st_d) (void)hipStreamSynchronize(h->st_d);
  if (h->pin) (void)hipHostFree(h->pin);
  delete h;
}

template <typename Tv>
static int upload(GhBuf& buf, const std::vector<Tv>& v, hipStream_t st) {
  GH_CHECK(buf.ensure(std::max<size_t>(v.size(), 1) * sizeof(Tv)));
  if (!v.empty()) GH_HIP(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice, st));
  return GH_OK;
}

// Ranks and failure flags of ALL levels into one staging buffer, for ONE device-to-host copy: as 22 small
// copies into pageable memory (two per level) they took 22 us each, back to back, with the GPU idle --
// 0.5 of the 6.7 ms of a C4 compute().
struct GatherItem { const int* src; int count, dst; };
__global__ void hodlr_gather_kernel(const GatherItem* items, int* out) {
  const GatherItem it = items[blockIdx.x];
  for (int i = threadIdx.x; i < it.count; i += blockDim.x) out[it.dst + i] = it.src[i];
}

// rows [0, nrows) x columns [0, 16 ct) of a row-major block into LDS (pitch xp), zero where row >= nrows or
// column >= C; eight independent loads in flight per thread (a rolled load-store loop waits for every load in turn:
// 40 round trips per workgroup)
// (round 6) The factorisation's leaf product takes its input -- the un-factored U, which is V: the compaction writes the same values
// to both -- from the LEVEL-MAJOR copy VA (level l, row i, column k at VA[offv[l] + i R[l] + k]; a leaf's rows of a level are one
// contiguous piece) and writes the row-major U for the first time: the compaction no longer writes U (157 MB at C4) for this
// kernel to read back.  Same values in the same LDS image: the same bits.
struct LeafSrc { const double* VA; int nlev; int off[24], R[24]; long offv[24]; };
__device__ __forceinline__ void hodlr_stage_rows_va(double* Xs, int xp, const LeafSrc& ls, int row0, int nrows, int C, int ct) {
  // column c of the image = column k of level l: element (r, c) at VA[colbase[c] + (row0 + r) colR[c]], colbase[c] = offv[l] + k.
  // Lanes take consecutive COLUMNS (LDS writes free of bank conflicts at a pitch that is a multiple of 16 doubles; the reads are
  // the levels' pieces of a row, 24-120 contiguous bytes each, whose neighbours the next row's loads find in the caches)
  const int tid = threadIdx.x, w = 16 * ct;
  const int c = tid & 127, rh = tid >> 7;            // two rows per step
  const bool cok = c < C;
  long cb = 0;                                       // (no table in LDS: the image is exactly half a CU's LDS at CT = 5)
  int cr = 0;
  for (int t = 0; t < ls.nlev; ++t)
    if (cok && ls.R[t] > 0 && ls.off[t] <= c) { cb = ls.offv[t] + (c - ls.off[t]); cr = ls.R[t]; }
  if (c < w) {
#pragma unroll 1
    for (int r0 = 0; r0 < 128; r0 += 16) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + 2 * q + rh;
        v[q] = (cok && r < nrows) ? ls.VA[cb + (long)(row0 + r) * cr] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) Xs[(r0 + 2 * q + rh) * xp + c] = v[q];
    }
  }
}
__device__ __forceinline__ void hodlr_stage_rows(double* Xs, int xp, const double* src, long ld, int nrows, int C, int ct) {
  const int w = 16 * ct, tot = 128 * w;
  for (int e0 = threadIdx.x; e0 < tot; e0 += 8 * 256) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + 256 * q, r = e / w, c = e - r * w;
      const bool ok = e < tot && r < nrows && c < C;
      v[q] = src[ok ? (long)r * ld + c : 0];
      if (!ok) v[q] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + 256 * q, r = e / w, c = e - r * w;
      if (e < tot) Xs[r * xp + c] = v[q];
    }
  }
}

// X_leaf <- K_leaf^-1 X_leaf for every leaf, in place, ONE workgroup per leaf: the leaf's rows of X (<= 128 x
// 16 CT columns) are staged in LDS once, K_leaf^-1 (a 128 x 128 slot, identity- or zero-padded) streams through
// the A operand straight from HBM, the 128 x 16 CT result goes back over the rows it came from.  The generic tile
// kernel took this as 16384 workgroups of one 32 x 64 tile each into a scratch copy (every U row staged four times,
// every K^-1 slab twice) plus a copy back: 505 + 57 us of the C4 sweep for 0.6 GB of traffic.
// Wavefront w: row tiles 2w, 2w+1 (16 rows each) x all CT column tiles.
// rjobs != nullptr (round 5): the chunk products V^T X of the deepest level with a positive rank -- whose chunks are the
// leaves -- are formed here too, from the result tiles while they are in registers (see hodlr_updred_kernel: the result
// layout is the B operand layout of hodlr_red_kernel's k-steps, same order, same bits), saving that level's pass over U.
template <int CT>
__global__ __launch_bounds__(256) void hodlr_leaf_apply_kernel(const MMJob* __restrict__ jobs, const double* __restrict__ Kinv,
                                                               double* __restrict__ X, long ldx, long xcol0, int C,
                                                               const MMJob* __restrict__ rjobs = nullptr, const double* __restrict__ V2 = nullptr,
                                                               int R2 = 0, double* __restrict__ P = nullptr, long ldp = 0, LeafSrc ls = LeafSrc()) {
  // (no padding column: at CT = 5 the image is then exactly 80 KiB and TWO workgroups share a CU's 160 KiB -- with 81 columns
  //  it was 83 KiB, one workgroup = one wavefront per SIMD and nothing to hide the A operand's HBM latency behind; the price is
  //  a two-way bank conflict between the lane groups fk and fk + 2 of a B fragment read)
  constexpr int XP = 16 * CT;
  __shared__ double Xs[128 * XP];
  typedef double la_v4d __attribute__((ext_vector_type(4)));
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  double* const xb = X + (long)job.b_row * ldx + xcol0;
  const int ct = (C + 15) >> 4;                   // column tiles that hold anything (uniform)
  if (ls.VA) hodlr_stage_rows_va(Xs, XP, ls, job.b_row, job.m, C, ct);      // (the factorisation's first pass: X is written here for the first time)
  else hodlr_stage_rows(Xs, XP, xb, ldx, job.m, C, ct);
  __syncthreads();
  la_v4d acc[2][CT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = (la_v4d){0.0, 0.0, 0.0, 0.0};
  // A operand: K^-1(row 32 wave + 16 i + fr, k = 4 kk + fk), read as its mirror image K^-1(k, row) -- the leaf inverse is
  // symmetric (up to the last bit) and in this form the 16 lanes fr of a load are 128 contiguous bytes of row k instead of
  // 16 rows x 8 bytes (round 5: 208 -> see profiles/r05/hodlr_passes.md)
  const double* const ka = Kinv + job.a_off + (long)fk * 128 + 32 * wave + fr;
#pragma unroll 1
  for (int k0 = 0; k0 < 32; k0 += 8) {
    double a[8][2];
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q][0] = ka[(long)(4 * (k0 + q)) * 128]; a[q][1] = ka[(long)(4 * (k0 + q)) * 128 + 16]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double* const bp = Xs + (4 * (k0 + q) + fk) * XP + fr;
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        if (j >= ct) continue;
        const double b = bp[16 * j];
        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][1], b, acc[1][j], 0, 0, 0);
      }
    }
  }
  // f64 MFMA C/D map: row = (lane >> 4) + 4 reg, col = lane & 15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 32 * wave + 16 * i + fk + 4 * r;
      if (row >= job.m) continue;
#pragma unroll
      for (int j = 0; j < CT; ++j)
        if (16 * j + fr < C) xb[(long)row * ldx + 16 * j + fr] = acc[i][j][r];
    }
  if (!rjobs) return;                             // (uniform)
  const MMJob rj = rjobs[blockIdx.x];
  la_v4d acc2[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc2[j] = (la_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 8; ++q) {                   // k-step q = 4 i + r: rows 32 wave + 16 i + 4 r + fk
    const int row = 4 * (8 * wave + q) + fk;
    const double a2 = (fr < R2 && row < rj.kd) ? V2[rj.a_off + (long)row * R2 + fr] : 0.0;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      if (j >= ct) continue;
      // (what hodlr_red_kernel would read back from its LDS image of X: zero outside the leaf's rows and the C columns)
      const double bv = (row < job.m && 16 * j + fr < C) ? acc[q >> 2][j][q & 3] : 0.0;
      acc2[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bv, acc2[j], 0, 0, 0);
    }
  }
  __syncthreads();                                // (every wavefront is done with Xs)
  double* const part = Xs;                        // [3][CT][4][64]
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave - 1) * CT + j) * 4 + r) * 64 + lane] = acc2[j][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = fk + 4 * r, c = 16 * j + fr;
        if (k < R2 && c < C) {
          const double v = ((acc2[j][r] + part[((0 * CT + j) * 4 + r) * 64 + lane]) + part[((1 * CT + j) * 4 + r) * 64 + lane]) +
                           part[((2 * CT + j) * 4 + r) * 64 + lane];
          P[(long)(rj.o_row + k) * ldp + c] = v;
        }
      }
  }
}

// The per-chunk products V_l^T U of the factorisation sweep (R <= 16 columns of V, <= 128 rows of a chunk, C <= 16 CT
// columns of U), ONE workgroup per chunk in the manner of hodlr_leaf_apply_kernel: the chunk's rows of U staged in
// LDS once, V^T as the A operand straight from HBM (row k of the result = column k of V), the K = 128 rows split
// over the four wavefronts and their partial tiles added in a fixed order (bitwise repeatable).
//   O[(o_row + k) * ldo + o_col0 + c] = sum_row A[a_off + k + row * R] * B[(b_row + row) * ldb + b_col0 + c]
template <int CT>
__global__ __launch_bounds__(256) void hodlr_red_kernel(const MMJob* __restrict__ jobs, const double* __restrict__ A, int R,
                                                        const double* __restrict__ B, long ldb, long b_col0,
                                                        double* __restrict__ O, long ldo, long o_col0, int C) {
  constexpr int XP = 16 * CT + 1;
  __shared__ double Xs[128 * XP];
  typedef double rk_v4d __attribute__((ext_vector_type(4)));
  const MMJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const double* const bb = B + (long)job.b_row * ldb + b_col0;
  const int ct = (C + 15) >> 4;                   // column tiles that hold anything (uniform)
  hodlr_stage_rows(Xs, XP, bb, ldb, job.kd, C, ct);
  // this wavefront's eight k steps of the A operand: V^T(fr, 4 kk + fk) = V(row, fr)
  double a[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = 4 * (8 * wave + q) + fk;
    a[q] = (fr < R && row < job.kd) ? A[job.a_off + (long)row * R + fr] : 0.0;
  }
  __syncthreads();
  rk_v4d acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = (rk_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double* const bp = Xs + (4 * (8 * wave + q) + fk) * XP + fr;
#pragma unroll
    for (int j = 0; j < CT; ++j)
      if (j < ct) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bp[16 * j], acc[j], 0, 0, 0);
  }
  __syncthreads();                                // (Xs is free: the partial tiles of wavefronts 1-3 go there)
  double* const part = Xs;                        // [3][CT][4][64]
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave - 1) * CT + j) * 4 + r) * 64 + lane] = acc[j][r];
  }
  __syncthreads();
  if (wave == 0) {
    // f64 MFMA C/D map: row = (lane >> 4) + 4 reg, col = lane & 15
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = fk + 4 * r, c = 16 * j + fr;
        if (k < R && c < C) {
          const double v = ((acc[j][r] + part[((0 * CT + j) * 4 + r) * 64 + lane]) + part[((1 * CT + j) * 4 + r) * 64 + lane]) +
                           part[((2 * CT + j) * 4 + r) * 64 + lane];
          O[(long)(job.o_row + k) * ldo + o_col0 + c] = v;
        }
      }
  }
}
// The rank-R updates of the factorisation sweep, U[rows of a chunk, 0:C] -= U_l[rows, 0:R] * T[b_row : b_row+R, 0:C]
// (R <= 16, <= 128 rows, C <= 16 CT), one workgroup per chunk without any LDS: the accumulators start from the
// O tiles themselves (each lane's four rows x one column of a 16 x 16 tile, 128-byte row segments), K = R is
// padded to 16 only (the tile kernel pads to 32 and stages both operands), operands straight from HBM / L2.
//   O[(o_row + r) * ldo + c] -= sum_k A[a_off + r * a_rs + k] * B[(b_row + k) * ldb + c]
template <int CT>
__global__ __launch_bounds__(256) void hodlr_upd_kernel(const MMJob* __restrict__ jobs, const double* __restrict__ A, long a_rs,
                                                        const double* __restrict__ B, long ldb, double* __restrict__ O, long ldo, int C) {
  typedef double uk_v4d __attribute__((ext_vector_type(4)));
  const MMJob job = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int R = job.kd, nkk = (R + 3) >> 2, ct = (C + 15) >> 4;        // (uniform)
  double a[2][4], b[4][CT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int row = 32 * wave + 16 * i + fr, k = 4 * kk + fk;
      a[i][kk] = (row < job.m && k < R) ? -A[job.a_off + (long)row * a_rs + k] : 0.0;
    }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int k = 4 * kk + fk, c = 16 * j + fr;
      b[kk][j] = (k < R && c < C) ? B[(long)(job.b_row + k) * ldb + c] : 0.0;
    }
  double* const ob = O + (long)job.o_row * ldo;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (32 * wave + 16 * i >= job.m) continue;                         // (uniform)
    uk_v4d acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * i + fk + 4 * r, c = 16 * j + fr;
        acc[j][r] = (j < ct && row < job.m && c < C) ? ob[(long)row * ldo + c] : 0.0;
      }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk >= nkk) continue;
#pragma unroll
      for (int j = 0; j < CT; ++j)
        if (j < ct) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][kk], b[kk][j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * i + fk + 4 * r, c = 16 * j + fr;
        if (j < ct && row < job.m && c < C) ob[(long)row * ldo + c] = acc[j][r];
      }
  }
}
// hodlr_upd_kernel for level l and hodlr_red_kernel for the next shallower level l' in ONE pass over a chunk's rows of U
// (round 5).  The reduce of l' reads columns [0, off_l' + R_l') = [0, off_l) of U -- exactly what the update of l has just
// written: here the updated tiles go to HBM and into the LDS image the reduce multiplies from, and the sweep reads U once
// per level instead of twice (C4: upd<4> 52 us + red<4> 66 us per level, both HBM-bound).  The update's arithmetic is
// hodlr_upd_kernel's, the reduce multiplies the same doubles hodlr_red_kernel would have staged from HBM, in the same order:
// bit-identical to the two launches.  Needs the two levels' chunks to be the same rows (HLevel::chunk_geom).
template <int CT>
__global__ __launch_bounds__(256) void hodlr_updred_kernel(const MMJob* __restrict__ ujobs, const double* __restrict__ A, long a_rs,
                                                           const double* __restrict__ B, long ldb, double* __restrict__ O, long ldo, int C,
                                                           const MMJob* __restrict__ rjobs, const double* __restrict__ V2, int R2,
                                                           double* __restrict__ P, long ldp) {
  __shared__ double part[3 * CT * 4 * 64];
  typedef double uk_v4d __attribute__((ext_vector_type(4)));
  const MMJob job = ujobs[blockIdx.x];
  const MMJob rj = rjobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int R = job.kd, nkk = (R + 3) >> 2;                             // (uniform; every one of the CT column tiles is computed:
  // columns >= C are zero on both sides, and a `tile j < ceil(C / 16)` test in front of each matrix instruction made hipcc keep
  // both versions of every accumulator -- 256 VGPRs at CT = 4)
  // the reduce's A operand (V_l'^T: this wavefront's eight k steps, q = 4 i + r <-> rows 32 wave + 16 i + 4 r + fk), requested
  // first: it lands under the update
  double a2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = 4 * (8 * wave + q) + fk;
    a2[q] = (fr < R2 && row < rj.kd) ? V2[rj.a_off + (long)row * R2 + fr] : 0.0;
  }
  double b[4][CT];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int k = 4 * kk + fk, c = 16 * j + fr;
      b[kk][j] = (k < R && c < C) ? B[(long)(job.b_row + k) * ldb + c] : 0.0;
    }
  double* const ob = O + (long)job.o_row * ldo;
  uk_v4d acc2[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc2[j] = (uk_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uk_v4d acc[CT];
    const bool live = 32 * wave + 16 * i < job.m;                      // (uniform)
    // (one 16-row tile at a time, fenced: with both tiles' loads hoisted to the top hipcc needs 256 VGPRs at CT = 4 -- two
    //  wavefronts per SIMD for a kernel that lives on memory latency, 127-136 us per level against 52 + 66 for the two launches)
    double a[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int row = 32 * wave + 16 * i + fr, k = 4 * kk + fk;
      a[kk] = (row < job.m && k < R) ? -A[job.a_off + (long)row * a_rs + k] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * i + fk + 4 * r, c = 16 * j + fr;
        acc[j][r] = (live && row < job.m && c < C) ? ob[(long)row * ldo + c] : 0.0;
      }
    if (live) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk >= nkk) continue;
#pragma unroll
        for (int j = 0; j < CT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk][j], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 * wave + 16 * i + fk + 4 * r, c = 16 * j + fr;
          if (row < job.m && c < C) ob[(long)row * ldo + c] = acc[j][r];
        }
    }
    // The reduce, straight from the registers: the update's result layout (lane (fr, fk), register r of tile j = row 4 r + fk,
    // column 16 j + fr of this 16-row tile) IS the matrix instruction's B operand layout for the k-step over rows 4 r .. 4 r + 3
    // (B[k = fk][n = fr]) -- hodlr_red_kernel stages exactly these values through LDS and reads them back into this position.
    // Same k-steps in the same order (q = 4 i + r) on the same wavefront: the same bits.
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < CT; ++j)
        acc2[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[4 * i + r], acc[j][r], acc2[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave - 1) * CT + j) * 4 + r) * 64 + lane] = acc2[j][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = fk + 4 * r, c = 16 * j + fr;
        if (k < R2 && c < C) {
          const double v = ((acc2[j][r] + part[((0 * CT + j) * 4 + r) * 64 + lane]) + part[((1 * CT + j) * 4 + r) * 64 + lane]) +
                           part[((2 * CT + j) * 4 + r) * 64 + lane];
          P[(long)(rj.o_row + k) * ldp + c] = v;
        }
      }
  }
}
static int launch_mm(gh_hodlr* h, const MMJob* jobs, int njobs, int max_m, const double* A, long a_rs, long a_cs,
                     const double* B, long ldb, long b_col0, double* O, long ldo, long o_col0, int C, bool subtract, int mtiles = 1);
static int launch_red(gh_hodlr* h, const MMJob* jobs, int njobs, int R, const double* V, const double* B, long ldb, long b_col0,
                      double* O, long ldo, long o_col0, int C) {
  if (njobs <= 0 || C <= 0 || R <= 0) return GH_OK;
  if (R > 16 || C > 128) return launch_mm(h, jobs, njobs, R, V, 1, R, B, ldb, b_col0, O, ldo, o_col0, C, false, 1);
  // (one instantiation per number of 16-column tiles: the LDS image is 128 x (16 CT + 1) doubles, and with 17-50 KB
  //  instead of 83 several workgroups share a CU at the shallow levels, whose U has few columns yet)
#define GH_RED_LAUNCH(CT) hipLaunchKernelGGL(hodlr_red_kernel<CT>, dim3(njobs), dim3(256), 0, h->st, jobs, V, R, B, ldb, b_col0, O, ldo, o_col0, C)
  switch ((C + 15) / 16) {
    case 1: GH_RED_LAUNCH(1); break;
    case 2: GH_RED_LAUNCH(2); break;
    case 3: GH_RED_LAUNCH(3); break;
    case 4: GH_RED_LAUNCH(4); break;
    case 5: GH_RED_LAUNCH(5); break;
    default: GH_RED_LAUNCH(8); break;
  }
#undef GH_RED_LAUNCH
  GH_HIP(hipGetLastError());
  return GH_OK;
}

static int launch_upd(gh_hodlr* h, const MMJob* jobs, int njobs, int R, const double* A, long a_rs, const double* B, long ldb,
                      double* O, long ldo, int C) {
  if (njobs <= 0 || C <= 0 || R <= 0) return GH_OK;
  if (R > 16 || C > 128 || HCH > 128) return launch_mm(h, jobs, njobs, HCH, A, a_rs, 1, B, ldb, 0, O, ldo, 0, C, true, HCH / 32);
#define GH_UPD_LAUNCH(CT) hipLaunchKernelGGL(hodlr_upd_kernel<CT>, dim3(njobs), dim3(256), 0, h->st, jobs, A, a_rs, B, ldb, O, ldo, C)
  switch ((C + 15) / 16) {
    case 1: GH_UPD_LAUNCH(1); break;
    case 2: GH_UPD_LAUNCH(2); break;
    case 3: GH_UPD_LAUNCH(3); break;
    case 4: GH_UPD_LAUNCH(4); break;
    case 5: GH_UPD_LAUNCH(5); break;
    default: GH_UPD_LAUNCH(8); break;
  }
#undef GH_UPD_LAUNCH
  GH_HIP(hipGetLastError());
  return GH_OK;
}

static int hodlr_passes();
// update of level `L` (columns [0, C) of U, C = L->off) + reduce of level `nx` over the same columns in one pass; false when
// the pair cannot share a pass (the caller then launches the two kernels)
static bool updred_possible(const HLevel* L, const HLevel* nx, int C, int cpass) {
  return (hodlr_passes() & 2) && nx && !nx->top && !L->top && L->R <= 16 && nx->R <= 16 && C > 0 && C <= 128 && HCH == 128 &&
         nx->off + nx->R == C && C <= cpass && nx->chunk_geom == L->chunk_geom && !L->chunk_geom.empty();
}
static int launch_updred(gh_hodlr* h, const HLevel* L, const HLevel* nx, const double* A, long a_rs, const double* B, long ldb,
                         double* O, long ldo, int C, const double* V2, double* P, long ldp) {
  const MMJob* uj = (const MMJob*)L->d_upd_jobs.p;
  const MMJob* rj = (const MMJob*)nx->d_red_jobs.p;
#define GH_UR_LAUNCH(CT) hipLaunchKernelGGL(hodlr_updred_kernel<CT>, dim3(L->nchunks), dim3(256), 0, h->st, uj, A, a_rs, B, ldb, O, ldo, C, rj, V2, nx->R, P, ldp)
  switch ((C + 15) / 16) {
    case 1: GH_UR_LAUNCH(1); break;
    case 2: GH_UR_LAUNCH(2); break;
    case 3: GH_UR_LAUNCH(3); break;
    case 4: GH_UR_LAUNCH(4); break;
    case 5: GH_UR_LAUNCH(5); break;
    default: GH_UR_LAUNCH(8); break;
  }
#undef GH_UR_LAUNCH
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// mtiles: 32-row tiles of a job handled by ONE workgroup (the update passes: 4, i.e. a whole 128-row
// chunk -- 8192 workgroups of one tiny tile each spent their 50 us on being dispatched)
static int launch_mm(gh_hodlr* h, const MMJob* jobs, int njobs, int max_m, const double* A, long a_rs, long a_cs,
                     const double* B, long ldb, long b_col0, double* O, long ldo, long o_col0, int C, bool subtract, int mtiles) {
  if (njobs <= 0 || C <= 0 || max_m <= 0) return GH_OK;
  MMArgs a;
  a.mtiles = mtiles;
  a.jobs = jobs; a.A = A; a.a_rs = a_rs; a.a_cs = a_cs; a.B = B; a.ldb = ldb; a.b_col0 = b_col0;
  a.O = O; a.ldo = ldo; a.o_col0 = o_col0; a.C = C; a.subtract = subtract ? 1 : 0;
  hipLaunchKernelGGL(hodlr_mm_kernel, dim3(njobs, ((max_m + 31) / 32 + mtiles - 1) / mtiles, (C + 63) / 64), dim3(256), 0, h->st, a);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// level-major copy UL of the final U (every level's columns contiguous: what the wide solves' tile kernel wants), made on demand
static int ensure_ul(gh_hodlr* h) {
  if (h->ul_valid || h->Rtot <= 0) return GH_OK;
  const long n = h->n, Rtot = h->Rtot;
  hipStream_t st = h->st;
  GH_CHECK(h->UL.ensure((size_t)n * Rtot * sizeof(double)));
  std::vector<long> colbase(Rtot);
  std::vector<int> colld(Rtot);
  for (auto* L : h->levels)
    for (int kk = 0; kk < L->R; ++kk) { colbase[L->off + kk] = (long)n * L->off + kk; colld[L->off + kk] = L->R; }
  // (cached like the job tables: same ranks, same map)
  bool same = h->col_Rtot == Rtot && h->col_sig.size() == h->levels.size();
  for (size_t q = 0; same && q < h->levels.size(); ++q) same = h->col_sig[q] == h->levels[q]->R;
  if (!same) {
    GH_CHECK(upload(h->d_colbase, colbase, st));
    GH_CHECK(upload(h->d_colld, colld, st));
    h->col_Rtot = Rtot;
    h->col_sig.clear();
    for (auto* L : h->levels) h->col_sig.push_back(L->R);
  }
  hipLaunchKernelGGL(hodlr_relayout_kernel, dim3(2048), dim3(256), 0, st, h->UA.d(), (long)n, (int)Rtot,
                     (const long*)h->d_colbase.p, (const int*)h->d_colld.p, h->UL.d());
  GH_HIP(hipGetLastError());
  h->ul_valid = true;
  return GH_OK;
}
// X[:, xcol0 : xcol0+C] <- (level lv)^-1 applied (hodlr.h:244-253 for every node of the level)
// (U == nullptr: the level-major copy UL is used -- solves; else the row-major UA with pitch ldu)
static int apply_level(gh_hodlr* h, HLevel* L, double* X, long ldx, long xcol0, int C, const double* U, long ldu) {
  if (L->R == 0 || C <= 0) return GH_OK;
  if (!U) GH_CHECK(ensure_ul(h));
  const int R = L->R, nn = (int)L->node_ids.size();
  const double* Vl = h->VA.d() + (long)h->n * L->off;
  if (C <= MV_C && R <= 32) {
    const long Cp = h->cpass;
    const double* Ub = U ? U + L->off : h->UL.d() + (long)h->n * L->off;
    const long u_rs = U ? ldu : R;
    const MMJob* uj = (const MMJob*)(U ? L->d_upd_jobs.p : L->d_updl_jobs.p);
    hipLaunchKernelGGL(hodlr_mv_reduce_kernel, dim3(L->nchunks), dim3(256), 0, h->st, (const MMJob*)L->d_red_jobs.p, Vl, R, X, ldx, xcol0,
                       h->P.d(), Cp, C);
    hipLaunchKernelGGL(hodlr_sum_narrow_kernel, dim3(nn, 2 * R), dim3(256), 0, h->st, h->P.d(), (const int*)L->d_crange.p, R, Cp, C, h->Tsum.d());
    GH_HIP(hipGetLastError());
    if (L->top) GH_CHECK(h->sub.allreduce(h->sub.ctx, L->top_level, h->Tsum.d(), 2 * R, C, Cp, h->st));
    GH_CHECK(launch_mm(h, (const MMJob*)L->d_smul_jobs.p, nn, 2 * R, L->sinv.d(), 2 * R, 1,
                       h->Tsum.d(), Cp, 0, h->Tout.d(), Cp, 0, C, false));
    hipLaunchKernelGGL(hodlr_mv_update_kernel, dim3(L->nchunks), dim3(128), 0, h->st, uj, Ub, u_rs, h->Tout.d(), Cp, X, ldx, xcol0, C);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  // (a pseudo-level goes in passes of CPASS columns whatever this handle's own pass width: the devices below the
  //  ancestor must agree on the number and the shape of the sums they complete together)
  const int pw = L->top ? CPASS : h->cpass;
  for (int cp = 0; cp < C; cp += pw) {
    const int cw = std::min(pw, C - cp);
    const long Cp = h->cpass;
    // reduce: P[chunk] = V_chunk^T X_chunk
    GH_CHECK(launch_mm(h, (const MMJob*)L->d_red_jobs.p, L->nchunks, R, Vl, 1, R,
                       X, ldx, xcol0 + cp, h->P.d(), Cp, 0, cw, false));
    hipLaunchKernelGGL(hodlr_sum_kernel, dim3(nn, 2 * R), dim3(64 * SUM_NS), 0, h->st, h->P.d(), (const int*)L->d_crange.p, R, Cp, cw, h->Tsum.d());
    GH_HIP(hipGetLastError());
    if (L->top) GH_CHECK(h->sub.allreduce(h->sub.ctx, L->top_level, h->Tsum.d(), 2 * R, cw, Cp, h->st));
    // core: Tout = S^-1 Tsum
    GH_CHECK(launch_mm(h, (const MMJob*)L->d_smul_jobs.p, nn, 2 * R, L->sinv.d(), 2 * R, 1,
                       h->Tsum.d(), Cp, 0, h->Tout.d(), Cp, 0, cw, false));
    // update: X_chunk -= U_chunk * Tout[half]
    if (U)
      GH_CHECK(launch_mm(h, (const MMJob*)L->d_upd_jobs.p, L->nchunks, HCH, U + L->off, ldu, 1,
                         h->Tout.d(), Cp, 0, X, ldx, xcol0 + cp, cw, true, HCH / 32));
    else
      GH_CHECK(launch_mm(h, (const MMJob*)L->d_updl_jobs.p, L->nchunks, HCH, h->UL.d() + (long)h->n * L->off, R, 1,
                         h->Tout.d(), Cp, 0, X, ldx, xcol0 + cp, cw, true, HCH / 32));
  }
  return GH_OK;
}
// X rows of every leaf <- K_leaf^-1 X
// red / red_done: the sweep's first call -- form the chunk products of level `red` over the same columns in the same pass when
// its chunks are the leaves (then *red_done = true and the caller skips that level's reduce)
static int hodlr_passes();
static int apply_leaves(gh_hodlr* h, double* X, long ldx, long xcol0, int C, const HLevel* red = nullptr, bool* red_done = nullptr,
                        const LeafSrc* src = nullptr) {
  const LeafSrc ls = src ? *src : LeafSrc();         // (src: only on the 128-row-leaf path with ONE column pass -- leaf_src_possible())
  if (red_done) *red_done = false;
  if (C <= 0) return GH_OK;
  if (C <= MV_C && h->max_leaf <= 256) {
    hipLaunchKernelGGL(hodlr_mv_leaf_kernel, dim3((unsigned)h->leaves.size()), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p,
                       h->leaf_inv.d(), (long)h->leaf_pitch, X, ldx, xcol0, C);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  if (h->leaf_pitch == 128 && h->max_leaf <= 128) {
    // one workgroup per leaf, in place; column passes of <= 128 (80 where that covers the rest: less LDS, fewer MFMAs)
    bool fuse = (hodlr_passes() & 2) && red && red_done && C <= 128 && xcol0 == 0 && red->R > 0 && red->R <= 16 && !red->top &&
                red->off + red->R == C && C <= h->cpass && red->chunk_geom.size() == 2 * h->leaves.size();
    for (size_t q = 0; fuse && q < h->leaves.size(); ++q)
      fuse = red->chunk_geom[2 * q] == h->leaves[q].start && red->chunk_geom[2 * q + 1] == h->leaves[q].size;
    if (fuse) {
      const unsigned nl = (unsigned)h->leaves.size();
      const MMJob* rj = (const MMJob*)red->d_red_jobs.p;
      const double* V2 = h->VA.d() + (long)h->n * red->off;
      if (C <= 80) hipLaunchKernelGGL(hodlr_leaf_apply_kernel<5>, dim3(nl), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p, h->leaf_inv.d(), X, ldx, xcol0, C,
                                      rj, V2, red->R, h->P.d(), (long)h->cpass, ls);
      else hipLaunchKernelGGL(hodlr_leaf_apply_kernel<8>, dim3(nl), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p, h->leaf_inv.d(), X, ldx, xcol0, C,
                              rj, V2, red->R, h->P.d(), (long)h->cpass, ls);
      GH_HIP(hipGetLastError());
      *red_done = true;
      return GH_OK;
    }
    for (int cp = 0; cp < C;) {
      const int cw = std::min(128, C - cp);
      const unsigned nl = (unsigned)h->leaves.size();
      if (cw <= 80) hipLaunchKernelGGL(hodlr_leaf_apply_kernel<5>, dim3(nl), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p, h->leaf_inv.d(), X, ldx, xcol0 + cp, cw,
                                       (const MMJob*)nullptr, (const double*)nullptr, 0, (double*)nullptr, 0L, ls);
      else hipLaunchKernelGGL(hodlr_leaf_apply_kernel<8>, dim3(nl), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p, h->leaf_inv.d(), X, ldx, xcol0 + cp, cw,
                              (const MMJob*)nullptr, (const double*)nullptr, 0, (double*)nullptr, 0L, ls);
      GH_HIP(hipGetLastError());
      cp += cw;
    }
    return GH_OK;
  }
  for (int cp = 0; cp < C; cp += h->cpass) {
    const int cw = std::min(h->cpass, C - cp);
    GH_CHECK(launch_mm(h, (const MMJob*)h->d_leaf_jobs.p, (int)h->leaves.size(), h->max_leaf, h->leaf_inv.d(), h->leaf_pitch, 1,
                       X, ldx, xcol0 + cp, h->Y.d(), h->cpass, 0, cw, false));
    const long tot = h->n * cw;
    hipLaunchKernelGGL(hodlr_copyrows_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 65535)), dim3(256), 0, h->st,
                       h->Y.d(), (long)h->cpass, X, ldx, xcol0 + cp, (long)h->n, cw);
    GH_HIP(hipGetLastError());
  }
  return GH_OK;
}
// full solve on X (n x C): leaves, then levels bottom-up (hodlr.h:107-114)
// gh_debug_set_hodlr_passes (A/B in one process, tests): bit 0 = the narrow solve in shared passes (round 5), bit 1 = the
// factorisation sweep's update of level l and reduce of the next level in one pass over U; default: both
static int g_hodlr_passes = -1;
static int hodlr_passes() {
  if (g_hodlr_passes < 0) g_hodlr_passes = 3;
  return g_hodlr_passes;
}
// 1: leaves of 129 .. 256 rows through the pivoted Gauss-Jordan in place (the path every leaf of more than 256 rows takes)
// instead of the 2 x 2 blocked Cholesky: validation arm, tests/test_gpu_hodlr.py
static int g_hodlr_leaf_gj = 0;
extern "C" int gh_debug_set_hodlr_leaf_gj(int on) {
  const int prev = g_hodlr_leaf_gj;
  g_hodlr_leaf_gj = on ? 1 : 0;
  return prev;
}
extern "C" int gh_debug_set_hodlr_passes(int mask) {
  const int prev = hodlr_passes();
  g_hodlr_passes = mask < 0 ? 3 : (mask & 3);
  return prev;
}
// the narrow solve: leaves (symmetric form), then per level "sum + core product" and ONE pass over the rows that applies this
// level's update and forms the next level's chunk products (separate passes where the two levels' chunks differ)
static int solve_narrow(gh_hodlr* h, double* X, long ldx, int C) {
  hipLaunchKernelGGL(hodlr_mv_leaf_sym_kernel, dim3((unsigned)h->leaves.size()), dim3(256), 0, h->st, (const MMJob*)h->d_leaf_jobs.p,
                     h->leaf_inv.d(), (long)h->leaf_pitch, X, ldx, 0L, C);
  std::vector<HLevel*> Ls;
  for (int l = (int)h->levels.size() - 1; l >= 0; --l) if (h->levels[l]->R > 0) Ls.push_back(h->levels[l]);
  const long Cp = h->cpass;
  auto pass = [&](HLevel* up, HLevel* red) {
    HLevel* g = up ? up : red;
    const size_t lds = up ? (size_t)128 * (up->R <= 16 ? 17 : 33) * sizeof(double) : 0;
    hipLaunchKernelGGL(hodlr_mv_updred_kernel, dim3(g->nchunks), dim3(256), lds, h->st,
                       up ? (const MMJob*)up->d_upd_jobs.p : (const MMJob*)nullptr, up ? h->UA.d() + up->off : (const double*)nullptr,
                       (long)h->Rtot, (const double*)h->Tout.d(),
                       red ? (const MMJob*)red->d_red_jobs.p : (const MMJob*)nullptr, red ? h->VA.d() + (long)h->n * red->off : (const double*)nullptr,
                       red ? red->R : 0, h->P.d(), Cp, X, ldx, 0L, C);
  };
  if (!Ls.empty()) pass(nullptr, Ls[0]);
  for (size_t i = 0; i < Ls.size(); ++i) {
    HLevel* L = Ls[i];
    // (one workgroup per node adds the partials of ALL 2R rows: fine while a half has <= 64 chunks -- the deep levels, many nodes;
    //  the few nodes of the top levels have up to N / 256 chunks per half and keep one workgroup per (node, row) + the product)
    const int nn = (int)L->node_ids.size();
    if ((long)L->nchunks <= 128L * nn) {
      hipLaunchKernelGGL(hodlr_mv_summm_kernel, dim3((unsigned)nn), dim3(256), 0, h->st, h->P.d(), (const int*)L->d_crange.p, L->R, Cp, C,
                         (const double*)L->sinv.d(), h->Tout.d());
    } else {
      hipLaunchKernelGGL(hodlr_sum_narrow_kernel, dim3(nn, 2 * L->R), dim3(256), 0, h->st, h->P.d(), (const int*)L->d_crange.p, L->R, Cp, C, h->Tsum.d());
      GH_CHECK(launch_mm(h, (const MMJob*)L->d_smul_jobs.p, nn, 2 * L->R, L->sinv.d(), 2 * L->R, 1, h->Tsum.d(), Cp, 0, h->Tout.d(), Cp, 0, C, false));
    }
    HLevel* nx = i + 1 < Ls.size() ? Ls[i + 1] : nullptr;
    if (nx && nx->chunk_geom == L->chunk_geom) pass(L, nx);
    else { pass(L, nullptr); if (nx) pass(nullptr, nx); }
  }
  GH_HIP(hipGetLastError());
  return GH_OK;
}
static int solve_all(gh_hodlr* h, double* X, long ldx, int C) {
  if ((hodlr_passes() & 1) && C <= MV_C && h->max_leaf <= 256 && h->sub.depth == 0) {
    bool ok = true;
    for (auto* L : h->levels) ok = ok && !L->top && L->R <= 32 && (L->R == 0 || !L->chunk_geom.empty());
    if (ok) return solve_narrow(h, X, ldx, C);
  }
  GH_CHECK(apply_leaves(h, X, ldx, 0, C));
  for (int l = (int)h->levels.size() - 1; l >= 0; --l)
    GH_CHECK(apply_level(h, h->levels[l], X, ldx, 0, C, nullptr, 0));
  return GH_OK;
}

// enqueue only: logdet[b] of matrix b goes to d_logdet[b] (device), a singular block raises h->flags[0]
// (tables: device copies of offs / sizes / scratch offsets kept by the caller; *tables_valid says they
//  already hold this batch's values)
// tsum / tsum_R: see gj_small_kernel (the caller then skips hodlr_sbuild_kernel)
// ---------------------------------------------------------------------------------------------------------------
// (round 6) "sum + core inverse + core product" of one level in ONE launch, a workgroup per node, for the levels with many small
// nodes (C4: levels 5-10, three launches of 5-15 us each per level for a few kFLOP per node):
//   Tsum (2R x C, in LDS only) = the node's chunk partials added up -- hodlr_sum_kernel's order: eight slices of consecutive chunks,
//        each summed from 0.0 in chunk order, then the slice sums in order;
//   S^-1 = the pivoted Gauss-Jordan of gj_small4_kernel (columns over the four wavefronts; a column's arithmetic does not depend on
//        which wavefront holds it), built from Tsum's own-level columns, written to `sinv` for the solves; log|det S| likewise;
//   Tout = S^-1 Tsum[:, 0:Cmm] on the matrix pipe with hodlr_mm_kernel's tile and k order.
// The same doubles through the same operations as the three launches: the same bits (tests/test_gpu_hodlr.py).
template <int NMAX>
__global__ __launch_bounds__(256) void hodlr_core_kernel(const double* __restrict__ P, const int* __restrict__ crange, int R, long Cp, int C,
                                                         int coff, int Cmm, double* __restrict__ sinv, double* __restrict__ logdet,
                                                         int* __restrict__ fail, double* __restrict__ Tout) {
  constexpr int CW = NMAX / 4;
  constexpr int TP = 129;                            // row pitch of the Tsum image (C <= 128)
  __shared__ double ts[NMAX * TP];
  __shared__ double ainv[NMAX * NMAX];
  __shared__ double fcol[2][64];
  __shared__ int sp[2];
  const int node = blockIdx.x, n = 2 * R, tid = threadIdx.x;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- (1) Tsum
  for (int e = tid; e < n * C; e += 256) {
    const int row = e / C, c = e - row * C;
    const int half = row < R ? 1 : 0, k = row < R ? row : row - R;
    const int cb = crange[(node * 2 + half) * 2], ce = crange[(node * 2 + half) * 2 + 1];
    const int per = (ce - cb + SUM_NS - 1) / SUM_NS;
    double t = 0.0;
    for (int q = 0; q < SUM_NS; ++q) {
      const int lo = cb + q * per, hi = lo + per < ce ? lo + per : ce;
      double v = 0.0;
      for (int ch = lo; ch < hi; ++ch) v += P[((long)ch * R + k) * Cp + c];
      t += v;
    }
    ts[row * TP + c] = t;
  }
  __syncthreads();
  // ---- (2) S = [[I, V1^T U1], [V0^T U0, I]] (hodlr.h:229-232) and its inverse: gj_small4_kernel on the LDS image
  const bool row = lane < n;
  double m[CW];
#pragma unroll
  for (int q = 0; q < CW; ++q) {
    const int c = 4 * q + w;
    double v = (lane == c) ? 1.0 : 0.0;
    if (row && c < n) {
      if (lane < R && c >= R) v = ts[lane * TP + coff + c - R];
      else if (lane >= R && c < R) v = ts[lane * TP + coff + c];
    }
    m[q] = (row && c < n) ? v : 0.0;
  }
  bool used = false, bad = false;
  int myk = 0, pcw[CW];
  double mypiv = 1.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < n && !bad) {
      const int buf = k & 1, own = k & 3, qk = k >> 2;
      if (w == own) {
        const bool cand = row && !used;
        const unsigned key = cand ? (unsigned)__double2hiint(fabs(m[qk])) : 0u;
        const unsigned rm = gj_row_max_u32(key);
        const unsigned mx = max((unsigned)__builtin_amdgcn_readlane((int)rm, 0), (unsigned)__builtin_amdgcn_readlane((int)rm, 16));
        const unsigned long long who = __builtin_amdgcn_ballot_w64(cand && key == mx);
        const int p = (int)__builtin_ctzll(who | (1ull << 63));
        const double pv = gj_bcast(m[qk], p);
        fcol[buf][lane] = m[qk];
        if (lane == 0) sp[buf] = (who == 0ull || !(fabs(pv) > 0.0)) ? -1 : p;
      }
      __syncthreads();
      const int p = __builtin_amdgcn_readfirstlane(sp[buf]);
      if (p < 0) { bad = true; }
      else {
        const double f = fcol[buf][lane];
        const double pv = fcol[buf][p];
        if (lane == p) { used = true; myk = k; }
        if (lane == k) mypiv = pv;
        const double rinv = 1.0 / pv;
#pragma unroll
        for (int q = 0; q < CW; ++q) {
          const int c = 4 * q + w;
          if (c == k) pcw[q] = p;
          if (c != k && c < n) {
            const double pr = gj_bcast(m[q], p) * rinv;
            m[q] = (lane == p) ? pr : fma(-f, pr, m[q]);
          }
        }
        if (w == own) m[qk] = (lane == p) ? rinv : -f * rinv;
      }
    }
  }
  if (bad) {                                       // (uniform over the workgroup)
    if (tid == 0) { atomicExch(fail, node + 1); logdet[node] = 0.0; }
    return;
  }
  double* const M = sinv + (long)node * n * n;
  if (row) {
#pragma unroll
    for (int q = 0; q < CW; ++q) {
      const int c = 4 * q + w;
      if (c < n) { M[(long)myk * n + pcw[q]] = m[q]; ainv[myk * n + pcw[q]] = m[q]; }
    }
  }
  if (w == 0) {
    const double lg = log(fabs(mypiv));
    double ld = 0.0;
    for (int k = 0; k < n; ++k) ld += gj_bcast(lg, k);
    if (lane == 0) logdet[node] = ld;
  }
  __syncthreads();
  // ---- (3) Tout = S^-1 Tsum[:, 0:Cmm]: hodlr_mm_kernel's 32 x 64 tile (wavefront w: 16-row block w & 1, 16-column blocks
  //          2 (w >> 1), 2 (w >> 1) + 1), one 32-deep k block (2R <= 32)
  typedef double cm_v4d __attribute__((ext_vector_type(4)));
  const int fr = lane & 15, fk = lane >> 4, bi = w & 1, bj = 2 * (w >> 1);
  for (int c0 = 0; c0 < Cmm; c0 += 64) {
    cm_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int ar = 16 * bi + fr, k = 4 * kk + fk;
      const double av = (ar < n && k < n) ? ainv[ar * n + k] : 0.0;
      const int cA = c0 + 16 * bj + fr, cB = cA + 16;
      const double b0 = (k < n && cA < Cmm) ? ts[k * TP + cA] : 0.0;
      const double b1 = (k < n && cB < Cmm) ? ts[k * TP + cB] : 0.0;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rw = 16 * bi + fk + 4 * r;
      if (rw >= n) continue;
      double* const o = Tout + ((long)node * n + rw) * Cp + c0 + 16 * bj + fr;
      if (c0 + 16 * bj + fr < Cmm) o[0] = acc0[r];
      if (c0 + 16 * bj + 16 + fr < Cmm) o[16] = acc1[r];
    }
  }
}
static int g_hodlr_core_fused = 1;
extern "C" int gh_debug_set_hodlr_core_fused(int on) {
  const int prev = g_hodlr_core_fused;
  g_hodlr_core_fused = on ? 1 : 0;
  return prev;
}

static int batched_inverse(gh_hodlr* h, double* base, const std::vector<long>& offs, const std::vector<int>& sizes,
                           double* d_logdet, GhBuf* const* tables = nullptr, bool tables_valid = false,
                           const double* tsum = nullptr, int tsum_R = 0) {
  const int nb = (int)sizes.size();
  if (nb == 0) return GH_OK;
  std::vector<long> sc(nb);
  long tot = 0;
  for (int i = 0; i < nb; ++i) { sc[i] = tot; tot += sizes[i]; }
  GhPooledBuf l_offs, l_sizes, l_sc, d_sd, d_si;
  GhBuf& d_offs = tables ? *tables[0] : (GhBuf&)l_offs;
  GhBuf& d_sizes = tables ? *tables[1] : (GhBuf&)l_sizes;
  GhBuf& d_sc = tables ? *tables[2] : (GhBuf&)l_sc;
  if (!tables || !tables_valid) {
    GH_CHECK(upload(d_offs, offs, h->st));
    GH_CHECK(upload(d_sizes, sizes, h->st));
    GH_CHECK(upload(d_sc, sc, h->st));
  }
  int nmax = 0;
  for (int v : sizes) nmax = std::max(nmax, v);
  if (nmax <= 32 && nb <= 64) {   // few cores (the top levels): a workgroup per matrix, columns over its four wavefronts
    if (nmax <= 16) hipLaunchKernelGGL(gj_small4_kernel<16>, dim3(nb), dim3(256), 0, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p, nb, d_logdet, (int*)h->flags.p, tsum, (long)h->cpass, tsum_R);
    else hipLaunchKernelGGL(gj_small4_kernel<32>, dim3(nb), dim3(256), 0, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p, nb, d_logdet, (int*)h->flags.p, tsum, (long)h->cpass, tsum_R);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  if (nmax <= 32) {                  // the Woodbury cores: one wavefront per matrix
    const dim3 grid((unsigned)((nb + 3) / 4));
    if (nmax <= 8) hipLaunchKernelGGL(gj_small_kernel<8>, grid, dim3(256), 0, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p, nb, d_logdet, (int*)h->flags.p, tsum, (long)h->cpass, tsum_R);
    else if (nmax <= 16) hipLaunchKernelGGL(gj_small_kernel<16>, grid, dim3(256), 0, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p, nb, d_logdet, (int*)h->flags.p, tsum, (long)h->cpass, tsum_R);
    else hipLaunchKernelGGL(gj_small_kernel<32>, grid, dim3(256), 0, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p, nb, d_logdet, (int*)h->flags.p, tsum, (long)h->cpass, tsum_R);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  if (tsum) {                                     // (cores too big for the wavefront kernel: build them first)
    hipLaunchKernelGGL(hodlr_sbuild_kernel, dim3(nb), dim3(256), 0, h->st, tsum, (long)h->cpass, tsum_R, base);
    GH_HIP(hipGetLastError());
  }
  GH_CHECK(d_sd.ensure(tot * sizeof(double)));
  GH_CHECK(d_si.ensure(tot * sizeof(int)));
  // dynamic LDS for the in-LDS path: the largest matrix of the batch if it fits (<= 144 KiB), else none
  size_t lds_bytes = ((size_t)nmax * (nmax | 1) + nmax) * sizeof(double);
  if (lds_bytes > 144 * 1024) lds_bytes = 0;
  if (lds_bytes > 0) {
    static bool attr_set = false;
    if (!attr_set) {
      GH_HIP(hipFuncSetAttribute((const void*)gj_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(gj_inverse_kernel, dim3(nb), dim3(256), lds_bytes, h->st, base, (const long*)d_offs.p, (const int*)d_sizes.p,
                     d_sd.d(), (int*)d_si.p, (const long*)d_sc.p, d_logdet, (int*)h->flags.p, (int)(lds_bytes / sizeof(double)));
  GH_HIP(hipGetLastError());
  return GH_OK;
}

extern "C" int gh_hodlr_compute(gh_hodlr* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                                const double* yerr, double* logdet_out) {
  if (!h || !k || !x || !yerr || n <= 0) { gh_set_error("bad argument to compute"); return GH_ERR_BAD_ARG; }
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  if (n > 0x3fffffffL) { gh_set_error("HODLR: n too large"); return GH_ERR_BAD_ARG; }
  GH_HIP(hipSetDevice(h->opts.device));
  GH_CHECK(k->upload());
  hipStream_t st = h->st;
  // phase stamps on stderr (how the stalls of the split tree were found): a build-time aid, -DGH_HODLR_PHASE_MARKS
#ifdef GH_HODLR_PHASE_MARKS
  const auto dbg_t0 = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    fprintf(stderr, "[hodlr %p] %s at %.2f ms\n", (void*)h, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count());
  };
#else
  auto mark = [](const char*) {};
#endif
  h->computed = false;
  h->n = n; h->ndim = ndim;
  GH_CHECK(h->x.ensure((size_t)n * ndim * sizeof(double)));
  GH_CHECK(h->yerr.ensure((size_t)n * sizeof(double)));
  GH_CHECK(gh_to_device(h->x.d(), x, (size_t)n * ndim, st));
  GH_CHECK(gh_to_device(h->yerr.d(), yerr, (size_t)n, st));
  GH_CHECK(h->scal.ensure(64));
  mark("inputs enqueued");

  // ---- tree (hodlr.h:47-64), breadth first; kept from the previous compute() when n and min_size are the same
  const int min_size = h->opts.min_size;
  const int l0 = h->sub.depth;                   // levels [0, l0) are the pseudo-levels of a sub-tree handle (HSub)
  if (l0 > 0 && ((int)h->sub.half.size() != l0 || (int)h->sub.R.size() != l0 || (int)h->sub.T.size() != l0 || !h->sub.allreduce)) {
    gh_set_error("HODLR: incomplete sub-tree description"); return GH_ERR_BAD_ARG;
  }
  if (h->tree_n != n || h->tree_min != min_size || h->tree_sub != h->sub.sig()) {
    h->reset_tree();
    h->nodes.push_back({0, (int)n, (int)n / 2, l0, 0});
    for (size_t q = 0; q < h->nodes.size(); ++q) {
      HNode nd = h->nodes[q];
      if (nd.half >= min_size) {
        h->nodes.push_back({nd.start, nd.half, nd.half / 2, nd.level + 1, 0});
        h->nodes.push_back({nd.start + nd.half, nd.size - nd.half, (nd.size - nd.half) / 2, nd.level + 1, 0});
        if ((int)h->levels.size() <= nd.level) h->levels.resize(nd.level + 1, nullptr);
        if (!h->levels[nd.level]) h->levels[nd.level] = new HLevel();
        h->levels[nd.level]->node_ids.push_back((int)q);
      } else {
        h->nodes[q].is_leaf = 1;
        long off = h->leaves.empty() ? 0 : h->leaves.back().off + (long)h->leaves.back().size * h->leaves.back().size;
        h->leaves.push_back({nd.start, nd.size, off});
      }
    }
    // pseudo-levels: the ancestor at level l, cut down to the local rows -- one of its halves is empty here
    if ((int)h->levels.size() < l0) h->levels.resize(l0, nullptr);
    for (int l = 0; l < l0; ++l) {
      h->levels[l] = new HLevel();
      h->levels[l]->top = true;
      h->levels[l]->top_level = l;
      h->levels[l]->node_ids.push_back((int)h->nodes.size());
      h->nodes.push_back({0, (int)n, h->sub.half[l] == 0 ? (int)n : 0, l, 0});
    }
    h->tree_n = n; h->tree_min = min_size; h->tree_sub = h->sub.sig();
  }
  h->max_leaf = 0;
  for (auto& lf : h->leaves) h->max_leaf = std::max(h->max_leaf, lf.size);

  // ---- leaves: exact blocks -> explicit inverses + log-dets (hodlr.h:223-227, 87-89)
  // log|det| of every factored block (leaves, then the cores level by level) is collected in ld_all
  // on the device and summed on the host after the ONE synchronisation that ends compute(); failure
  // flags likewise (flags[0]: singular Gauss-Jordan block, flags[2..3]: leaf Cholesky info).
  size_t n_blocks = h->leaves.size();
  for (auto* L : h->levels) n_blocks += L->node_ids.size();
  GH_CHECK(h->ld_all.ensure(std::max<size_t>(n_blocks, 1) * sizeof(double)));
  GH_CHECK(h->flags.ensure(4 * sizeof(int)));
  GH_HIP(hipMemsetAsync(h->ld_all.p, 0, std::max<size_t>(n_blocks, 1) * sizeof(double), st));
  GH_HIP(hipMemsetAsync(h->flags.p, 0, 4 * sizeof(int), st));
  size_t ld_at = 0;
  // The leaf stage (build, batched Cholesky + inverse, K^-1 = L^-T L^-1: 1.4 ms at C4) depends on
  // nothing the ACA produces, so it is issued on the second stream under the ACA of the top levels.
  GhPooledBuf linv;                              // (lives until the final synchronisation: two streams touch it)
  GhPooledBuf lstk, l22b, lwk;                   // (the 129..256-row leaf path's work blocks: as linv)
  bool leaves_done = false;
  auto leaf_stage = [&](hipStream_t st) -> int {
    struct StreamSwap { gh_hodlr* h; hipStream_t keep; ~StreamSwap() { h->st = keep; } } swap_guard{h, h->st};
    h->st = st;                                  // (launch_mm / batched_inverse issue on h->st)
  if (h->max_leaf <= 128) {
    // Leaves are symmetric positive definite and fit the dense solver's 128 x 128 diagonal-block
    // kernel: build them identity-padded into 128 x 128 slots, factor + invert the factors as ONE
    // batched launch of potf2_inv_mfma_kernel (79 us per block, a workgroup each), log-det from the
    // factor's diagonal, K^-1 = L^-T L^-1 as one batched product.  (Gauss-Jordan with pivoting, the
    // general path below, spends 7 ms on the 2048 leaves of C4; this one ~1.5 ms.)
    const int nl = (int)h->leaves.size();
    const size_t slot = (size_t)128 * 128;
    GH_CHECK(h->leaf_inv.ensure(nl * slot * sizeof(double)));
    long long* d_info = (long long*)((int*)h->flags.p + 2);
    if (!h->leaf_tab_up) GH_CHECK(upload(h->d_leaves, h->leaves, st));
    // K_leaf^-1 = L^-T L^-1 and log|K_leaf| in ONE launch per batch: gh_potf2.hip, potf2_kinv_kernel (round 6; it was the build + the
    // batched factorisation + a log-det kernel + a batched transpose + a batched product: five launches, 2.3 GB through the L2s).
    // Kernels of the a + b F(r^2) form are evaluated INSIDE that launch; the others keep the build launch in front of it.
    static_assert(sizeof(LeafDesc) == 16, "potf2_kinv_kernel reads LeafDesc as {int start, size; long off}");
    if (k->fast.ok && g_hodlr_leaf_fused) {
      GH_CHECK(gh_launch_potf2_kinv_kernel_batched(h->leaf_inv.d(), 128, (int64_t)slot, h->ld_all.d() + ld_at, d_info, nl, k->fast,
                                                   h->x.d(), h->yerr.d(), ndim, h->d_leaves.p, st));
    } else {
      hipLaunchKernelGGL(hodlr_leaf_build_kernel, dim3(nl, 8), dim3(256), 0, st, k->d_nodes, (int)k->nodes.size(), k->fast, ndim,
                         h->x.d(), h->yerr.d(), (const LeafDesc*)h->d_leaves.p, h->leaf_inv.d(), 128);
      GH_HIP(hipGetLastError());
      GH_CHECK(gh_launch_potf2_kinv_batched(h->leaf_inv.d(), 128, (int64_t)slot, h->ld_all.d() + ld_at, d_info, nl, st));
    }
    ld_at += nl;
    {
      std::vector<MMJob> jobs(nl);
      for (int i = 0; i < nl; ++i) jobs[i] = {(long)(i * slot), h->leaves[i].start, h->leaves[i].start, h->leaves[i].size, h->leaves[i].size};
      if (!h->leaf_tab_up) {
        GH_CHECK(upload(h->d_leaf_jobs, jobs, st));
        h->leaf_tab_up = true;
      }
    }
    h->leaf_pitch = 128;
  } else if (h->max_leaf <= 256 && !g_hodlr_leaf_gj) {
    // Leaves of 129 .. 256 rows -- the reference's tree stops splitting below 2 min_size, so with min_size = 100 most problem
    // sizes have leaves of up to 199 rows (N = 50000: 256 leaves of 195 / 196) -- went through the pivoted Gauss-Jordan in place
    // in HBM: 14 of the 17 ms of a step at N = 50000 (round 5 profile).  Same recipe as above on 256 x 256 identity-padded slots,
    // as 2 x 2 blocks of 128 with the batched kernels there are:  L11 = chol(A11);  W = L21^T = L11^-1 A12;  A22 -= W^T W;
    // L22 = chol(A22);  L^-1 = [[L11^-1, 0], [X, L22^-1]],  X = -L22^-1 L21 L11^-1;  K^-1 = L^-T L^-1 block by block
    // (the full symmetric matrix is stored: the products read rows and, by symmetry, columns).
    const int nl = (int)h->leaves.size();
    const size_t slot = (size_t)256 * 256, blk = (size_t)128 * 128;
    GH_CHECK(h->leaf_inv.ensure(nl * slot * sizeof(double)));
    GH_CHECK(lstk.ensure(nl * 2 * blk * sizeof(double)));        // per leaf Z = [L11^-T | X'^T]  (128 x 256), X' = L22^-1 L21 L11^-1 = -X
    GH_CHECK(l22b.ensure(nl * 2 * blk * sizeof(double)));        // L22^-1, L22^-T
    GH_CHECK(lwk.ensure(nl * 2 * blk * sizeof(double)));         // L21, T^T
    long long* d_info = (long long*)((int*)h->flags.p + 2);
    if (!h->leaf_tab_up) GH_CHECK(upload(h->d_leaves, h->leaves, st));
    hipLaunchKernelGGL(hodlr_leaf_build_kernel, dim3(nl, 16), dim3(256), 0, st, k->d_nodes, (int)k->nodes.size(), k->fast, ndim,
                       h->x.d(), h->yerr.d(), (const LeafDesc*)h->d_leaves.p, h->leaf_inv.d(), 256);
    GH_HIP(hipGetLastError());
    double* const S = h->leaf_inv.d();
    std::vector<MMJob> jobs(nl);
    for (int i = 0; i < nl; ++i) jobs[i] = {(long)(i * slot), h->leaves[i].start, h->leaves[i].start, h->leaves[i].size, h->leaves[i].size};
    if (!h->leaf_tab_up) {
      GH_CHECK(upload(h->d_leaf_jobs, jobs, st));
      h->leaf_tab_up = true;
    }
    // work blocks per leaf: Z = [L11^-T | X'^T] (128 x 256, k contiguous), L11^-1, L21 then T^T, L22^-1, L22^-T
    GH_CHECK(linv.ensure(nl * blk * sizeof(double)));            // L11^-1 (row-major, from the factorisation kernel)
    double* const Z = lstk.d();                                  // pitch 256
    double* const W = lwk.d();                                   // block 0: L21, block 1: T^T   (pitch 128)
    const long s2 = (long)slot, sZ = (long)(2 * blk), sW = (long)(2 * blk), sb = (long)blk;
#define GH_BMM(ACC, C_, ldc_, sc_, A_, lda_, sa_, B_, ldb_, sb_, K_)                                                              \
    hipLaunchKernelGGL(hodlr_bmm_nt_kernel<ACC>, dim3(nl), dim3(256), 0, st, C_, (long)(ldc_), (long)(sc_), (const double*)(A_), (long)(lda_), (long)(sa_), \
                       (const double*)(B_), (long)(ldb_), (long)(sb_), (long)(K_))
    GH_CHECK(gh_launch_potf2_batched(S, 256, (int64_t)slot, linv.d(), (int64_t)blk, d_info, nl, st));                             // L11 in place, L11^-1
    hipLaunchKernelGGL(hodlr_transpose128_kernel, dim3(nl, 16), dim3(256), 0, st, (const double*)linv.d(), 128L, sb, Z, 256L, sZ);    // Z[:, 0:128] = L11^-T
    GH_BMM(false, W, 128, sW, S + 128 * 256, 256, s2, linv.d(), 128, sb, 128);                                                     // L21 = A21 L11^-T
    GH_BMM(true, S + 128 * 256 + 128, 256, s2, W, 128, sW, W, 128, sW, 128);                                                       // A22 -= L21 L21^T
    GH_CHECK(gh_launch_potf2_batched(S + 128 * 256 + 128, 256, (int64_t)slot, l22b.d(), (int64_t)(2 * blk), d_info, nl, st));     // L22 in place, L22^-1
    hipLaunchKernelGGL(hodlr_leaf_logdet256_kernel, dim3(nl), dim3(256), 0, st, (const double*)S, h->ld_all.d() + ld_at);
    ld_at += nl;
    hipLaunchKernelGGL(hodlr_transpose128_kernel, dim3(nl, 16), dim3(256), 0, st, (const double*)l22b.d(), 128L, sZ, l22b.d() + blk, 128L, sZ);   // L22^-T
    GH_BMM(false, W + blk, 128, sW, Z, 256, sZ, W, 128, sW, 128);                                                                   // T^T = L11^-T L21^T   (T = L21 L11^-1)
    GH_BMM(false, Z + 128, 256, sZ, W + blk, 128, sW, l22b.d(), 128, sZ, 128);                                                      // X'^T = T^T L22^-T    (X' = L22^-1 T = -X)
    GH_HIP(hipMemsetAsync(S, 0, nl * slot * sizeof(double), st));                                                                  // (both factors are used up; K21, K12 are formed by subtraction)
    GH_BMM(false, S, 256, s2, Z, 256, sZ, Z, 256, sZ, 256);                                                                         // K11 = L11^-T L11^-1 + X^T X
    GH_BMM(false, S + 128 * 256 + 128, 256, s2, l22b.d() + blk, 128, sZ, l22b.d() + blk, 128, sZ, 128);                             // K22 = L22^-T L22^-1
    GH_BMM(true, S + 128 * 256, 256, s2, l22b.d() + blk, 128, sZ, Z + 128, 256, sZ, 128);                                           // K21 = L22^-T X = -L22^-T X'
    GH_BMM(true, S + 128, 256, s2, Z + 128, 256, sZ, l22b.d() + blk, 128, sZ, 128);                                                 // K12 = K21^T
#undef GH_BMM
    GH_HIP(hipGetLastError());
    h->leaf_pitch = 256;
  } else {
  {
    const int nl = (int)h->leaves.size();
    const long tot = h->leaves.back().off + (long)h->leaves.back().size * h->leaves.back().size;
    GH_CHECK(h->leaf_inv.ensure(tot * sizeof(double)));
    GH_CHECK(upload(h->d_leaves, h->leaves, st));
    hipLaunchKernelGGL(hodlr_leaf_build_kernel, dim3(nl, 8), dim3(256), 0, st, k->d_nodes, (int)k->nodes.size(), k->fast, ndim,
                       h->x.d(), h->yerr.d(), (const LeafDesc*)h->d_leaves.p, h->leaf_inv.d(), 0);
    GH_HIP(hipGetLastError());
    std::vector<long> offs(nl);
    std::vector<int> sizes(nl);
    std::vector<MMJob> jobs(nl);
    for (int i = 0; i < nl; ++i) {
      offs[i] = h->leaves[i].off; sizes[i] = h->leaves[i].size;
      jobs[i] = {h->leaves[i].off, h->leaves[i].start, h->leaves[i].start, h->leaves[i].size, h->leaves[i].size};
    }
    GH_CHECK(upload(h->d_leaf_jobs, jobs, st));
    GH_CHECK(batched_inverse(h, h->leaf_inv.d(), offs, sizes, h->ld_all.d() + ld_at));
    ld_at += nl;
  }
  // leaf job rows use a per-job A stride = its own size: encode through a_rs = 0 -> handled below
  // (hodlr_mm_kernel takes one a_rs per launch, so leaves are launched with a_rs = max_leaf after
  //  re-packing: simpler -- store every leaf inverse with row pitch max_leaf)
  // NOTE: leaf inverses were produced with pitch == size; repack to pitch max_leaf when sizes differ.
  {
    bool uniform = true;
    for (auto& lf : h->leaves) if (lf.size != h->max_leaf) uniform = false;
    if (!uniform) {
      const int nl = (int)h->leaves.size(), ml = h->max_leaf;
      GhBuf packed;
      GH_CHECK(packed.ensure((size_t)nl * ml * ml * sizeof(double)));
      GH_HIP(hipMemsetAsync(packed.p, 0, (size_t)nl * ml * ml * sizeof(double), st));
      std::vector<MMJob> jobs(nl);
      for (int i = 0; i < nl; ++i) {
        const LeafDesc& lf = h->leaves[i];
        GH_HIP(hipMemcpy2DAsync(packed.d() + (size_t)i * ml * ml, ml * sizeof(double), h->leaf_inv.d() + lf.off,
                                lf.size * sizeof(double), lf.size * sizeof(double), lf.size, hipMemcpyDeviceToDevice, st));
        jobs[i] = {(long)i * ml * ml, lf.start, lf.start, lf.size, lf.size};
      }
      GH_HIP(hipStreamSynchronize(st));
      std::swap(h->leaf_inv.p, packed.p);
      std::swap(h->leaf_inv.bytes, packed.bytes);
      GH_CHECK(upload(h->d_leaf_jobs, jobs, st));
    }
  }
    h->leaf_pitch = h->max_leaf;
  }

    leaves_done = true;
    return GH_OK;
  };

  // ---- ACA of every level into column-major scratch, ranks back to the host
  // Column capacity of the scratch: the caller's cap, else 256 to start with (doubled, up to RANK_CAP,
  // for a level one of whose blocks is cut short by it -- that level is then redone).
  // The levels are independent, so they are all ENQUEUED before the host looks at any result: the
  // clustered levels (several workgroups per node, spin barriers: they must not share the chip with
  // another spinning grid) one after the other on the solver's stream, the one-workgroup-per-node
  // levels beside them on a second stream; one synchronisation instead of one per level (each cost
  // a ~60 us bubble, and levels 8-10 of C4 -- 1.4 ms -- now run under levels 0-7).  Every level
  // gets its own n x rcap scratch; if that is more than 12 GiB in total the levels share one and go
  // one at a time.
  const bool user_cap = h->opts.max_rank > 0;
  const int rcap0 = user_cap ? h->opts.max_rank : std::min(256, RANK_CAP);
  const int nlev = (int)h->levels.size();
  const int aca_fence = 0, aca_multi = 3;   // bit 0: batched candidate search (one-workgroup nodes); bit 1: clusters draw the next row before the norms barrier         // (fence-free cluster barrier, 8 then 64 candidate rows per search pass: DESIGN.md section 4)
  // (round 5: "more than 12 GiB" was a 64-GB-card habit; an MI355X has 288 GB.  Above 12 GiB the question is put to the device:
  //  all levels at once while their scratch fits in 40 % of what is free now -- N = 700000 went one level at a time, 34 ms)
  bool concurrent = nlev - l0 > 1;
  {
    const double need = (double)n * rcap0 * sizeof(double) * (nlev - l0);
    if (concurrent && need > 12.0 * (1u << 30)) {
      size_t free_b = 0, tot_b = 0;
      if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
      // (blocks parked in the pool by the previous compute() of this handle count as used there and are what will be handed out again)
      concurrent = need <= 0.4 * (double)free_b + (double)gh_pool_parked_bytes();
    }
  }
  if (concurrent && !h->st_b) {
    hipStream_t shq[4] = {nullptr, nullptr, nullptr, nullptr};
    if (h->shared_streams && gh_shared_streams(h->opts.device, shq)) h->st_b = shq[2];
    if ((!h->shared_streams && hipStreamCreateWithFlags(&h->st_b, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); h->st_b = nullptr; }
  }
  struct AcaLevel { GhPooledBuf Tcm, idx, sync, part; int G = 1, rcap = 0; int flags[2] = {0, 0}; char* syncp = nullptr; bool sync_cleared = false;
                    bool timed = false; };   // timed: its nodes wrote their durations behind the two flags (the one-workgroup launch)
  GhPooledBuf sync_all;                                  // the levels' barrier counters / selections / flags: one buffer, ONE memset
  std::vector<AcaLevel> al(nlev);
  GhPooledBuf shared_Tcm;
  std::vector<GhBuf*> levelB(nlev, nullptr);
  struct Cleanup { std::vector<GhBuf*>& v; ~Cleanup() { for (auto* b : v) delete b; } } cleanup{levelB};
  const int pstride = 8 + 2 * ACA_MAXR;
  // 0: level l goes through the workgroup kernel; 64 / 128 / 256: every block of the level has at most that many rows and
  // columns and the level goes through hodlr_aca_wave_kernel
  if ((int)h->wave_bad.size() != nlev) h->wave_bad.assign(nlev, 0);
  auto wave_mr = [&](int l) -> int {
    if (!g_hodlr_wave_aca || h->wave_bad[l] || h->levels[l]->top || al[l].G != 1) return 0;
    int mx = 0;
    for (int id : h->levels[l]->node_ids) { const HNode& nd = h->nodes[id]; mx = std::max(mx, std::max(nd.half, nd.size - nd.half)); }
    // (the interpreter's registers beside 128 of mirrors would spill: kernels off the a + b F(r^2) form stop at 128 x 128)
    return mx <= 64 ? 64 : mx <= 128 ? 128 : (mx <= 256 && k->fast.ok) ? 256 : 0;
  };
  // buffers of level l for column capacity rc, counters cleared on stream sx
  auto prepare_level = [&](int l, int rc, hipStream_t sx) -> int {
    HLevel* L = h->levels[l];
    AcaLevel& a = al[l];
    const int nn = (int)L->node_ids.size();
    a.rcap = rc;
    GhBuf& T = (concurrent ? (GhBuf&)a.Tcm : (GhBuf&)shared_Tcm);
    GH_CHECK(T.ensure((size_t)n * rc * sizeof(double)));
    GH_CHECK(a.idx.ensure((size_t)n * sizeof(int)));
    GH_CHECK(a.part.ensure((size_t)nn * a.G * pstride * sizeof(double)));
    if (a.sync_cleared) {                                  // (its slice of sync_all was cleared with all the others)
      a.sync_cleared = false;
    } else {
      GH_CHECK(a.sync.ensure((size_t)nn * (sizeof(unsigned) + 2 * sizeof(int)) + 2 * sizeof(int)));
      a.syncp = (char*)a.sync.p;
      GH_HIP(hipMemsetAsync(a.syncp, 0, (size_t)nn * (sizeof(unsigned) + 2 * sizeof(int)) + 2 * sizeof(int), sx));
    }
    return GH_OK;
  };
  // enqueue the ACA of level l with column capacity rc on stream sx (no synchronisation)
  auto enqueue_level = [&](int l, int rc, hipStream_t sx) -> int {
    HLevel* L = h->levels[l];
    AcaLevel& a = al[l];
    const int nn = (int)L->node_ids.size();
    GH_CHECK(prepare_level(l, rc, sx));
    GH_CHECK(aca_lds_attr());
    GhBuf& T = (concurrent ? (GhBuf&)a.Tcm : (GhBuf&)shared_Tcm);
    unsigned* d_bars = (unsigned*)a.syncp;
    int* d_sel = (int*)(d_bars + nn);
    int* d_fail = d_sel + nn;
    if (const int mr = wave_mr(l)) {                       // blocks of <= 256 x 256: a wavefront per node
#define GH_ACA_WAVE(F, EE)                                                                                        \
      hipLaunchKernelGGL((hodlr_aca_wave_kernel<F, EE>), dim3((nn + AW_NODES - 1) / AW_NODES), dim3(64 * AW_NODES), 0, sx, \
                         k->d_nodes, (int)k->nodes.size(), k->fast, ndim, h->x.d(), (const LvlNode*)L->d_nodes.p, nn, T.d(), (long)n, rc, \
                         (int*)L->d_ranks.p, h->opts.tol, (unsigned long long)(unsigned)h->opts.seed, l, d_fail + 1)
      if (k->fast.ok) { if (mr == 64) GH_ACA_WAVE(true, 1); else if (mr == 128) GH_ACA_WAVE(true, 2); else GH_ACA_WAVE(true, 4); }
      else            { if (mr == 64) GH_ACA_WAVE(false, 1); else GH_ACA_WAVE(false, 2); }
#undef GH_ACA_WAVE
      GH_HIP(hipGetLastError());
      return GH_OK;
    }
#define GH_ACA_LAUNCH(F, CLU)                                                                                          \
    hipLaunchKernelGGL((hodlr_aca_kernel<F, CLU>), dim3(nn * a.G), dim3(ACA_THREADS), a.G == 1 ? ACA_DYN_BYTES : 0, sx, k->d_nodes, (int)k->nodes.size(),  \
                       k->fast, ndim, h->x.d(), (const LvlNode*)L->d_nodes.p, T.d(), (long)n, rc, (int*)a.idx.p,     \
                       (int*)L->d_ranks.p, h->opts.tol, (unsigned long long)(unsigned)h->opts.seed, l,               \
                       a.G, d_bars, a.part.d(), pstride, d_sel, d_fail, aca_multi, aca_fence, d_fail + 1,            \
                       (const AcaSeg*)nullptr, 0, a.G == 1 ? ACA_CAPD : 0)
    if (a.G == 1) { if (k->fast.ok) GH_ACA_LAUNCH(true, false); else GH_ACA_LAUNCH(false, false); }
    else          { if (k->fast.ok) GH_ACA_LAUNCH(true, true); else GH_ACA_LAUNCH(false, true); }
#undef GH_ACA_LAUNCH
    GH_HIP(hipGetLastError());
    return GH_OK;
  };
  // all clustered levels `cl` as ONE launch of at most 256 workgroups (al[l].G already balanced)
  auto enqueue_fused = [&](const std::vector<int>& cl, int rc, hipStream_t sx, GhBuf& segbuf, int threads = ACA_THREADS) -> int {
    std::vector<AcaSeg> segs;
    int wg = 0;
    bool ones_only = true;                                 // (one-workgroup nodes only: the launch carries the LDS mirrors)
    for (int l : cl) ones_only = ones_only && al[l].G == 1;
    GH_CHECK(aca_lds_attr());
    for (int l : cl) {
      HLevel* L = h->levels[l];
      AcaLevel& a = al[l];
      const int nn = (int)L->node_ids.size();
      GH_CHECK(prepare_level(l, rc, sx));
      unsigned* d_bars = (unsigned*)a.syncp;
      int* d_sel = (int*)(d_bars + nn);
      int* d_fail = d_sel + nn;
      int* d_dur = nullptr;
      const int* d_order = nullptr;
      if (ones_only && g_hodlr_lpt) {
        // The nodes of a level differ in cost by 4x (C4, level 8: 205 us on average, ~800 us for the few whose search runs dry
        // first), and a long one dispatched late is the end of phase 1: every node reports how long it took, and the next
        // compute() of the handle launches the level longest first.  Inside an optimiser loop the costs hardly move.
        d_dur = d_fail + 2;
        a.timed = true;
        if ((int)L->aca_dur.size() == nn) {
          std::vector<int> ord(nn);
          for (int q = 0; q < nn; ++q) ord[q] = q;
          std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return L->aca_dur[x] > L->aca_dur[y]; });
          GH_CHECK(upload(L->d_order, ord, sx));
          d_order = (const int*)L->d_order.p;
        }
      }
      segs.push_back({(const LvlNode*)L->d_nodes.p, a.Tcm.d(), (int*)a.idx.p, (int*)L->d_ranks.p, d_bars, a.part.d(), d_sel,
                      d_fail, d_fail + 1, l, a.G, wg, nn * a.G, d_dur, d_order});
      wg += nn * a.G;
    }
    GH_CHECK(upload(segbuf, segs, sx));
#define GH_ACA_LAUNCH(F, CLU)                                                                                          \
    hipLaunchKernelGGL((hodlr_aca_kernel<F, CLU>), dim3(wg), dim3(threads), ones_only ? ACA_DYN_BYTES : 0, sx, k->d_nodes, (int)k->nodes.size(),    \
                       k->fast, ndim, h->x.d(), (const LvlNode*)nullptr, (double*)nullptr, (long)n, rc, (int*)nullptr, \
                       (int*)nullptr, h->opts.tol, (unsigned long long)(unsigned)h->opts.seed, 0,                     \
                       1, (unsigned*)nullptr, (double*)nullptr, pstride, (int*)nullptr, (int*)nullptr, aca_multi, aca_fence, \
                       (int*)nullptr, (const AcaSeg*)segbuf.p, (int)segs.size(), ones_only ? ACA_CAPD : 0)
    if (ones_only) { if (k->fast.ok) GH_ACA_LAUNCH(true, false); else GH_ACA_LAUNCH(false, false); }
    else           { if (k->fast.ok) GH_ACA_LAUNCH(true, true); else GH_ACA_LAUNCH(false, true); }
#undef GH_ACA_LAUNCH
    GH_HIP(hipGetLastError());
    return GH_OK;
  };
  // flags and ranks of level l back to the host (a device-to-host copy into pageable memory holds the
  // host until the stream gets there, so these are issued only after EVERY level has been enqueued)
  auto fetch_level = [&](int l, hipStream_t sx) -> int {
    HLevel* L = h->levels[l];
    AcaLevel& a = al[l];
    const int nn = (int)L->node_ids.size();
    int* d_fail = (int*)((unsigned*)a.syncp + nn) + nn;
    L->ranks.resize(nn);
    GH_HIP(hipMemcpyAsync(a.flags, d_fail, 2 * sizeof(int), hipMemcpyDeviceToHost, sx));
    GH_HIP(hipMemcpyAsync(L->ranks.data(), L->d_ranks.p, nn * sizeof(int), hipMemcpyDeviceToHost, sx));
    return GH_OK;
  };
  // after a synchronisation: validate level l, redo it with more columns while a block is cut short
  auto settle_level = [&](int l) -> int {
    AcaLevel& a = al[l];
    for (;;) {
      if (a.flags[0]) { gh_set_error("HODLR: cluster barrier of the ACA kernel timed out at level %d", l); return GH_ERR_HIP; }
      if (!a.flags[1]) return GH_OK;
      if (a.flags[1] >= 2) {
        // a block of this level asked the wavefront-per-node kernel for more than AW_RW_OF(E) columns: the level again, with the
        // workgroup kernel (same draws, same results), and the handle remembers it for its next compute()
        h->wave_bad[l] = 1;
        GH_CHECK(enqueue_level(l, a.rcap, st));
        GH_CHECK(fetch_level(l, st));
        GH_HIP(hipStreamSynchronize(st));
        continue;
      }
      // hodlr.h:147 lets the rank grow to min(rows, cols); a cut-short block would be a silently wrong answer
      if (user_cap || a.rcap >= RANK_CAP) {
        gh_set_error("HODLR: an off-diagonal block of level %d needs a rank above %d to reach tol = %g (%s); "
                     "the factorisation is not usable", l, a.rcap, h->opts.tol,
                     user_cap ? "opts.max_rank" : "the solver's ceiling: loosen tol, raise min_size or use the dense solver");
        return user_cap ? GH_ERR_BAD_ARG : GH_ERR_RANK;
      }
      GH_CHECK(enqueue_level(l, std::min(2 * a.rcap, RANK_CAP), st));
      GH_CHECK(fetch_level(l, st));
      GH_HIP(hipStreamSynchronize(st));
    }
  };
  auto rank_of_level = [&](int l) {
    HLevel* L = h->levels[l];
    if (L->top) L->ranks.assign(1, h->sub.R[l]);         // (given: the ACA of the ancestor ran elsewhere)
    L->R = 0;
    for (int r : L->ranks) L->R = std::max(L->R, r);
    L->off = h->Rtot;
    h->Rtot += L->R;
    h->maxR = std::max(h->maxR, L->R);
  };
  // serial mode only (one scratch shared by the levels): park the level's factors in a compact buffer
  auto compact_level = [&](int l) -> int {
    HLevel* L = h->levels[l];
    const int nn = (int)L->node_ids.size();
    rank_of_level(l);
    if (L->R > 0) {
      levelB[l] = new GhPooledBuf();
      GH_CHECK(levelB[l]->ensure((size_t)n * L->R * sizeof(double)));
      GH_HIP(hipMemsetAsync(levelB[l]->p, 0, (size_t)n * L->R * sizeof(double), st));
      hipLaunchKernelGGL(hodlr_compact_kernel, dim3(nn, std::max(8, std::min(512, 2048 / nn))), dim3(256), 0, st, shared_Tcm.d(), (long)n, (const LvlNode*)L->d_nodes.p,
                         (const int*)L->d_ranks.p, L->R, levelB[l]->d(), (long)L->R, 0L, (double*)nullptr, 0L, 0L);
      GH_HIP(hipGetLastError());
    }
    return GH_OK;
  };
  h->Rtot = 0; h->maxR = 0; h->max_chunks = 0;
  for (int l = 0; l < nlev; ++l) {
    HLevel* L = h->levels[l];
    const int nn = (int)L->node_ids.size();
    std::vector<LvlNode> ln(nn);
    const int seed_off = (l >= l0 && l - l0 < (int)h->sub.seed_off.size()) ? h->sub.seed_off[l - l0] : 0;
    for (int q = 0; q < nn; ++q) { const HNode& nd = h->nodes[L->node_ids[q]]; ln[q] = {nd.start, nd.half, nd.size, seed_off}; }
    if (!L->nodes_up) { GH_CHECK(upload(L->d_nodes, ln, st)); L->nodes_up = true; }
    GH_CHECK(L->d_ranks.ensure(nn * sizeof(int)));
    if (L->top) {                                        // no ACA here: its "rank" is the level's, the staged factors are zero-padded to it
      GH_HIP(hipMemcpyAsync(L->d_ranks.p, &h->sub.R[l], sizeof(int), hipMemcpyHostToDevice, st));
      al[l].G = 1;
      continue;
    }
    // cluster size: as many workgroups per node as keep the whole grid resident (nodes * G <= 256)
    // and leave every thread `ept` columns (measured at C4 with a switch that went in round 4: 2 -> 12.5 ms,
    // 4 -> 13.1, 8 -> 14.1, 16 -> 16.0: the step is bound by per-thread memory latency, not by the
    // barriers, so more and smaller workgroups win)
    int G = 1;
    {
      const int ept = 2;
      int min_half = INT32_MAX;
      for (int q = 0; q < nn; ++q) min_half = std::min(min_half, ln[q].half);
      while (G * 2 * nn <= 256 && (long)(G * 2) * ACA_THREADS * ept <= min_half) G *= 2;
    }
    al[l].G = G;
  }
  if (concurrent && h->st_b) {
    {
      // (a memset per level in front of every launch: thirteen ~7 us fill kernels, 100 us before the one-workgroup levels started)
      size_t off = 0;
      std::vector<size_t> at(nlev, 0);
      for (int l = l0; l < nlev; ++l) {
        if (h->levels[l]->top) continue;
        at[l] = off;
        off += (size_t)gh_round_up((int64_t)(h->levels[l]->node_ids.size() * (sizeof(unsigned) + 2 * sizeof(int)) + 2 * sizeof(int)), 256);   // bars | sel | fail, trunc | dur
      }
      GH_CHECK(sync_all.ensure(std::max<size_t>(off, 256)));
      GH_HIP(hipMemsetAsync(sync_all.p, 0, std::max<size_t>(off, 256), st));
      for (int l = l0; l < nlev; ++l) if (!h->levels[l]->top) { al[l].syncp = (char*)sync_all.p + at[l]; al[l].sync_cleared = true; }
    }
    GH_HIP(hipEventRecord(h->ev_b, st));                   // x and the node tables are uploaded
    GH_HIP(hipStreamWaitEvent(h->st_b, h->ev_b, 0));
    // Clustered levels: one launch, the 256 workgroup slots dealt so that the per-thread load is as even
    // as it gets (start from <= 32 workgroups per level, then keep doubling the cluster of the level
    // whose threads carry most columns).
    std::vector<int> cl;
    for (int l = l0; l < nlev; ++l) if (al[l].G > 1) cl.push_back(l);
    if (cl.size() >= 2) {
      std::vector<int> gmax(nlev, 1), half(nlev, 1);
      int total = 0;
      for (int l : cl) {
        const int nn = (int)h->levels[l]->node_ids.size();
        gmax[l] = al[l].G;
        int mh = INT32_MAX;
        for (int id : h->levels[l]->node_ids) mh = std::min(mh, h->nodes[id].half);
        half[l] = mh;
        int G = 1;
        while (G * 2 <= gmax[l] && nn * G * 2 <= g_hodlr_coop_wgs / 8) G *= 2;
        al[l].G = G;
        // (round 6: a clusterable level left with one workgroup per node -- level 5 of C4: 32 blocks of 4096 x 4096, 0.54 ms per
        //  node -- stays in the cooperative launch as one-workgroup segments at its end while the launch still fits the chip: in
        //  the one-workgroup launch its nodes found no SIMD with room beside a cooperative workgroup before ~0.5 ms and were the
        //  tail of phase 1, profiles/r06/hodlr_phase1_registers.md)
        if (G > 1 || (g_hodlr_coop_singles && nn <= g_hodlr_coop_wgs / 8)) total += nn * G;       // (reserved: the doubling below stays inside the budget)
      }
      for (;;) {
        int best = -1;
        double load = 0.0;
        for (int l : cl) {
          const int nn = (int)h->levels[l]->node_ids.size();
          if (al[l].G < 2 || al[l].G * 2 > gmax[l] || total + nn * al[l].G > g_hodlr_coop_wgs) continue;
          const double ld = (double)half[l] / al[l].G;
          if (ld > load) { load = ld; best = l; }
        }
        if (best < 0) break;
        total += (int)h->levels[best]->node_ids.size() * al[best].G;
        al[best].G *= 2;
      }
      // (the clusters below the root at 1 / g_hodlr_coop_lower of that width: see there)
      for (size_t q = 1; q < cl.size(); ++q) { int& G = al[cl[q]].G; int d = g_hodlr_coop_lower; while (d > 1 && G >= 4) { G /= 2; d /= 2; } }
      std::vector<int> fused, single;
      {
        int used = 0;
        for (int l : cl) if (al[l].G > 1) used += (int)h->levels[l]->node_ids.size() * al[l].G;
        for (int l : cl) {
          const int nn = (int)h->levels[l]->node_ids.size();
          if (al[l].G > 1) fused.push_back(l);
          // (G = 1 segments.  With the clusters below the root at half width there is room for a second such level -- C4: level 6,
          //  64 blocks of 2048 x 2048, which start at 0 instead of waiting ~0.24 ms for a CU: 3.40 -> 3.31 ms)
          else if (g_hodlr_coop_singles && nn <= g_hodlr_coop_wgs / 4 && used + nn <= g_hodlr_coop_wgs) { fused.push_back(l); used += nn; }
          else single.push_back(l);
        }
      }
      if (h->aca_fused_ev[0] == nullptr) { GH_HIP(hipEventCreate(&h->aca_fused_ev[0])); GH_HIP(hipEventCreate(&h->aca_fused_ev[1])); }
      GH_HIP(hipEventRecord(h->aca_fused_ev[0], st));
      GH_CHECK(enqueue_fused(fused, rcap0, st, h->d_aca_segs));
      GH_HIP(hipEventRecord(h->aca_fused_ev[1], st));
      h->aca_timed = true;
      // The one-workgroup-per-node levels and the leaf stage are independent of the fused launch and of
      // each other, but HIP multiplexes streams onto 4 hardware queues (3 seen by this library: more
      // streams than that just share a queue and serialise -- measured: a "fourth stream" ran its kernels
      // behind the fused launch).  So: three queues -- the solver's stream (fused launch first) and two
      // side streams -- and the items are dealt longest-first onto the least loaded queue, with the
      // durations MEASURED in the previous compute() of this handle (HIP events; a default guess the
      // first time): ranks, and with them the cost profile, hardly move inside an optimiser loop.
      if (!h->st_c) {
        hipStream_t shq[4] = {nullptr, nullptr, nullptr, nullptr};
        if (h->shared_streams && gh_shared_streams(h->opts.device, shq)) h->st_c = shq[3];
        if ((!h->shared_streams && hipStreamCreateWithFlags(&h->st_c, hipStreamNonBlocking) != hipSuccess) ||
            hipEventCreateWithFlags(&h->ev_c, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); h->st_c = nullptr; }
      }
      if (!h->st_d && h->shared_streams && h->st_c) {
        hipStream_t shq[4] = {nullptr, nullptr, nullptr, nullptr};
        if (gh_shared_streams(h->opts.device, shq) && shq[1] && hipEventCreateWithFlags(&h->ev_d, hipEventDisableTiming) == hipSuccess) h->st_d = shq[1];
        else (void)hipGetLastError();
      }
      std::vector<int> ones = single;
      for (int l = l0; l < nlev; ++l) if (gmax[l] == 1) ones.push_back(l);
      // The one-workgroup-per-node levels as ONE launch too (segments in order of decreasing block size: the long
      // workgroups are dispatched first): launched one per queue they were balanced by hand over three queues with last
      // compute()'s durations, and the queue that drew the two slowest levels ended 0.4 ms after the others.  One grid
      // leaves the balancing to the dispatcher: C4 5.36 -> 5.24 ms.  (EVERY level in one grid, clustered segments first, was
      // no better: 5.30.)
      // the deep levels that take the wavefront-per-node kernel: launches of their own, first, on the fourth queue (the
      // process-wide chain stream: high priority) where there is one
      bool wave_on_d = false;
      {
        std::vector<int> keep;
        hipStream_t sw = h->st_d ? h->st_d : h->st_b;
        for (int l : ones) {
          if (!wave_mr(l)) { keep.push_back(l); continue; }
          if (sw == h->st_d && !wave_on_d) { GH_HIP(hipStreamWaitEvent(h->st_d, h->ev_b, 0)); wave_on_d = true; }
          GH_CHECK(enqueue_level(l, rcap0, sw));
        }
        ones.swap(keep);
        if (wave_on_d) { GH_HIP(hipEventRecord(h->ev_d, h->st_d)); GH_HIP(hipStreamWaitEvent(st, h->ev_d, 0)); }
      }
      if (ones.size() >= 2 && h->st_c) {
        std::sort(ones.begin(), ones.end());
#ifdef ACA_ONES_DEEPEST_FIRST
        std::reverse(ones.begin(), ones.end());
#endif
#ifdef ACA_ONES_LAST_FIRST
        std::rotate(ones.begin(), ones.end() - 1, ones.end());       // deepest level first, then by decreasing block size
#endif
        GH_HIP(hipStreamWaitEvent(h->st_c, h->ev_b, 0));
        if (h->st_d) GH_HIP(hipStreamWaitEvent(h->st_d, h->ev_b, 0));
        // (round 5, measured and left out -- HISTORY.md: the leaf chain first and this launch behind it: +1.5 %; the levels with
        //  blocks of <= 256 rows as a launch of their own with 256 / 128 / 64 threads per workgroup: 0 / +2.4 / +8 %.  The three
        //  pieces of this phase are bound by what they ask of the chip together, not by their order or their shapes.)
        GH_CHECK(enqueue_fused(ones, rcap0, h->st_b, h->d_aca_segs1));
        GH_CHECK(leaf_stage(h->st_c));
        GH_HIP(hipEventRecord(h->ev_c, h->st_c));
        GH_HIP(hipStreamWaitEvent(st, h->ev_c, 0));
        h->aca_timed = false;
        ones.clear();
      }
      struct Item { int level; double cost; };             // level -1: the leaf stage
      std::vector<Item> items;
      const bool have = (int)h->aca_ms.size() == nlev + 2;     // [0..nlev): levels, [nlev]: fused launch, [nlev+1]: leaf stage
      for (int l : ones) items.push_back({l, have && h->aca_ms[l] > 0 ? h->aca_ms[l] : 1.0});
      if (!leaves_done) items.push_back({-1, have && h->aca_ms[nlev + 1] > 0 ? h->aca_ms[nlev + 1] : 1.2});
      std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.cost > y.cost; });
      hipStream_t qs[4] = {st, h->st_b, h->st_c ? h->st_c : h->st_b, h->st_d ? h->st_d : h->st_b};
      double load[4] = {have && h->aca_ms[nlev] > 0 ? h->aca_ms[nlev] : 1.5, 0.0, h->st_c ? 0.0 : 1e30, h->st_d ? 0.0 : 1e30};
      if (h->st_c) GH_HIP(hipStreamWaitEvent(h->st_c, h->ev_b, 0));
      if (h->st_d) GH_HIP(hipStreamWaitEvent(h->st_d, h->ev_b, 0));
      h->aca_ev_used = 0;
      auto stamp = [&](hipStream_t sx) -> int {               // record the next timing event on sx
        if (h->aca_ev_used == h->aca_ev.size()) {
          hipEvent_t e;
          GH_HIP(hipEventCreate(&e));
          h->aca_ev.push_back(e);
        }
        GH_HIP(hipEventRecord(h->aca_ev[h->aca_ev_used++], sx));
        return GH_OK;
      };
      h->aca_items.clear();
      // (the fused launch was enqueued above, between two stamps on st)
      for (const Item& it : items) {
        int q = 0;
        for (int w = 1; w < 4; ++w) if (load[w] < load[q]) q = w;
        load[q] += it.cost;
        GH_CHECK(stamp(qs[q]));
        if (it.level >= 0) GH_CHECK(enqueue_level(it.level, rcap0, qs[q])); else GH_CHECK(leaf_stage(qs[q]));
        GH_CHECK(stamp(qs[q]));
        h->aca_items.push_back(it.level);
      }
      if (h->st_c) { GH_HIP(hipEventRecord(h->ev_c, h->st_c)); GH_HIP(hipStreamWaitEvent(st, h->ev_c, 0)); }
      if (h->st_d) { GH_HIP(hipEventRecord(h->ev_d, h->st_d)); GH_HIP(hipStreamWaitEvent(st, h->ev_d, 0)); }
    } else {
      for (int l = l0; l < nlev; ++l) GH_CHECK(enqueue_level(l, rcap0, al[l].G > 1 ? st : h->st_b));
      GH_CHECK(leaf_stage(h->st_b));
    }
    GH_HIP(hipEventRecord(h->ev_b, h->st_b));
    GH_HIP(hipStreamWaitEvent(st, h->ev_b, 0));
    {
      // one gather launch + one copy into pinned memory for the ranks and the two failure flags of every level
      std::vector<GatherItem> items;
      int tot = 0;
      for (int l = l0; l < nlev; ++l) {
        HLevel* L = h->levels[l];
        const int nn = (int)L->node_ids.size();
        items.push_back({(const int*)L->d_ranks.p, nn, tot}); tot += nn;
        items.push_back({(const int*)((unsigned*)al[l].syncp + nn) + nn, 2 + (al[l].timed ? nn : 0), tot}); tot += 2 + (al[l].timed ? nn : 0);
      }
      // (pinned host memory, read by the kernel in place: the item table needs no copy of its own)
      const size_t need = (size_t)tot + 4 + items.size() * (sizeof(GatherItem) / sizeof(int));
      if (need > h->h_gather_cap) {
        if (h->h_gather) (void)hipHostFree(h->h_gather);
        h->h_gather = nullptr; h->h_gather_cap = 0;
        GH_HIP(hipHostMalloc((void**)&h->h_gather, need * 2 * sizeof(int), hipHostMallocDefault));
        h->h_gather_cap = need * 2;
      }
      GatherItem* const h_items = (GatherItem*)(h->h_gather + ((tot + 3) / 4) * 4);       // (16-byte aligned, behind the results)
      memcpy(h_items, items.data(), items.size() * sizeof(GatherItem));
      GH_CHECK(h->d_gather.ensure((size_t)tot * sizeof(int)));
      hipLaunchKernelGGL(hodlr_gather_kernel, dim3((unsigned)items.size()), dim3(256), 0, st, (const GatherItem*)h_items, (int*)h->d_gather.p);
      GH_HIP(hipGetLastError());
      GH_HIP(hipMemcpyAsync(h->h_gather, h->d_gather.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, st));
      GH_HIP(hipStreamSynchronize(st));
      int at = 0;
      for (int l = l0; l < nlev; ++l) {
        HLevel* L = h->levels[l];
        const int nn = (int)L->node_ids.size();
        L->ranks.assign(h->h_gather + at, h->h_gather + at + nn); at += nn;
        al[l].flags[0] = h->h_gather[at]; al[l].flags[1] = h->h_gather[at + 1]; at += 2;
        if (al[l].timed) { L->aca_dur.assign(h->h_gather + at, h->h_gather + at + nn); at += nn; }
      }
    }
    if (h->aca_timed) {                                 // durations for the next compute()'s schedule
      h->aca_ms.assign(nlev + 2, 0.0);
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->aca_fused_ev[0], h->aca_fused_ev[1]) == hipSuccess) h->aca_ms[nlev] = ms;
      for (size_t q = 0; q < h->aca_items.size() && 2 * q + 1 < h->aca_ev_used; ++q) {
        if (hipEventElapsedTime(&ms, h->aca_ev[2 * q], h->aca_ev[2 * q + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        const int lv = h->aca_items[q];
        h->aca_ms[lv >= 0 ? lv : nlev + 1] = ms;
      }
      h->aca_timed = false;
    }
#ifdef GH_ACA_TIMES
    {
      static int calls = 0;
      if (++calls == 3)
        for (int l = l0; l < nlev; ++l) {
          if (al[l].G != 1) { fprintf(stderr, "[aca] level %d: G = %d\n", l, al[l].G); continue; }
          const int nn = (int)h->levels[l]->node_ids.size();
          std::vector<double> hp((size_t)nn * pstride);
          (void)hipMemcpy(hp.data(), al[l].part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost);
          double mx = 0, sum = 0, t0min = 1e300, t1max = 0, ps = 0, rem = 0;
          int who = 0;
          for (int q = 0; q < nn; ++q) {
            const double* e = hp.data() + (size_t)q * pstride;
            sum += e[0]; ps += e[3]; rem += e[2];
            if (e[0] > mx) { mx = e[0]; who = q; }
            t0min = std::min(t0min, e[4]); t1max = std::max(t1max, e[4] + e[0]);
          }
          const double* w = hp.data() + (size_t)who * pstride;
          fprintf(stderr, "[aca] level %2d: %4d nodes  per-node us mean %8.1f max %8.1f (node %d: rank %d, rows left %d, passes %d)  mean passes %.1f "
                  "mean rows left %.1f  first start %.1f .. last end %.1f us (100 MHz clock)\n", l, nn, sum / nn * 0.01, mx * 0.01, who, (int)w[1], (int)w[2], (int)w[3],
                  ps / nn, rem / nn, t0min * 0.01, t1max * 0.01);
        }
    }
#endif
    for (int l = 0; l < nlev; ++l) { if (l >= l0) GH_CHECK(settle_level(l)); rank_of_level(l); }
  } else {
    for (int l = 0; l < l0; ++l) rank_of_level(l);
    mark("serial ACA starts");
    for (int l = l0; l < nlev; ++l) {
      GH_CHECK(enqueue_level(l, rcap0, st));
      mark("  level enqueued");
      GH_CHECK(fetch_level(l, st));
      GH_HIP(hipStreamSynchronize(st));
      mark("  level synchronised");
      GH_CHECK(settle_level(l));
      GH_CHECK(compact_level(l));
    }
  }
  GhPooledBuf& Tcm = shared_Tcm;
  GhPooledBuf idx;
  Tcm.release();
  idx.release();
  mark("ACA done, ranks known");

  // ---- UA / VA (n x Rtot) and the per-level chunk / job tables
  const long Rtot = std::max(h->Rtot, 1);
  GH_CHECK(h->UA.ensure((size_t)n * Rtot * sizeof(double)));
  GH_CHECK(h->VA.ensure((size_t)n * Rtot * sizeof(double)));
  // UA / VA are written in full by the compaction when every level's nodes cover all n rows (a complete tree: level l
  // has 2^l internal nodes -- the case of C4); only then can the two memsets (157 MB each at C4) be skipped
  bool complete = true;
  for (int l = l0; l < nlev; ++l) if (h->levels[l]->node_ids.size() != ((size_t)1 << (l - l0))) complete = false;   // (a pseudo-level covers every local row)
  bool fused_compact = nlev <= 24;                     // (CompactSegs)
  for (int l = 0; l < nlev; ++l) if (levelB[l]) fused_compact = false;
  // (round 6) the compaction writes the level-major copy only and the leaf product -- the first thing that touches U -- reads that and
  // writes the row-major U itself, where that product is ONE pass of the 128-row-leaf kernel over all columns (LeafSrc)
  const bool u_from_v = g_hodlr_u_from_v && fused_compact && complete && l0 == 0 && nlev <= 24 && h->leaf_pitch == 128 && h->max_leaf <= 128 &&
                        h->Rtot > MV_C && h->Rtot <= 128;
  if (!(complete && fused_compact)) {
    GH_HIP(hipMemsetAsync(h->UA.p, 0, (size_t)n * Rtot * sizeof(double), st));
    GH_HIP(hipMemsetAsync(h->VA.p, 0, (size_t)n * Rtot * sizeof(double), st));
  }
  if (fused_compact) {
    std::vector<CompactSeg> segs;
    int b0 = 0;
    for (int l = 0; l < nlev; ++l) {
      HLevel* L = h->levels[l];
      if (L->R == 0) continue;
      const int nn = (int)L->node_ids.size(), ny = std::max(8, std::min(512, 2048 / nn));
      segs.push_back({L->top ? h->sub.T[l] : al[l].Tcm.d(), (const LvlNode*)L->d_nodes.p, (const int*)L->d_ranks.p, L->R, ny, (long)L->off, (long)n * L->off,
                      (long)L->R, b0, nn * ny});
      b0 += nn * ny;
    }
    if (b0 > 0) {
      CompactSegs cs;
      cs.n = (int)segs.size();
      for (int q = 0; q < cs.n; ++q) cs.s[q] = segs[q];
      hipLaunchKernelGGL(hodlr_compact_all_kernel, dim3((unsigned)b0), dim3(256), 0, st, cs,
                         (long)n, u_from_v ? (double*)nullptr : h->UA.d(), (long)Rtot, h->VA.d());
      GH_HIP(hipGetLastError());
    }
  }
  for (int l = 0; l < nlev; ++l) {
    HLevel* L = h->levels[l];
    if (L->R == 0) continue;
    const int R = L->R, nn = (int)L->node_ids.size();
    if (fused_compact) {
      // (done above, all levels in one launch)
    } else if (levelB[l]) {
      // (serial mode) scatter the compact level buffer into column block [off, off+R) of UA and VA
      GH_HIP(hipMemcpy2DAsync(h->UA.d() + L->off, Rtot * sizeof(double), levelB[l]->p, R * sizeof(double), R * sizeof(double), n, hipMemcpyDeviceToDevice, st));
      GH_HIP(hipMemcpyAsync(h->VA.d() + (long)n * L->off, levelB[l]->p, (size_t)n * R * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
      // every rank is known by now: the level's scratch goes straight into its column block (22 strided
      // device copies per compute() before)
      hipLaunchKernelGGL(hodlr_compact_kernel, dim3(nn, std::max(8, std::min(512, 2048 / nn))), dim3(256), 0, st, L->top ? h->sub.T[l] : al[l].Tcm.d(), (long)n, (const LvlNode*)L->d_nodes.p,
                         (const int*)L->d_ranks.p, R, h->UA.d(), (long)Rtot, (long)L->off, h->VA.d(), (long)R, (long)n * L->off);
      GH_HIP(hipGetLastError());
    }
    GH_CHECK(L->sinv.ensure((size_t)nn * 4 * R * R * sizeof(double)));
    if (L->tab_R == R && L->tab_off == L->off && L->tab_Rtot == Rtot) { h->max_chunks = std::max(h->max_chunks, L->nchunks); continue; }
    std::vector<Chunk> chunks;
    std::vector<int> crange(nn * 4);
    std::vector<MMJob> red, upd, updl, smul(nn);
    for (int q = 0; q < nn; ++q) {
      const HNode& nd = h->nodes[L->node_ids[q]];
      for (int half = 0; half < 2; ++half) {
        const int r0 = half == 0 ? nd.start : nd.start + nd.half;
        const int cnt = half == 0 ? nd.half : nd.size - nd.half;
        crange[(q * 2 + half) * 2] = (int)chunks.size();
        for (int s = 0; s < cnt; s += HCH) {
          const int nr = std::min(HCH, cnt - s);
          const int ch = (int)chunks.size();
          chunks.push_back({q, half, r0 + s, nr});
          red.push_back({(long)(r0 + s) * R, r0 + s, ch * R, R, nr});                    // A = V_l rows (level-major block, transposed access)
          upd.push_back({(long)(r0 + s) * Rtot, q * 2 * R + (half == 0 ? 0 : R), r0 + s, nr, R});
          updl.push_back({(long)(r0 + s) * R, q * 2 * R + (half == 0 ? 0 : R), r0 + s, nr, R});      // same against UL
        }
        crange[(q * 2 + half) * 2 + 1] = (int)chunks.size();
      }
      smul[q] = {(long)q * 4 * R * R, q * 2 * R, q * 2 * R, 2 * R, 2 * R};
    }
    L->nchunks = (int)chunks.size();
    L->chunk_geom.clear();
    for (const Chunk& c : chunks) { L->chunk_geom.push_back(c.row0); L->chunk_geom.push_back(c.nrows); }
    h->max_chunks = std::max(h->max_chunks, L->nchunks);
    GH_CHECK(upload(L->d_chunks, chunks, st));
    GH_CHECK(upload(L->d_crange, crange, st));
    GH_CHECK(upload(L->d_red_jobs, red, st));
    GH_CHECK(upload(L->d_upd_jobs, upd, st));
    GH_CHECK(upload(L->d_updl_jobs, updl, st));
    GH_CHECK(upload(L->d_smul_jobs, smul, st));
    L->tab_R = R; L->tab_off = L->off; L->tab_Rtot = Rtot;
  }
  mark("tables enqueued");
  // The ACA scratch (n x 256 doubles per level: 6 GB at C4, 12 GB for a 524288-row sub-tree) goes back to the block
  // cache NOW, not when compute() returns, in a SPLIT tree: the next sub-tree of this device starts while this one waits
  // for the others in its top levels, and found the cache empty -- tens of GB of hipMalloc / hipFree per compute(),
  // stalls of 1.4-2.8 s at N = 2M over four sub-trees on one GPU.  That takes a host synchronisation (another handle may pick the
  // blocks up on a stream of its own), 20 us in the middle of a 3.4-ms step: a handle that owns its whole tree keeps the scratch
  // until the end of compute() instead -- the guard below synchronises before the buffers' destructors run on any path out.
  struct SyncBeforeRelease { hipStream_t st; ~SyncBeforeRelease() { (void)hipStreamSynchronize(st); } } sync_before_release{st};
  if (l0 > 0) {
    GH_HIP(hipStreamSynchronize(st));
    mark("U, V assembled");
    for (auto& a : al) { a.Tcm.release(); a.idx.release(); a.sync.release(); a.part.release(); }
    sync_all.release();
    for (auto*& b : levelB) { delete b; b = nullptr; }
  }
  {
    size_t maxnodes = 1;
    for (auto* L : h->levels) maxnodes = std::max(maxnodes, L->node_ids.size() * (size_t)std::max(L->R, 1));
    h->cpass = std::max(CPASS, (h->maxR + 63) / 64 * 64);       // the core build handles a level's R columns in ONE pass
    GH_CHECK(h->P.ensure((size_t)std::max(h->max_chunks, 1) * std::max(h->maxR, 1) * h->cpass * sizeof(double)));
    GH_CHECK(h->Tsum.ensure(maxnodes * 2 * h->cpass * sizeof(double)));
    GH_CHECK(h->Tout.ensure(maxnodes * 2 * h->cpass * sizeof(double)));
    GH_CHECK(h->Y.ensure((size_t)n * h->cpass * sizeof(double)));
  }

  // ---- leaves (enqueued above, beside the ACA, when the levels run concurrently)
  if (!leaves_done) GH_CHECK(leaf_stage(st));
  mark("work arrays, leaf stage enqueued");

  // ---- factorisation sweep (hodlr.h:75-103, level-batched): leaves into every U, then levels bottom-up
  bool red_ready = false;                            // the chunk products of the next level to be processed are in P already
  if (h->Rtot > 0) {
    const HLevel* deepest = nullptr;
    for (int q = nlev - 1; q >= 0 && !deepest; --q) if (h->levels[q]->R > 0) deepest = h->levels[q];
    LeafSrc ls;
    if (u_from_v) {
      ls.VA = h->VA.d(); ls.nlev = nlev;
      for (int q = 0; q < nlev; ++q) { ls.off[q] = h->levels[q]->off; ls.R[q] = h->levels[q]->R; ls.offv[q] = (long)n * h->levels[q]->off; }
    }
    GH_CHECK(apply_leaves(h, h->UA.d(), Rtot, 0, h->Rtot, deepest, &red_ready, u_from_v ? &ls : nullptr));
  }
  std::vector<size_t> top_ld(l0, (size_t)-1);        // where in ld_all the core of pseudo-level l put its log|det|
  bool local_done = (l0 == 0);
  for (int l = nlev - 1; l >= 0; --l) {
    HLevel* L = h->levels[l];
    if (l < l0 && !local_done) {
      // everything below needs the other devices: tell the owner of the split that this one has finished on its own
      mark("local sweep enqueued");
      GH_HIP(hipStreamSynchronize(st));
      mark("local sweep done");
      local_done = true;
      if (h->sub.local_done) GH_CHECK(h->sub.local_done(h->sub.ctx));
    }
    if (L->R == 0) continue;
    const int R = L->R, nn = (int)L->node_ids.size();
    if (L->top) {
      // The ancestor's core: V^T U over its own columns, each device the rows it holds, completed over the devices below
      // the ancestor; every one of them then inverts the same 2R x 2R matrix and updates its own rows of the shallower U's.
      GH_CHECK(launch_red(h, (const MMJob*)L->d_red_jobs.p, L->nchunks, R, h->VA.d() + (long)n * L->off,
                          h->UA.d(), Rtot, L->off, h->P.d(), h->cpass, 0, R));
      hipLaunchKernelGGL(hodlr_sum_kernel, dim3(nn, 2 * R), dim3(64 * SUM_NS), 0, st, h->P.d(), (const int*)L->d_crange.p, R, (long)h->cpass, R, h->Tsum.d());
      GH_HIP(hipGetLastError());
      GH_CHECK(h->sub.allreduce(h->sub.ctx, l, h->Tsum.d(), 2 * R, R, (long)h->cpass, st));
      const std::vector<long> offs1(1, 0L);
      const std::vector<int> sizes1(1, 2 * R);
      GhBuf* const tabs[3] = {&L->d_gj_offs, &L->d_gj_sizes, &L->d_gj_sc};
      GH_CHECK(batched_inverse(h, L->sinv.d(), offs1, sizes1, h->ld_all.d() + ld_at, tabs, L->gj_R == R, h->Tsum.d(), R));
      L->gj_R = R;
      top_ld[l] = ld_at;
      ld_at += 1;
      GH_CHECK(apply_level(h, L, h->UA.d(), Rtot, 0, L->off, h->UA.d(), Rtot));
      continue;
    }
    // S = I + [0, V1^T U1; V0^T U0, 0] with the CURRENT U of this level.  The products V_l^T U that the
    // core needs (columns [off, off + R)) and the ones that applying this level's inverse to the shallower
    // levels' U needs (columns [0, off)) read the same V_l chunks and neighbouring columns of the same U
    // rows: ONE reduce + sum over columns [0, off + R) serves both (two launches fewer per level).
    const int Call = L->off + R;
    const bool merged = Call <= h->cpass;
    // (round 6) levels of many small nodes: sum + core inverse + core product in one launch (hodlr_core_kernel)
    const bool core_fused = g_hodlr_core_fused && merged && L->off > 0 && nn >= 32 && 2 * R <= 32 && Call <= 128 &&
                            L->nchunks <= 128 * nn;      // (a workgroup adds up ITS node's chunk partials: few chunks per node)
    if (core_fused) {
      if (!red_ready)
        GH_CHECK(launch_red(h, (const MMJob*)L->d_red_jobs.p, L->nchunks, R, h->VA.d() + (long)n * L->off,
                            h->UA.d(), Rtot, 0, h->P.d(), h->cpass, 0, Call));
      red_ready = false;
#define GH_CORE_LAUNCH(NM) hipLaunchKernelGGL(hodlr_core_kernel<NM>, dim3(nn), dim3(256), 0, st, h->P.d(), (const int*)L->d_crange.p, R, (long)h->cpass, Call, \
                                              L->off, L->off, L->sinv.d(), h->ld_all.d() + ld_at, (int*)h->flags.p, h->Tout.d())
      if (2 * R <= 8) GH_CORE_LAUNCH(8); else if (2 * R <= 16) GH_CORE_LAUNCH(16); else GH_CORE_LAUNCH(32);
#undef GH_CORE_LAUNCH
      GH_HIP(hipGetLastError());
      L->gj_R = -1;                                    // (the separate launch's tables were not refreshed)
      ld_at += nn;
      const HLevel* nx = nullptr;
      for (int q = l - 1; q >= 0 && !nx; --q) if (h->levels[q]->R > 0) nx = h->levels[q];
      if (updred_possible(L, nx, L->off, h->cpass)) {
        GH_CHECK(launch_updred(h, L, nx, h->UA.d() + L->off, Rtot, h->Tout.d(), h->cpass, h->UA.d(), Rtot, L->off,
                               h->VA.d() + (long)n * nx->off, h->P.d(), h->cpass));
        red_ready = true;
      } else {
        GH_CHECK(launch_upd(h, (const MMJob*)L->d_upd_jobs.p, L->nchunks, R, h->UA.d() + L->off, Rtot,
                            h->Tout.d(), h->cpass, h->UA.d(), Rtot, L->off));
      }
      continue;
    }
    if (merged) {
      // (red_ready: the deeper level's update pass has already formed this level's chunk products -- hodlr_updred_kernel)
      if (!red_ready)
        GH_CHECK(launch_red(h, (const MMJob*)L->d_red_jobs.p, L->nchunks, R, h->VA.d() + (long)n * L->off,
                            h->UA.d(), Rtot, 0, h->P.d(), h->cpass, 0, Call));
      red_ready = false;
      hipLaunchKernelGGL(hodlr_sum_kernel, dim3(nn, 2 * R), dim3(64 * SUM_NS), 0, st, h->P.d(), (const int*)L->d_crange.p, R, (long)h->cpass, Call, h->Tsum.d());
      GH_HIP(hipGetLastError());
    } else {
      GH_CHECK(launch_red(h, (const MMJob*)L->d_red_jobs.p, L->nchunks, R, h->VA.d() + (long)n * L->off,
                          h->UA.d(), Rtot, L->off, h->P.d(), h->cpass, 0, R));
      hipLaunchKernelGGL(hodlr_sum_kernel, dim3(nn, 2 * R), dim3(64 * SUM_NS), 0, st, h->P.d(), (const int*)L->d_crange.p, R, (long)h->cpass, R, h->Tsum.d());
      GH_HIP(hipGetLastError());
    }
    const double* const core_src = h->Tsum.d() + (merged ? L->off : 0);      // V_l^T U[:, own columns]: the core is built from it inside the inverse
    std::vector<long> offs(nn);
    std::vector<int> sizes(nn, 2 * R);
    for (int q = 0; q < nn; ++q) offs[q] = (long)q * 4 * R * R;
    {
      GhBuf* const tabs[3] = {&L->d_gj_offs, &L->d_gj_sizes, &L->d_gj_sc};
      GH_CHECK(batched_inverse(h, L->sinv.d(), offs, sizes, h->ld_all.d() + ld_at, tabs, L->gj_R == R, core_src, R));
      L->gj_R = R;
    }
    ld_at += nn;
    // apply this level's inverse to the U's of all shallower levels: columns [0, off)
    if (merged && L->off > 0) {
      // (Tsum already holds V_l^T U[:, 0:off]: core product and update only)
      GH_CHECK(launch_mm(h, (const MMJob*)L->d_smul_jobs.p, nn, 2 * R, L->sinv.d(), 2 * R, 1,
                         h->Tsum.d(), h->cpass, 0, h->Tout.d(), h->cpass, 0, L->off, false));
      const HLevel* nx = nullptr;
      for (int q = l - 1; q >= 0 && !nx; --q) if (h->levels[q]->R > 0) nx = h->levels[q];
      if (updred_possible(L, nx, L->off, h->cpass)) {
        GH_CHECK(launch_updred(h, L, nx, h->UA.d() + L->off, Rtot, h->Tout.d(), h->cpass, h->UA.d(), Rtot, L->off,
                               h->VA.d() + (long)n * nx->off, h->P.d(), h->cpass));
        red_ready = true;
      } else {
        GH_CHECK(launch_upd(h, (const MMJob*)L->d_upd_jobs.p, L->nchunks, R, h->UA.d() + L->off, Rtot,
                            h->Tout.d(), h->cpass, h->UA.d(), Rtot, L->off));
      }
    } else {
      GH_CHECK(apply_level(h, L, h->UA.d(), Rtot, 0, L->off, h->UA.d(), Rtot));
    }
  }
  // (the level-major copy of the final U that the WIDE solves multiply from is made when one of them asks for it -- ensure_ul;
  //  the narrow solve of a log-likelihood reads the row-major U: 90 us of every C4 step for a copy nothing read)
  h->ul_valid = false;
  // (into PINNED host memory: a copy to pageable memory is staged and waited for inside the call -- two of them were ~45 us
  //  between the last kernel and the return)
  const size_t nld = std::max<size_t>(n_blocks, 1);
  if (h->pin_doubles < nld + 2) {
    if (h->pin) (void)hipHostFree(h->pin);
    h->pin = nullptr; h->pin_doubles = 0;
    GH_HIP(hipHostMalloc((void**)&h->pin, (nld + 2 + 1024) * sizeof(double), hipHostMallocDefault));
    h->pin_doubles = nld + 2 + 1024;
  }
  double* const ld_host = h->pin;
  int fl[4] = {0, 0, 0, 0};
  GH_HIP(hipMemcpyAsync(ld_host, h->ld_all.p, nld * sizeof(double), hipMemcpyDeviceToHost, st));
  GH_HIP(hipMemcpyAsync(ld_host + nld, h->flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  GH_HIP(hipStreamSynchronize(st));
  memcpy(fl, ld_host + nld, 4 * sizeof(int));
  long long leaf_info = 0;
  memcpy(&leaf_info, fl + 2, sizeof(long long));
  if (leaf_info != 0) { gh_set_error("HODLR: a leaf block is not positive definite"); return GH_ERR_NOT_PD; }
  if (fl[0] != 0) { gh_set_error("HODLR: singular block encountered (matrix %d of its batch)", fl[0] - 1); return GH_ERR_NOT_PD; }
  // (a sub-tree handle reports the blocks it owns alone; the ancestors' cores, the same on every device below them,
  //  are handed to the owner of the split separately)
  h->sub.ld_top.assign(l0, 0.0);
  for (int l = 0; l < l0; ++l) if (top_ld[l] != (size_t)-1) { h->sub.ld_top[l] = ld_host[top_ld[l]]; ld_host[top_ld[l]] = 0.0; }
  double logdet = 0.0;
  for (size_t i = 0; i < ld_at; ++i) logdet += ld_host[i];          // leaves first, then the cores bottom-up: fixed order
  h->logdet = logdet;
  h->computed = true;
  if (logdet_out) *logdet_out = logdet;
  return GH_OK;
}

static int need(gh_hodlr* h) {
  if (!h) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (!h->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  GH_HIP(hipSetDevice(h->opts.device));
  return GH_OK;
}

extern "C" int gh_hodlr_solve(gh_hodlr* h, const double* b, int64_t nrhs, double* out) {
  GH_CHECK(need(h));
  if (!b || !out || nrhs <= 0) { gh_set_error("bad argument to solve"); return GH_ERR_BAD_ARG; }
  const size_t tot = (size_t)h->n * nrhs;
  GH_CHECK(h->rhs.ensure(tot * sizeof(double)));
  GH_CHECK(gh_to_device(h->rhs.d(), b, tot, h->st));
  GH_CHECK(solve_all(h, h->rhs.d(), nrhs, (int)nrhs));
  return gh_from_device(out, h->rhs.d(), tot, h->st);
}
extern "C" int gh_hodlr_dot_solve(gh_hodlr* h, const double* y, double* out) {
  GH_CHECK(need(h));
  if (!y || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  GH_CHECK(h->rhs.ensure((size_t)h->n * sizeof(double)));
  GH_CHECK(h->work.ensure((size_t)h->n * sizeof(double)));
  GH_CHECK(gh_to_device(h->rhs.d(), y, (size_t)h->n, h->st));
  const double* yd = y;                              // (a device-resident y is read where it is)
  if (!gh_is_device_ptr(y)) { GH_CHECK(gh_to_device(h->work.d(), y, (size_t)h->n, h->st)); yd = h->work.d(); }
  GH_CHECK(solve_all(h, h->rhs.d(), 1, 1));
  GH_CHECK(h->dotp.ensure(256 * sizeof(double)));
  hipLaunchKernelGGL(hodlr_dot_kernel, dim3(256), dim3(256), 0, h->st, yd, h->rhs.d(), (long)h->n, h->dotp.d());
  hipLaunchKernelGGL(hodlr_dot_kernel, dim3(1), dim3(256), 0, h->st, h->dotp.d(), (const double*)nullptr, 256L, h->scal.d());
  GH_HIP(hipGetLastError());
  double v = 0.0;
  GH_HIP(hipMemcpyAsync(&v, h->scal.d(), sizeof(double), hipMemcpyDeviceToHost, h->st));
  GH_HIP(hipStreamSynchronize(h->st));
  *out = v;
  return GH_OK;
}
extern "C" int gh_hodlr_get_inverse(gh_hodlr* h, double* out) {
  GH_CHECK(need(h));
  if (!out) { gh_set_error("null output"); return GH_ERR_BAD_ARG; }
  const long n = h->n;
  GH_CHECK(h->rhs.ensure((size_t)n * n * sizeof(double)));
  GH_HIP(hipMemsetAsync(h->rhs.p, 0, (size_t)n * n * sizeof(double), h->st));
  hipLaunchKernelGGL(hodlr_eye_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->rhs.d(), n);
  GH_HIP(hipGetLastError());
  GH_CHECK(solve_all(h, h->rhs.d(), n, (int)n));
  return gh_from_device(out, h->rhs.d(), (size_t)n * n, h->st);
}
extern "C" int gh_hodlr_ranks(const gh_hodlr* h, int32_t* ranks_out, int32_t max_out, int32_t* n_out) {
  if (!h || !n_out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  int cnt = 0;
  for (auto* L : h->levels)
    for (int r : L->ranks) { if (ranks_out && cnt < max_out) ranks_out[cnt] = r; ++cnt; }
  *n_out = cnt < max_out ? cnt : max_out;
  return GH_OK;
}

// ===================================================================== the tree split over several devices
// gh_hodlr_mgpu_* (include/george_amd.h): the top log2(P) levels of the tree are shared, sub-tree p is an ordinary
// gh_hodlr handle on devices[p] in sub-tree mode (HSub above).  One host thread per device; the threads meet at host
// barriers, and the only data they exchange after the ancestors' factors have been dealt out are the 2R x C sums of the
// top levels, through pinned host memory.
namespace {
std::mutex g_hm_dev_mu[16];          // one clustered (spin-waiting) ACA grid per PHYSICAL device at a time ("virtual devices")

struct gh_hodlr_mgpu_impl;
struct HmRank {
  int p = 0, dev = 0;
  gh_hodlr* h = nullptr;
  gh_kernel kern;
  GhBuf xg;                                   // all N points: only where the ACA of a top node runs
  std::vector<GhBuf*> stage;                  // [depth] local rows of the ancestors' factors, column-major n x R_l
  long row0 = 0, n = 0;
  double* pin = nullptr;                      // pinned: [0, cap) this device's partial sums, [cap, 2 cap) the completed sums
  size_t pin_cap = 0, pin_cnt = 0;
  std::unique_lock<std::mutex> dev_lock;
  int rc = GH_OK;
  std::string err;
  double ld = 0.0;
  gh_hodlr_mgpu_impl* owner = nullptr;
  std::vector<int> seed_off;
};
struct HmTop {                                // a node above the split
  int level = 0, q = 0, start = 0, half = 0, size = 0;
  int runner = 0;                             // the rank whose device runs its ACA
  int first = 0, span = 1;                    // the ranks below it: [first, first + span)
  int rank = 0;
  GhBuf Tcm, packed;                          // on the runner's device: ACA scratch; the same rows dealt into one contiguous chunk per rank
  std::vector<long> pack_off;
  HostBarrier bar;
};
struct gh_hodlr_mgpu_impl {
  gh_hodlr_mgpu_opts opts;
  int P = 1, depth = 0;
  std::vector<HmRank> ranks;
  std::vector<std::vector<HmTop*>> top;       // [level][q]
  HostBarrier world;
  std::atomic<int> abort{0};
  int64_t n = 0;
  int ndim = 0;
  bool computed = false;
  double logdet = 0.0;
  std::vector<int> all_ranks;
  int64_t top_n = -1;                         // the top nodes in `top` were laid out for this many points / this min_size
  int top_min = -1;
  void clear_top() { for (auto& lv : top) for (auto* t : lv) { if (t) { (void)hipSetDevice(ranks[t->runner].dev); delete t; } } top.clear(); }
};

// out[k * n_m + i] = Tcm[k * N + row0 + i]: the rows of one device out of a node's column-major factors
__global__ void hodlr_pack_rows_kernel(const double* Tcm, long N, long row0, long n_m, int r, double* out) {
  const long tot = n_m * r;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const long k = e / n_m, i = e % n_m;
    out[e] = Tcm[k * N + row0 + i];
  }
}

// hodlr.h:136-221 for ONE node above the split, on this handle's device against all N points (x_dev).  Same kernel,
// same cluster rule and same retry ladder as the levels of gh_hodlr_compute; nd.pad = the node's index in its level.
int aca_top_node(gh_hodlr* h, gh_kernel* k, const double* x_dev, long N, int ndim, int level, LvlNode nd, GhBuf& Tcm, int* rank_out) {
  hipStream_t st = h->st;
  GhPooledBuf d_node, d_rank, idx, sync, part;       // (all used on st only, and the call ends synchronised)
  GH_CHECK(d_node.ensure(sizeof(LvlNode)));
  GH_CHECK(d_rank.ensure(sizeof(int)));
  GH_HIP(hipMemcpyAsync(d_node.p, &nd, sizeof(LvlNode), hipMemcpyHostToDevice, st));
  int G = 1;
  {
    const int ept = 2;
    while (G * 2 <= 256 && (long)(G * 2) * ACA_THREADS * ept <= nd.half) G *= 2;
  }
  const int aca_fence = 0, aca_multi = 1;
  const int pstride = 8 + 2 * ACA_MAXR;
  const bool user_cap = h->opts.max_rank > 0;
  int rc = user_cap ? h->opts.max_rank : std::min(256, RANK_CAP);
  GH_CHECK(idx.ensure((size_t)N * sizeof(int)));
  GH_CHECK(sync.ensure(sizeof(unsigned) + sizeof(int) + 2 * sizeof(int)));
  GH_CHECK(part.ensure((size_t)G * pstride * sizeof(double)));
  GH_CHECK(aca_lds_attr());
  for (;;) {
    GH_CHECK(Tcm.ensure((size_t)N * rc * sizeof(double)));
    GH_HIP(hipMemsetAsync(sync.p, 0, sizeof(unsigned) + sizeof(int) + 2 * sizeof(int), st));
    unsigned* d_bars = (unsigned*)sync.p;
    int* d_sel = (int*)(d_bars + 1);
    int* d_fail = d_sel + 1;
#define GH_ACA_LAUNCH(F, CLU)                                                                                               \
    hipLaunchKernelGGL((hodlr_aca_kernel<F, CLU>), dim3(G), dim3(ACA_THREADS), G == 1 ? ACA_DYN_BYTES : 0, st, k->d_nodes, (int)k->nodes.size(), k->fast, \
                       ndim, x_dev, (const LvlNode*)d_node.p, Tcm.d(), N, rc, (int*)idx.p, (int*)d_rank.p, h->opts.tol,  \
                       (unsigned long long)(unsigned)h->opts.seed, level, G, d_bars, part.d(), pstride, d_sel, d_fail,   \
                       aca_multi, aca_fence, d_fail + 1, (const AcaSeg*)nullptr, 0, G == 1 ? ACA_CAPD : 0)
    if (G == 1) { if (k->fast.ok) GH_ACA_LAUNCH(true, false); else GH_ACA_LAUNCH(false, false); }
    else        { if (k->fast.ok) GH_ACA_LAUNCH(true, true); else GH_ACA_LAUNCH(false, true); }
#undef GH_ACA_LAUNCH
    GH_HIP(hipGetLastError());
    int flags[2] = {0, 0};
    GH_HIP(hipMemcpyAsync(flags, d_fail, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(rank_out, d_rank.p, sizeof(int), hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    if (flags[0]) { gh_set_error("HODLR: cluster barrier of the ACA kernel timed out at level %d", level); return GH_ERR_HIP; }
    if (!flags[1]) return GH_OK;
    if (user_cap || rc >= RANK_CAP) {
      gh_set_error("HODLR: an off-diagonal block of level %d needs a rank above %d to reach tol = %g (%s); the factorisation is not usable",
                   level, rc, h->opts.tol, user_cap ? "opts.max_rank" : "the solver's ceiling: loosen tol, raise min_size or use the dense solver");
      return user_cap ? GH_ERR_BAD_ARG : GH_ERR_RANK;
    }
    rc = std::min(2 * rc, RANK_CAP);
  }
}

// the 2R x C sums of a top level, completed over the ranks below the ancestor (HSub::allreduce)
int hm_allreduce(void* ctx, int level, double* dT, int rows, int cols, long pitch, hipStream_t st) {
  HmRank& r = *(HmRank*)ctx;
  gh_hodlr_mgpu_impl* H = r.owner;
  HmTop& t = *H->top[level][r.p >> (H->depth - level)];
  const size_t cnt = (size_t)rows * cols;
  if (cnt > r.pin_cap) {                      // (no other rank reads this one's buffer between two all-reduces; this rank's
    GH_HIP(hipStreamSynchronize(st));          //  own copy of the previous sums back to the device may still be in flight)
    if (r.pin) (void)hipHostFree(r.pin);
    r.pin = nullptr; r.pin_cap = 0;
    const size_t cap = std::max<size_t>(2 * cnt, 1 << 16);
    GH_HIP(hipHostMalloc((void**)&r.pin, 2 * cap * sizeof(double), hipHostMallocDefault));
    r.pin_cap = cap;
  }
  GH_HIP(hipMemcpy2DAsync(r.pin, cols * sizeof(double), dT, pitch * sizeof(double), cols * sizeof(double), rows, hipMemcpyDeviceToHost, st));
  GH_HIP(hipStreamSynchronize(st));
  r.pin_cnt = cnt;
  if (!t.bar.wait()) { gh_set_error("aborted: another device failed"); return GH_ERR_HIP; }
  double* sum = r.pin + r.pin_cap;
  for (int m = t.first; m < t.first + t.span; ++m) {            // fixed order: every rank adds up the same numbers the same way
    const HmRank& o = H->ranks[m];
    if (o.pin_cnt != cnt) { gh_set_error("HODLR split: ranks %d and %d disagree on the shape of a level-%d sum", r.p, m, level); H->abort.store(1); return GH_ERR_HIP; }
    if (m == t.first) memcpy(sum, o.pin, cnt * sizeof(double));
    else for (size_t e = 0; e < cnt; ++e) sum[e] += o.pin[e];
  }
  if (!t.bar.wait()) { gh_set_error("aborted: another device failed"); return GH_ERR_HIP; }
  GH_HIP(hipMemcpy2DAsync(dT, pitch * sizeof(double), sum, cols * sizeof(double), cols * sizeof(double), rows, hipMemcpyHostToDevice, st));
  return GH_OK;
}
int hm_local_done(void* ctx) {
  HmRank& r = *(HmRank*)ctx;
  if (r.dev_lock.owns_lock()) r.dev_lock.unlock();
  return GH_OK;
}

template <typename F>
int hm_run(gh_hodlr_mgpu_impl* H, F fn) {
  H->abort.store(0);
  H->world.reset();
  for (auto& lv : H->top) for (auto* t : lv) t->bar.reset();
  std::vector<std::thread> th;
  for (int i = 0; i < H->P; ++i) {
    th.emplace_back([H, i, &fn]() {
      HmRank& r = H->ranks[i];
      r.rc = GH_OK; r.err.clear();
      if (hipSetDevice(r.dev) != hipSuccess) { r.rc = GH_ERR_HIP; r.err = "hipSetDevice failed"; H->abort.store(1); return; }
      const int rc = fn(r);
      if (r.dev_lock.owns_lock()) r.dev_lock.unlock();
      if (rc != GH_OK) { r.rc = rc; r.err = gh_last_error(); H->abort.store(1); }
    });
  }
  for (auto& t : th) t.join();
  int first = GH_OK;
  for (auto& r : H->ranks) {
    if (r.rc == GH_OK) continue;
    if (first == GH_OK || r.err.find("aborted") == std::string::npos) {       // (prefer a real message to "saw the abort flag")
      first = r.rc;
      gh_set_error("sub-tree %d (device %d): %s", r.p, r.dev, r.err.c_str());
      if (r.err.find("aborted") == std::string::npos) break;
    }
  }
  return first;
}
}  // namespace

struct gh_hodlr_mgpu : gh_hodlr_mgpu_impl {};

extern "C" void gh_hodlr_mgpu_destroy(gh_hodlr_mgpu* H) {
  if (!H) return;
  H->clear_top();
  for (auto& r : H->ranks) {
    (void)hipSetDevice(r.dev);
    if (r.h) gh_hodlr_destroy(r.h);
    for (auto* b : r.stage) delete b;
    r.xg.release();
    if (r.kern.d_nodes) { (void)hipFree(r.kern.d_nodes); r.kern.d_nodes = nullptr; }
    if (r.pin) (void)hipHostFree(r.pin);
  }
  delete H;
}

extern "C" int gh_hodlr_mgpu_create(const gh_hodlr_mgpu_opts* opts, gh_hodlr_mgpu** out) {
  if (!opts || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  const int P = opts->n_dev;
  if (P < 1 || P > 16 || (P & (P - 1))) { gh_set_error("HODLR split: n_dev must be 1, 2, 4, 8 or 16 (got %d)", P); return GH_ERR_BAD_ARG; }
  const int ndev = gh_device_count();
  if (ndev <= 0) { gh_set_error("no HIP device available: the george_amd HODLR solver needs an MI355X"); return GH_ERR_HIP; }
  bool dup = false;
  for (int i = 0; i < P; ++i) {
    if (opts->devices[i] < 0 || opts->devices[i] >= ndev) { gh_set_error("HODLR split: device %d does not exist (%d visible)", opts->devices[i], ndev); return GH_ERR_BAD_ARG; }
    for (int j = 0; j < i; ++j) if (opts->devices[j] == opts->devices[i]) dup = true;
  }
  (void)dup;
  gh_hodlr_mgpu* H = new gh_hodlr_mgpu();
  H->opts = *opts;
  H->P = P;
  for (H->depth = 0; (1 << H->depth) < P; ++H->depth) {}
  H->ranks.resize(P);
  H->world.n = P;
  H->world.abort = &H->abort;
  for (int i = 0; i < P; ++i) {
    HmRank& r = H->ranks[i];
    r.p = i; r.dev = opts->devices[i]; r.owner = H;
    gh_hodlr_opts o;
    memset(&o, 0, sizeof(o));
    o.device = r.dev; o.min_size = opts->min_size; o.seed = opts->seed; o.max_rank = opts->max_rank; o.tol = opts->tol;
    const int rc = gh_hodlr_create(&o, &r.h);
    if (rc != GH_OK) { gh_hodlr_mgpu_destroy(H); return rc; }
    for (int l = 0; l < H->depth; ++l) r.stage.push_back(new GhBuf());
  }
  *out = H;
  return GH_OK;
}

// Host logic only (no device is touched): the rows of every sub-tree and, per level of the sub-trees, the index of each
// sub-tree's first internal node in the GLOBAL level (what keys a node's random stream) for a tree of n points split
// over n_dev devices.  seed_off: n_dev x max_levels, row-major, zero padded.  GH_ERR_BAD_ARG when a node above the split
// would be a leaf.  gh_hodlr_mgpu_compute lays its tree out through this function.
extern "C" int gh_hodlr_mgpu_layout(int64_t n, int32_t n_dev, int32_t min_size, int64_t* row0, int64_t* nrows,
                                    int32_t* seed_off, int32_t max_levels, int32_t* n_levels) {
  if (n <= 0 || n > 0x3fffffffL || n_dev < 1 || n_dev > 16 || (n_dev & (n_dev - 1)) || !row0 || !nrows) {
    gh_set_error("bad argument to layout"); return GH_ERR_BAD_ARG;
  }
  if (min_size < 1) min_size = 1;
  int depth = 0;
  while ((1 << depth) < n_dev) ++depth;
  struct Seg { int64_t start, size; };
  std::vector<Seg> cur(1, Seg{0, n});
  for (int l = 0; l < depth; ++l) {
    std::vector<Seg> next;
    for (const Seg& sg : cur) {
      const int64_t half = sg.size / 2;                      // hodlr.h:48: internal iff size / 2 >= min_size
      if (half < min_size) {
        gh_set_error("HODLR split: %lld points are too few for %d devices with min_size = %d (a node of level %d would be a leaf)",
                     (long long)n, n_dev, min_size, l);
        return GH_ERR_BAD_ARG;
      }
      next.push_back({sg.start, half});
      next.push_back({sg.start + half, sg.size - half});
    }
    cur.swap(next);
  }
  std::vector<std::vector<int>> cnt(n_dev);
  size_t maxl = 0;
  for (int p = 0; p < n_dev; ++p) {
    row0[p] = cur[p].start; nrows[p] = cur[p].size;
    std::vector<int64_t> sizes(1, cur[p].size);
    while (!sizes.empty()) {
      std::vector<int64_t> nx;
      int internal = 0;
      for (int64_t sz : sizes) if (sz / 2 >= min_size) { ++internal; nx.push_back(sz / 2); nx.push_back(sz - sz / 2); }
      if (internal == 0) break;
      cnt[p].push_back(internal);
      sizes.swap(nx);
    }
    maxl = std::max(maxl, cnt[p].size());
  }
  if (n_levels) *n_levels = (int32_t)maxl;
  if (seed_off) {
    for (int p = 0; p < n_dev; ++p)
      for (int l = 0; l < max_levels; ++l) {
        int v = 0;
        for (int o = 0; o < p; ++o) if ((size_t)l < cnt[o].size()) v += cnt[o][l];
        seed_off[(size_t)p * max_levels + l] = v;
      }
  }
  return GH_OK;
}

extern "C" int gh_hodlr_mgpu_compute(gh_hodlr_mgpu* H, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                                     const double* yerr, double* logdet_out) {
  if (!H || !k || !x || !yerr || n <= 0) { gh_set_error("bad argument to compute"); return GH_ERR_BAD_ARG; }
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  if (n > 0x3fffffffL) { gh_set_error("HODLR: n too large"); return GH_ERR_BAD_ARG; }
  if (gh_is_device_ptr(x) || gh_is_device_ptr(yerr)) { gh_set_error("HODLR split: x and yerr must be host pointers"); return GH_ERR_BAD_ARG; }
  H->computed = false;
  H->n = n; H->ndim = ndim;
  const int P = H->P, depth = H->depth, min_size = std::max(1, H->opts.min_size);
  // ---- the tree above the split (hodlr.h:47-64) and the rows of every sub-tree
  // (kept from one compute() to the next while n is the same: a top node holds 8 N rcap bytes of ACA scratch)
  const bool keep_top = H->top_n == n && H->top_min == min_size && (int)H->top.size() == depth;
  H->top_n = -1;
  if (!keep_top) { H->clear_top(); H->top.resize(depth); }
  struct Seg { int start, size; };
  std::vector<Seg> cur(1, Seg{0, (int)n});
  for (int l = 0; l < depth; ++l) {
    std::vector<Seg> next;
    for (int q = 0; q < (int)cur.size(); ++q) {
      const int half = cur[q].size / 2;
      if (half < min_size) {
        gh_set_error("HODLR split: %lld points are too few for %d devices with min_size = %d (a node of level %d would be a leaf)",
                     (long long)n, P, min_size, l);
        H->clear_top();
        return GH_ERR_BAD_ARG;
      }
      if (!keep_top) {
        HmTop* t = new HmTop();
        t->level = l; t->q = q; t->start = cur[q].start; t->half = half; t->size = cur[q].size;
        t->span = P >> l; t->first = q * t->span; t->runner = t->first + (l % t->span);
        t->bar.n = t->span; t->bar.abort = &H->abort;
        H->top[l].push_back(t);
      }
      next.push_back({cur[q].start, half});
      next.push_back({cur[q].start + half, cur[q].size - half});
    }
    cur.swap(next);
  }
  // rows of every sub-tree, and where its nodes sit in the global levels (a node's random stream is keyed by that)
  {
    int64_t r0[16], nr[16];
    int32_t nl = 0;
    GH_CHECK(gh_hodlr_mgpu_layout(n, P, min_size, r0, nr, nullptr, 0, &nl));
    std::vector<int32_t> so((size_t)P * std::max(nl, 1), 0);
    GH_CHECK(gh_hodlr_mgpu_layout(n, P, min_size, r0, nr, so.data(), nl, &nl));
    for (int p = 0; p < P; ++p) {
      H->ranks[p].row0 = (long)r0[p]; H->ranks[p].n = (long)nr[p];
      H->ranks[p].seed_off.assign(so.begin() + (size_t)p * nl, so.begin() + (size_t)(p + 1) * nl);
    }
  }
#ifdef GH_HODLR_PHASE_MARKS
  const bool dbg = true;
#else
  const bool dbg = false;
#endif
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
  const int rc = hm_run(H, [&](HmRank& r) -> int {
    gh_hodlr* h = r.h;
    hipStream_t st = h->st;
    double tm[6] = {0, 0, 0, 0, 0, 0};
    struct Report { bool on; int p; double* tm; ~Report() { if (on) fprintf(stderr, "[hodlr split] rank %d: top ACA done %.2f, met %.2f, rows pulled %.2f, met %.2f, lock %.2f, compute returned %.2f ms\n", p, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]); } } report{dbg, r.p, tm};
    // a private copy of the kernel program on this device (a gh_kernel caches ONE device copy)
    if (r.kern.d_nodes) { (void)hipFree(r.kern.d_nodes); r.kern.d_nodes = nullptr; }
    r.kern.nodes = k->nodes; r.kern.ndim = k->ndim; r.kern.size = k->size; r.kern.fast = k->fast; r.kern.device = -1;
    GH_CHECK(r.kern.upload());
    // ---- the ACA of the top nodes this device runs
    std::vector<HmTop*> mine;
    for (auto& lv : H->top) for (auto* t : lv) if (t->runner == r.p) mine.push_back(t);
    if (!mine.empty()) {
      GH_CHECK(r.xg.ensure((size_t)n * ndim * sizeof(double)));
      GH_CHECK(gh_to_device(r.xg.d(), x, (size_t)n * ndim, st));
      std::lock_guard<std::mutex> lk(g_hm_dev_mu[r.dev & 15]);
      for (HmTop* t : mine) {
        GH_CHECK(aca_top_node(h, &r.kern, r.xg.d(), (long)n, ndim, t->level, LvlNode{t->start, t->half, t->size, t->q}, t->Tcm, &t->rank));
        t->pack_off.assign(t->span, 0);
        GH_CHECK(t->packed.ensure(std::max<size_t>((size_t)t->size * t->rank, 1) * sizeof(double)));
        long off = 0;
        for (int m = 0; m < t->span && t->rank > 0; ++m) {
          const HmRank& o = H->ranks[t->first + m];
          t->pack_off[m] = off;
          const long tot = o.n * t->rank;
          hipLaunchKernelGGL(hodlr_pack_rows_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 4096)), dim3(256), 0, st,
                             t->Tcm.d(), (long)n, o.row0, o.n, t->rank, t->packed.d() + off);
          off += tot;
        }
        GH_HIP(hipGetLastError());
      }
      GH_HIP(hipStreamSynchronize(st));
    }
    tm[0] = ms_since();
    if (!H->world.wait()) { gh_set_error("aborted: another device failed"); return GH_ERR_HIP; }
    tm[1] = ms_since();
    // ---- every device pulls its rows of each ancestor's factors
    HSub& sub = h->sub;
    sub.depth = depth;
    sub.half.assign(depth, 0); sub.R.assign(depth, 0); sub.T.assign(depth, nullptr);
    sub.seed_off = r.seed_off;
    sub.ctx = &r; sub.allreduce = hm_allreduce; sub.local_done = hm_local_done;
    for (int l = 0; l < depth; ++l) {
      int R = 0;
      for (auto* t : H->top[l]) R = std::max(R, t->rank);
      HmTop* t = H->top[l][r.p >> (depth - l)];
      sub.R[l] = R;
      sub.half[l] = (r.row0 >= t->start + t->half) ? 1 : 0;
      GhBuf* sg = r.stage[l];
      GH_CHECK(sg->ensure(std::max<size_t>((size_t)r.n * R, 1) * sizeof(double)));
      sub.T[l] = sg->d();
      if (R > t->rank) GH_HIP(hipMemsetAsync(sg->d() + (size_t)r.n * t->rank, 0, (size_t)r.n * (R - t->rank) * sizeof(double), st));
      if (t->rank > 0) {
        const double* src = t->packed.d() + t->pack_off[r.p - t->first];
        const size_t bytes = (size_t)r.n * t->rank * sizeof(double);
        const int sdev = H->ranks[t->runner].dev;
        if (sdev == r.dev) GH_HIP(hipMemcpyAsync(sg->p, src, bytes, hipMemcpyDeviceToDevice, st));
        else GH_HIP(hipMemcpyPeerAsync(sg->p, r.dev, src, sdev, bytes, st));
      }
    }
    GH_HIP(hipStreamSynchronize(st));
    tm[2] = ms_since();
    if (!H->world.wait()) { gh_set_error("aborted: another device failed"); return GH_ERR_HIP; }
    tm[3] = ms_since();
    // ---- the sub-tree: the single-device code (released for the next sub-tree of this device once its own part is done)
    r.dev_lock = std::unique_lock<std::mutex>(g_hm_dev_mu[r.dev & 15]);
    tm[4] = ms_since();
    const int rcc = gh_hodlr_compute(h, &r.kern, x + r.row0 * ndim, r.n, ndim, yerr + r.row0, &r.ld);
    tm[5] = ms_since();
    return rcc;
  });
  if (rc != GH_OK) return rc;
  // log|det|: the sub-trees' own blocks in tree order, then the ancestors' cores bottom-up (each from the first device below it)
  double logdet = 0.0;
  for (auto& r : H->ranks) logdet += r.ld;
  for (int l = depth - 1; l >= 0; --l)
    for (auto* t : H->top[l]) logdet += H->ranks[t->first].h->sub.ld_top[l];
  H->logdet = logdet;
  // ranks, level by level
  H->all_ranks.clear();
  for (int l = 0; l < depth; ++l) for (auto* t : H->top[l]) H->all_ranks.push_back(t->rank);
  for (size_t l = depth;; ++l) {
    bool any = false;
    for (auto& r : H->ranks)
      if (l < r.h->levels.size()) { any = true; for (int v : r.h->levels[l]->ranks) H->all_ranks.push_back(v); }
    if (!any) break;
  }
  H->computed = true;
  H->top_n = n; H->top_min = min_size;
  if (logdet_out) *logdet_out = logdet;
  return GH_OK;
}

extern "C" int gh_hodlr_mgpu_solve(gh_hodlr_mgpu* H, const double* b, int64_t nrhs, double* out) {
  if (!H) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (!H->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  if (!b || !out || nrhs <= 0) { gh_set_error("bad argument to solve"); return GH_ERR_BAD_ARG; }
  if (gh_is_device_ptr(b) || gh_is_device_ptr(out)) { gh_set_error("HODLR split: b and out must be host pointers"); return GH_ERR_BAD_ARG; }
  // (n, nrhs) row-major: the rows of a sub-tree are one contiguous slice
  return hm_run(H, [&](HmRank& r) -> int { return gh_hodlr_solve(r.h, b + r.row0 * nrhs, nrhs, out + r.row0 * nrhs); });
}
extern "C" int gh_hodlr_mgpu_dot_solve(gh_hodlr_mgpu* H, const double* y, double* out) {
  if (!H || !y || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  if (!H->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  std::vector<double> a((size_t)H->n);
  GH_CHECK(gh_hodlr_mgpu_solve(H, y, 1, a.data()));
  double v = 0.0;
  for (int64_t i = 0; i < H->n; ++i) v += y[i] * a[i];
  *out = v;
  return GH_OK;
}
extern "C" int gh_hodlr_mgpu_ranks(const gh_hodlr_mgpu* H, int32_t* ranks_out, int32_t max_out, int32_t* n_out) {
  if (!H || !n_out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  int cnt = 0;
  for (int v : H->all_ranks) { if (ranks_out && cnt < max_out) ranks_out[cnt] = v; ++cnt; }
  *n_out = cnt < max_out ? cnt : max_out;
  return GH_OK;
}
extern "C" int gh_hodlr_mgpu_rows(const gh_hodlr_mgpu* H, int64_t* row0, int64_t* nrows) {
  if (!H || !row0 || !nrows) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  for (int p = 0; p < H->P; ++p) { row0[p] = H->ranks[p].row0; nrows[p] = H->ranks[p].n; }
  return GH_OK;
}
