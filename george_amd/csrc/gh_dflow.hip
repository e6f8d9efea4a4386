// gh_dflow.hip -- the dense factorisation as ONE persistent launch: tile tasks handed out from dependency-ordered
// queues, progress published in HBM counters.  For the sizes where the launch chain of gh_chol.hip is bound by its
// own boundaries (Np < 24576: every panel hand-over drains the chip, the chain waits for whole launches although it
// needs single tiles; DESIGN.md section 4, "The factorisation schedule").
//
// Same arithmetic as the launch chain, tile for tile: every tile (i, j) of the lower triangle receives
//     A_ij -= sum_{k < j} L_ik L_jk^T          k ascending, in slabs of 16 through the same MFMA sequence
//     L_ij  = A_ij L_jj^-T   (i > j)           or   L_jj, L_jj^-1 = potf2(A_jj)
// and since the accumulators of every product start from -C and end as C = -acc (exact), HOW the k range of a tile
// is cut into passes does not change a bit: the factor, its diagonal inverses and the log-determinant are
// bit-identical to factor_lookahead_deep()'s (tests/test_gpu_dataflow.py).
//
// Who does what.  One workgroup, a launch of its own, is the DIAGONAL worker: for j = 0, 1, ...: L_j,j-1 = A_j,j-1 L_j-1,j-1^-T
// (the sub-diagonal tile), A_jj -= L_j,j-1 L_j,j-1^T, potf2(A_jj) -- the critical path, on a CU of its own.  Every other
// workgroup is a WORKER that takes tasks from five queues (priority order; p = the panel being factored):
//   q0 : rows of panel p's diagonal block: the multiplies by L_jj^-T and the in-panel updates, one k step at a time;
//   q1 : the eight rows below them (the next diagonal block): the same, eagerly, and the steps that carry panel p into the
//        next block column's diagonal-block tiles -- so that those tiles are current when their panel begins;
//   q2 : rows further down -- per link j ONE task per half tile: the tile's in-panel k range and the multiply by L_jj^-T
//        (left-looking inside the panel, as the launch chain's rows-below stream);
//   q3 : the previous panel's contribution to the far tiles of the next block column, in four pieces as its columns complete;
//   q4 : everything older than the previous panel, 128 x 128 tiles, k ranges of up to 2048 (the bulk of the flops).
// WHO FINDS THE RUNNABLE TASKS: whoever finishes one.  The host gives every producer -- a task, the diagonal worker's
// sub-diagonal multiply of step j, its potf2 of step j -- the list of the tasks that read what it writes (its CANDIDATES:
// the next pass over the same tile, the products that take the finished L tile as an operand, the multiplies that wait for
// L_jj^-1).  After publishing its counters the producer's 256 threads check their candidates' inputs, one candidate per
// thread, and append the runnable ones to their queue's READY LIST (a plain array as long as the queue: nothing wraps; a
// note per task says "already listed").  The producer of a task's LAST input lists it -- at once, with no scan and no
// window.  A worker polls ONE cache line (the five lists' tails and heads, one wavefront transaction) and takes the first
// entry of the first non-empty list by compare-and-swap on its head.  (Forms tried before, profiles/r05/dataflow_*: workers
// claiming a runnable queue HEAD by compare-and-swap: one claim per 3.5 us, 8x slower than the launch chain; tickets per
// bucket with waiting owners: runnable tasks found late behind open buckets, 1.2-1.3x; every idle worker scanning windows
// of the queues: 500 x 420 loads per poll, 30x; one scanner workgroup filling the lists: windows that clog behind tasks
// whose inputs are late and a pass of latency per dependency hop, 3-8x.)
//
// Dependencies are not stored: they follow from a task's fields and three families of monotone counters,
//   D          diagonal steps finished (L_jj^-1 is readable when D > j),
//   rowh[i,h]  final L tiles in half-row (i, h), counted from column 0,
//   kd[t,h]    k steps applied to half-tile (t, h),
// written behind an agent-scope release by whoever finishes a task and polled relaxed, followed by ONE agent-scope
// acquire, by whoever wants to start one (/opt/skills/guides: Guideline 16's recipe, the one the retired persistent
// panel used bit-identically).  Forward progress: the queues are consistent with one topological order of all tasks
// (tests/test_dataflow_schedule.py replays them); the earliest unfinished task of that order has all its inputs finished,
// so the producer of its last input has listed it (or is about to) -- and a worker that is not resident has taken nothing.
// The diagonal worker is launched first and the workers' stream waits until it runs.  Every wait gives up after 2 s and
// raises the abort word.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "gh_common.h"
#include "gh_gemm_tile.h"
#include "gh_potf2_body.h"
#include "../../include/george_amd_debug.h"

#define DF_PW 8              // panel width in tiles (the launch chain's nb = 1024)
#define DF_NEARX 8           // rows [8 (p + 1), 8 (p + 1) + NEARX) are handled eagerly while panel p is factored
#define DF_NEARF 8           // tiles of block column p in rows < 8 p + NEARF take panel p - 1 one column at a time
#define DF_CHUNK 16          // lo queue: tile columns per task (K = 2048)
#define DF_NQ 5
#define DF_HALVES 0          // 1: the tasks next to the front as 64-row half tiles (measured: a K = 128 task is bound by its eight
                             //    dependent slab round trips, ~20 us whatever the tile height -- halves only double the task count)

struct DfTask {              // 16 bytes
  uint16_t i, j;             // tile
  uint16_t k0, k1;           // C -= L(i, k0:k1) L(j, k0:k1)^T   (k1 == k0: no update; then k0 = the k steps the tile must have)
  uint8_t half;              // 0, 1: rows [64 half, 64 half + 64) of the tile;  2: the whole tile
  uint8_t fin;               // then C <- C L_jj^-T (in place)
  uint16_t pad[3];
};
static_assert(sizeof(DfTask) == 16, "DfTask is one 16-byte load");
// counter words (unsigned), hot ones on lines of their own
#define DF_D 0
#define DF_ABORT 32
#define DF_KEY 64
#define DF_TAIL 128          // + q: entries of ready list q handed out to producers; + 8 + q: entries taken (the workers' heads)
#define DF_ROWH 320
static inline size_t df_off_kd(int nt) { return DF_ROWH + (size_t)((2 * nt + 31) / 32) * 32; }
static inline size_t df_off_lists(int nt) { return df_off_kd(nt) + (size_t)((nt * (nt + 1) + 31) / 32) * 32; }
// per task one "listed" note and one ready-list slot
static inline size_t df_words(int nt, size_t ntasks) { return df_off_lists(nt) + 2 * ntasks + 32; }

// ------------------------------------------------------------------------------------------------ the schedule (host)
struct DfSchedule {
  std::vector<DfTask> q[DF_NQ];
  // producers: task g (global index: queue by queue), then the diagonal worker's multiply of step j (total + j), then its potf2
  // of step j (total + nt + j); cand[cand_ptr[P] .. cand_ptr[P + 1]) = global indices of the tasks that read what P writes
  std::vector<uint32_t> cand_ptr, cand;
  size_t total() const { size_t n = 0; for (auto& v : q) n += v.size(); return n; }
};
static void df_build(int nt, DfSchedule& s) {
  typedef std::tuple<int, int, int, int, int, int> Key;
  std::vector<std::pair<Key, DfTask>> qs[DF_NQ];
  auto task = [](int i, int j, int k0, int k1, int half, int fin) {
    DfTask t; memset(&t, 0, sizeof(t));
    t.i = (uint16_t)i; t.j = (uint16_t)j; t.k0 = (uint16_t)k0; t.k1 = (uint16_t)k1; t.half = (uint8_t)half; t.fin = (uint8_t)fin;
    return t;
  };
  // a task next to the front: one whole tile, or its two halves
  auto near_task = [&](int q, Key key, int i, int j, int k0, int k1, int fin) {
    if (DF_HALVES) { for (int h = 0; h < 2; ++h) { std::get<5>(key) = h; qs[q].push_back({key, task(i, j, k0, k1, h, fin)}); } }
    else qs[q].push_back({key, task(i, j, k0, k1, 2, fin)});
  };
  // which queue the k = l step / the multiply at link l of a tile in row i belongs to: the row's place below panel l / 8
  auto rowq = [](int i, int link) { const int p = link / DF_PW; return i < DF_PW * (p + 1) ? 0 : (i < DF_PW * (p + 2) ? 1 : 2); };
  for (int j = 0; j < nt; ++j) {
    const int p = j / DF_PW, q0 = DF_PW * p;
    for (int i = j; i < nt; ++i) {
      const int kmax = std::max(0, i == j ? j - 1 : j);      // the workers' share of the tile's k range: [0, kmax)
      // (1) panels older than the previous one: the block column's group runs while panel p - 1 is factored (need order)
      if (p >= 2) {
        const int wend = std::min(DF_PW * (p - 1), kmax);
        for (int k0 = 0; k0 < wend; k0 += DF_CHUNK)
          qs[4].push_back({Key(p, i < DF_PW * (p + 1) ? 0 : 1, j, i, k0, 0), task(i, j, k0, std::min(k0 + DF_CHUNK, wend), 2, 0)});
      }
      // (2) the previous panel: eagerly for the tiles of the next diagonal block (q1, their rows are the eight below panel
      //     p - 1), in four pieces for the far ones (q3)
      if (p >= 1) {
        const int a0 = DF_PW * (p - 1), a1 = std::min(DF_PW * p, kmax);
        if (i < DF_PW * (p + 1)) {
          for (int k = a0; k < a1; ++k) near_task(1, Key(k, 1, j == k + 1 ? 1 : 0, i, j, 0), i, j, k, k + 1, 0);
        } else {
          static const int cut[5] = {0, 4, 6, 7, 8};
          for (int c = 0; c < 4; ++c) {
            const int k0 = a0 + cut[c], k1 = std::min(a0 + cut[c + 1], a1);
            if (k1 > k0) qs[3].push_back({Key(k1 - 1, j, i, 0, 0, 0), task(i, j, k0, k1, 2, 0)});
          }
        }
      }
      // (3) inside the panel: one k step at a time for the rows of this and of the next diagonal block
      if (kmax > q0 && i < DF_PW * (p + 2))
        for (int k = q0; k < kmax; ++k) near_task(rowq(i, k), Key(k, 1, j == k + 1 ? 1 : 0, i, j, 0), i, j, k, k + 1, 0);
      // (4) the multiply by L_jj^-T (rows j + 2 and below; row j + 1 is the diagonal worker's)
      if (i >= j + 2) {
        if (i < DF_PW * (p + 2)) near_task(rowq(i, j), Key(j, 0, 0, i, j, 0), i, j, j, j, 1);
        else {
          // far rows: the in-panel k range and the multiply as ONE task per HALF tile -- a row's eight tasks of a panel follow
          // one another (36 products), and as whole tiles (23-27 us per product beside a second workgroup on the CU) they took
          // longer than the diagonal worker needs for the panel: rows finished late, the bulk behind them stood still
          for (int h = 0; h < 2; ++h) qs[2].push_back({Key(j, i, h, 0, 0, 0), task(i, j, kmax > q0 ? q0 : j, j, h, 1)});
        }
      }
    }
  }
  for (int q = 0; q < DF_NQ; ++q) {
    std::stable_sort(qs[q].begin(), qs[q].end(), [](const std::pair<Key, DfTask>& a, const std::pair<Key, DfTask>& b) { return a.first < b.first; });
    s.q[q].clear();
    s.q[q].reserve(qs[q].size());
    for (auto& e : qs[q]) s.q[q].push_back(e.second);
  }
  // ---- candidates: for every task X and every input of X (df_ready()'s rule), X goes onto the list of that input's producer
  const size_t G = s.total();
  std::vector<std::vector<uint32_t>> lists(G + 2 * (size_t)nt);
  std::map<std::tuple<int, int, int, int>, uint32_t> kd_prod;       // (i, j, half 0|1, value) -> the task that brings the half tile to `value` steps
  std::map<std::tuple<int, int, int>, uint32_t> fin_prod;           // (i, j, half 0|1) -> the task that makes L(i, j) final
  {
    uint32_t g = 0;
    for (int q = 0; q < DF_NQ; ++q)
      for (const DfTask& t : s.q[q]) {
        const int h0 = t.half == 2 ? 0 : t.half, h1 = t.half == 2 ? 1 : t.half;
        for (int h = h0; h <= h1; ++h) {
          if (t.fin) fin_prod[std::make_tuple((int)t.i, (int)t.j, h)] = g;
          else kd_prod[std::make_tuple((int)t.i, (int)t.j, h, (int)t.k1)] = g;
        }
        ++g;
      }
  }
  auto final_producer = [&](int r, int c, int h) -> size_t {        // who makes half h of L(r, c) final
    if (r == c + 1) return G + (size_t)r;                           // the diagonal worker's multiply of step r
    return fin_prod.at(std::make_tuple(r, c, h));
  };
  {
    uint32_t g = 0;
    for (int q = 0; q < DF_NQ; ++q)
      for (const DfTask& t : s.q[q]) {
        const int h0 = t.half == 2 ? 0 : t.half, h1 = t.half == 2 ? 1 : t.half;
        std::vector<size_t> prods;
        if (t.k0 > 0) for (int h = h0; h <= h1; ++h) prods.push_back(kd_prod.at(std::make_tuple((int)t.i, (int)t.j, h, (int)t.k0)));
        if (t.k1 > t.k0) {
          const int c = t.k1 - 1;
          for (int h = 0; h < 2; ++h) prods.push_back(final_producer(t.j, c, h));
          if (t.i != t.j) for (int h = h0; h <= h1; ++h) prods.push_back(final_producer(t.i, c, h));
        }
        if (t.fin) prods.push_back(G + (size_t)nt + t.j);
        std::sort(prods.begin(), prods.end());
        prods.erase(std::unique(prods.begin(), prods.end()), prods.end());
        for (size_t P : prods) lists[P].push_back(g);
        ++g;
      }
  }
  s.cand_ptr.assign(1, 0u);
  s.cand.clear();
  for (auto& l : lists) { s.cand.insert(s.cand.end(), l.begin(), l.end()); s.cand_ptr.push_back((uint32_t)s.cand.size()); }
}

// the schedule of an nt x nt tile matrix, for the replay in tests/test_dataflow_schedule.py (host only, no device needed):
// counts[q] = tasks of queue q (DF_NQ = 5 queues); out (when not NULL): rows of 7 ints {queue, i, j, k0, k1, half, fin}, queue by
// queue in need order, at most max_rows of them
extern "C" int gh_debug_dflow_schedule(int32_t nt, int32_t* counts, int32_t* out, int64_t max_rows) {
  if (nt <= 0 || nt > 4096 || !counts) { gh_set_error("dflow_schedule: bad argument"); return GH_ERR_BAD_ARG; }
  DfSchedule s;
  df_build(nt, s);
  int64_t r = 0;
  for (int q = 0; q < DF_NQ; ++q) {
    counts[q] = (int32_t)s.q[q].size();
    if (!out) continue;
    for (const DfTask& t : s.q[q]) {
      if (r >= max_rows) return GH_OK;
      int32_t* o = out + 7 * r++;
      o[0] = q; o[1] = t.i; o[2] = t.j; o[3] = t.k0; o[4] = t.k1; o[5] = t.half; o[6] = t.fin;
    }
  }
  return GH_OK;
}

// the candidate lists of the same schedule: ptr (when not NULL) receives total + 2 nt + 1 offsets, cand the entries (at most
// max_cand); *n_cand = the number of entries
extern "C" int gh_debug_dflow_candidates(int32_t nt, uint32_t* ptr, uint32_t* cand, int64_t max_cand, int64_t* n_cand) {
  if (nt <= 0 || nt > 4096 || !n_cand) { gh_set_error("dflow_candidates: bad argument"); return GH_ERR_BAD_ARG; }
  DfSchedule s;
  df_build(nt, s);
  *n_cand = (int64_t)s.cand.size();
  if (ptr) memcpy(ptr, s.cand_ptr.data(), s.cand_ptr.size() * sizeof(uint32_t));
  if (cand) memcpy(cand, s.cand.data(), (size_t)std::min<int64_t>(max_cand, (int64_t)s.cand.size()) * sizeof(uint32_t));
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------ the kernel
struct DfArgs {
  double* A; long ld;
  double* dinv;
  long long* info;
  unsigned* cnt;
  const DfTask* tasks[DF_NQ];
  unsigned count[DF_NQ];
  const uint32_t* cand_ptr;      // candidates of producer P: cand[cand_ptr[P] .. cand_ptr[P + 1])
  const uint32_t* cand;
  unsigned qoff[DF_NQ + 1];      // global index of the first task of queue q; qoff[DF_NQ] = all tasks
  unsigned off_done[DF_NQ];      // cnt + off_done[q] + x: task x of queue q is on its ready list
  unsigned off_list[DF_NQ];      // cnt + off_list[q] + e: entry e of ready list q = task index + 1 (0: not written yet)
  unsigned off_kd;
  int nt;
  unsigned long long* trace;     // debugging aid (gh_debug_dflow_trace): [0] = records used, then 4 words per record; NULL: off
  unsigned trace_cap;
};
// one record: {start, end} in wall_clock64() ticks (100 MHz), {i | j << 16 | k0 << 32 | k1 << 48}, {kind | half << 8 | fin << 16 | block << 32};
// kind 0-2 = queue of a worker's task, 8 = diagonal worker waiting, 9 = its sub-diagonal multiply, 10 = its update, 11 = potf2
__device__ __forceinline__ void df_trace(const DfArgs& a, long long t0, unsigned i, unsigned j, unsigned k0, unsigned k1,
                                         unsigned kind, unsigned half, unsigned fin) {
  if (!a.trace || threadIdx.x != 0) return;
  const long long t1 = wall_clock64();
  const unsigned long long slot = __hip_atomic_fetch_add(a.trace, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (slot >= a.trace_cap) return;
  unsigned long long* r = a.trace + 1 + 4 * slot;
  r[0] = (unsigned long long)t0; r[1] = (unsigned long long)t1;
  r[2] = (unsigned long long)i | ((unsigned long long)j << 16) | ((unsigned long long)k0 << 32) | ((unsigned long long)k1 << 48);
  r[3] = (unsigned long long)kind | ((unsigned long long)half << 8) | ((unsigned long long)fin << 16) | ((unsigned long long)blockIdx.x << 32);
}

__device__ __forceinline__ unsigned df_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void df_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define DF_TIMEOUT_TICKS 200000000LL          // wall_clock64() runs at 100 MHz: 2 s

// which CU this wavefront runs on: XCC_ID and the CU / SH / SE fields of HW_ID (bits 8..15), never 0
__device__ __forceinline__ unsigned df_cu_key() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return 0x80000000u | ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
}

// all inputs of `t` there?  (one lane; the loads are issued together)
__device__ __forceinline__ bool df_ready(const DfArgs& a, const DfTask& t) {
  const unsigned* const cnt = a.cnt;
  const int h0 = t.half == 2 ? 0 : t.half, h1 = t.half == 2 ? 1 : t.half;
  const unsigned* const kd = cnt + a.off_kd + 2u * ((unsigned)t.i * (t.i + 1u) / 2u + t.j);
  const unsigned* const rh = cnt + DF_ROWH;
  const unsigned v0 = t.k0 ? df_ld(kd + h0) : 0xffffu, v1 = t.k0 ? df_ld(kd + h1) : 0xffffu;
  unsigned b0 = 0xffffu, b1 = 0xffffu, a0 = 0xffffu, a1 = 0xffffu;
  if (t.k1 > t.k0) {
    b0 = df_ld(rh + 2 * t.j); b1 = df_ld(rh + 2 * t.j + 1);
    if (t.i != t.j) { a0 = df_ld(rh + 2 * t.i + h0); a1 = df_ld(rh + 2 * t.i + h1); }
  }
  const unsigned d = t.fin ? df_ld(cnt + DF_D) : 0xffffu;
  return (v0 >= t.k0) & (v1 >= t.k0) & (b0 >= t.k1) & (b1 >= t.k1) & (a0 >= t.k1) & (a1 >= t.k1) & (d >= t.j + 1u);
}

// results of the calling workgroup to memory, then the counters (Guideline 16: plain stores, every wavefront drained,
// barrier, one lane: agent release, drained again -- inline assembly, the compiler may drop a wait it can prove
// redundant -- then the relaxed stores)
__device__ __forceinline__ void df_publish(unsigned* cnt, unsigned* w0, unsigned v0, unsigned* w1, unsigned v1) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    df_st(w0, v0);
    if (w1) df_st(w1, v1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the counters are out before anybody looks at the candidates
  }
}

// After df_publish(): the calling workgroup's threads check the inputs of producer P's candidates, one candidate per thread,
// and list the runnable ones.  Two producers that finish a task's last two inputs at the same time both find it runnable
// (each has its own counters out -- waited for -- before it reads the other's); the note's compare-and-swap lets one list it.
__device__ __forceinline__ void df_list_candidates(const DfArgs& a, unsigned P) {
  __syncthreads();
  unsigned* const cnt = a.cnt;
  const unsigned beg = a.cand_ptr[P], end = a.cand_ptr[P + 1];
  for (unsigned c = beg + threadIdx.x; c < end; c += blockDim.x) {
    const unsigned g = a.cand[c];
    int q = 0;
#pragma unroll
    for (int v = 1; v < DF_NQ; ++v) q += g >= a.qoff[v] ? 1 : 0;
    const unsigned x = g - a.qoff[q];
    unsigned* const note = cnt + a.off_done[q] + x;
    if (df_ld(note) != 0u) continue;
    const DfTask t = a.tasks[q][x];
    if (!df_ready(a, t)) continue;
    unsigned expect = 0u;
    if (!__hip_atomic_compare_exchange_strong(note, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
    const unsigned slot = __hip_atomic_fetch_add(cnt + DF_TAIL + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    df_st(cnt + a.off_list[q] + slot, x + 1u);
  }
}

// the diagonal worker's wait for up to four counters; false: aborted
__device__ __forceinline__ bool df_wait4(const DfArgs& a, const unsigned* w, unsigned need, int n, int* s_flag) {
  if (threadIdx.x == 0) {
    int good = 1;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
      unsigned lo = 0xffffffffu;
      for (int q = 0; q < n; ++q) { const unsigned v = df_ld(w + q); lo = v < lo ? v : lo; }
      if (lo >= need) break;
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63u) == 0u) {
        if (df_ld(a.cnt + DF_ABORT) != 0u) { good = 0; break; }
        if (wall_clock64() - t0 > DF_TIMEOUT_TICKS) { df_st(a.cnt + DF_ABORT, 2u); good = 0; break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = good;
  }
  __syncthreads();
  // (through a scalar register: the compiler must SEE that every lane takes the same way out of the caller's loop, or it
  //  is free to split the lanes over two loops -- see the workers' loop)
  const bool r = __builtin_amdgcn_readfirstlane(*s_flag) != 0;
  __syncthreads();
  return r;
}

// The diagonal worker: ONE workgroup, a launch of its own, one wavefront per SIMD (the 128 x 128 kernel needs all 256 vector
// registers: inlined into the workers' function it pushed 142 of them into scratch; as a called function it does not compile
// -- its inline assembly wants wave-uniform values in scalar registers; with the whole register file of a CU the compiler
// parks what the loop keeps alive in accumulation registers).  It is launched FIRST and the workers' stream waits until it runs
// (dflow_gate_kernel).
template <bool TR>
__global__ __launch_bounds__(256, 1) void dflow_diag_kernel(DfArgs a) {
  __shared__ __attribute__((aligned(1024))) double smem[9728];        // potf2: 8256 + 1288 doubles; GEMM: 8192
  __shared__ int s_state;
  const int tid = threadIdx.x;
  unsigned* const cnt = a.cnt;
  const long ld = a.ld;
  {
    if (tid == 0) df_st(cnt + DF_KEY, df_cu_key());
    __builtin_amdgcn_s_setprio(3);
    double* const s = smem; double* const dscr = smem + GH_POTF2_S_DOUBLES;
    int* const fail_at = (int*)(smem + 9600);
    for (int j = 0; j < a.nt; ++j) {
      double* const Ajj = a.A + (long)j * 128 * ld + (long)j * 128;
      long long tt = TR ? wall_clock64() : 0;
      if (j > 0) {
        double* const Asub = Ajj - 128;                                // tile (j, j - 1)
        const double* const dprev = a.dinv + (long)(j - 1) * 128 * 128;
        if (j > 1) {
          // the workers' share of both tiles: k < j - 1   (kd of (j, j - 1) and (j, j) are four consecutive words)
          const unsigned* const kd = cnt + a.off_kd + 2u * ((unsigned)j * (j + 1u) / 2u + (j - 1));
          if (!df_wait4(a, kd, (unsigned)(j - 1), 4, &s_state)) return;
          if (TR) { df_trace(a, tt, j, j, 0, 0, 8, 2, 0); tt = wall_clock64(); }
        }
        gh_tile128_nt<false>(smem, Asub, ld, Asub, ld, dprev, 128, 128);                 // L(j, j-1), in place
        df_publish(cnt, cnt + DF_ROWH + 2 * j, (unsigned)j, cnt + DF_ROWH + 2 * j + 1, (unsigned)j);
        df_list_candidates(a, a.qoff[DF_NQ] + (unsigned)j);
        if (TR) { df_trace(a, tt, j, j - 1, 0, 0, 9, 2, 1); tt = wall_clock64(); }
        gh_tile128_nt<true>(smem, Ajj, ld, Asub, ld, Asub, ld, 128);                     // A_jj -= L(j, j-1) L(j, j-1)^T
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (TR) { df_trace(a, tt, j, j, j - 1, j, 10, 2, 0); tt = wall_clock64(); }
      }
      const bool ok = gh_potf2::potf2_body(Ajj, ld, a.dinv + (long)j * 128 * 128, a.info, (long long)j * 128, s, dscr, fail_at);
      if (!ok) {                                                      // (uniform) not positive definite: everybody leaves
        __syncthreads();
        if (tid == 0) df_st(cnt + DF_ABORT, 1u);
        return;
      }
      df_publish(cnt, cnt + DF_D, (unsigned)(j + 1), nullptr, 0u);
      df_list_candidates(a, a.qoff[DF_NQ] + (unsigned)a.nt + (unsigned)j);
      if (TR) df_trace(a, tt, j, j, 0, 0, 11, 2, 0);
    }
  }
}

template <bool TR>
__global__ __launch_bounds__(256, 2) void dflow_worker_kernel(DfArgs a) {
  __shared__ __attribute__((aligned(1024))) double smem[8192];
  __shared__ DfTask s_task;
  __shared__ int s_state;
  __shared__ unsigned s_gid;               // global index of the task taken (= its producer number)
  const int tid = threadIdx.x;
  unsigned* const cnt = a.cnt;
  const long ld = a.ld;
  const unsigned mykey = df_cu_key();
  if (tid == 0) {
    // the workgroup that shares the diagonal worker's CU leaves (its matrix instructions would halve potf2's rate);
    // its key is there within microseconds of its launch -- bounded anyway (~0.5 ms)
    unsigned k = 0;
    for (int spins = 0; spins < 4000 && (k = df_ld(cnt + DF_KEY)) == 0u; ++spins) __builtin_amdgcn_s_sleep(4);
    s_state = (k == mykey) ? -1 : 0;
  }
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_state) < 0) return;
  __syncthreads();

  for (;;) {
    if (tid < 64) {
      // ---- take the first entry of the first non-empty ready list.  The five tails and the five heads share ONE cache line and
      // the wavefront reads it as one transaction (lane x loads word x): 500 polling workgroups reading ten words each, one load
      // per word, saturated that line's memory channel and every task on the chip ran 3-4x slower (profiles/r05/dataflow_trace_session_l.log)
      const int lane = tid;
      int st = 0;
      const long long t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        unsigned word = df_ld(cnt + (lane < 16 ? DF_TAIL + lane : DF_ABORT));
        if (__builtin_amdgcn_readlane(word, 16) != 0u) { st = -1; break; }                      // (lanes >= 16 read the abort word)
#ifndef DF_NO_DESYNC
        {
          // something is listed: every idle workgroup sees it within a poll period and would fire a compare-and-swap at the same
          // word.  Wait a pseudo-random 0-2 us and look again: most find the entry gone and never touch the head.
          bool any = false;
#pragma unroll
          for (int q = 0; q < DF_NQ; ++q) any = any || __builtin_amdgcn_readlane(word, 8 + q) < __builtin_amdgcn_readlane(word, q);
          if (any && spins > 0u) {
            const unsigned d = (blockIdx.x * 2654435761u + spins * 40503u) >> 27;        // 0 .. 31
            for (unsigned z = 0; z < d; ++z) __builtin_amdgcn_s_sleep(2);
            word = df_ld(cnt + (lane < 16 ? DF_TAIL + lane : DF_ABORT));
          }
        }
#endif
        bool alldone = true;
#pragma unroll
        for (int q = 0; q < DF_NQ; ++q) {
          if (st != 0) continue;
          const unsigned tail = __builtin_amdgcn_readlane(word, q);
          unsigned h = __builtin_amdgcn_readlane(word, 8 + q);
          if (h < a.count[q]) alldone = false;
          while (h < tail) {                                        // (wave-uniform loop; lane 0 acts)
            unsigned got = 0u, now = h;
            if (lane == 0) {
              unsigned expect = h;
              got = __hip_atomic_compare_exchange_strong(cnt + DF_TAIL + 8 + q, &expect, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
              now = expect;
            }
            got = __builtin_amdgcn_readfirstlane(got); now = __builtin_amdgcn_readfirstlane(now);
            if (got) {
              unsigned e = 0u;                                      // (the entry was stored before the tail moved past it)
              for (int tries = 0; tries < 200000 && (e = __builtin_amdgcn_readfirstlane(df_ld(cnt + a.off_list[q] + h))) == 0u; ++tries) __builtin_amdgcn_s_sleep(1);
              if (e == 0u) { if (lane == 0) df_st(cnt + DF_ABORT, 2u); st = -1; break; }
              if (lane == 0) { s_task = a.tasks[q][e - 1u]; s_gid = a.qoff[q] + e - 1u; }
              st = 1 + q; break;
            }
            h = now;
          }
        }
        if (st) break;
        if (alldone) { st = -1; break; }           // every task has been taken
#ifndef DF_IDLE_LONG
#define DF_IDLE_LONG 2
#endif
        if (spins < 4) __builtin_amdgcn_s_sleep(16); else if (spins < 16) __builtin_amdgcn_s_sleep(64);
        else { for (int z = 0; z < DF_IDLE_LONG; ++z) __builtin_amdgcn_s_sleep(127); }
        if ((++spins & 63u) == 0u && wall_clock64() - t0 > DF_TIMEOUT_TICKS) { if (lane == 0) df_st(cnt + DF_ABORT, 2u); st = -1; break; }
      }
      if (st > 0 && lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (lane == 0) s_state = st;
    }
    __syncthreads();
    // The way out of the loop must be UNIFORM for the compiler too: s_state comes from LDS, a per-lane value as far as it knows,
    // and with a per-lane exit it restructured this loop into two -- the lanes that never enter the block above circling
    // through the barrier and the task below on their own, lane 0 parked outside for ever: the scanner form's first build
    // re-ran its first task ~10^5 times per second with the result never published (profiles/r05/dataflow_lane_split.md).
    const int state = __builtin_amdgcn_readfirstlane(s_state);
    if (state < 0) return;
    // (the task's fields as SCALARS: every address below is wave-uniform, and the tile functions want their operand bases
    //  in scalar registers -- a buffer descriptor built from a vector register costs a v_readfirstlane per DMA)
    const unsigned* const tw = (const unsigned*)&s_task;
    const unsigned w0 = __builtin_amdgcn_readfirstlane(tw[0]), w1 = __builtin_amdgcn_readfirstlane(tw[1]),
                   w2 = __builtin_amdgcn_readfirstlane(tw[2]);
    struct { unsigned i, j, k0, k1, half, fin; } t = {w0 & 0xffffu, w0 >> 16, w1 & 0xffffu, w1 >> 16, w2 & 0xffu, (w2 >> 8) & 0xffu};
    const int s_q = state - 1;
    const long long tt = TR ? wall_clock64() : 0;
    __syncthreads();
    const int r0 = t.half == 1 ? 64 : 0;
    double* const C = a.A + ((long)t.i * 128 + r0) * ld + (long)t.j * 128;
    unsigned* const kd = cnt + a.off_kd + 2u * ((unsigned)t.i * (t.i + 1u) / 2u + t.j);
    if (t.k1 > t.k0) {
      const double* const Ao = a.A + ((long)t.i * 128 + r0) * ld + (long)t.k0 * 128;
      const double* const Bo = a.A + (long)t.j * 128 * ld + (long)t.k0 * 128;
      const long K = (long)(t.k1 - t.k0) * 128;
      if (t.half == 2) gh_tile128_nt<true>(smem, C, ld, Ao, ld, Bo, ld, K);
      else gh_tile64_nt<true>(smem, C, ld, Ao, ld, Bo, ld, K);
    }
    if (t.fin) {
      if (t.k1 > t.k0) {
        // the rows just written are this product's A operand: stores done, the CU's L1 dropped
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
      }
      if (t.half == 2) {
        gh_tile128_nt<false>(smem, C, ld, C, ld, a.dinv + (long)t.j * 128 * 128, 128, 128);
        df_publish(cnt, cnt + DF_ROWH + 2 * t.i, (unsigned)t.j + 1u, cnt + DF_ROWH + 2 * t.i + 1, (unsigned)t.j + 1u);
      } else {
        gh_tile64_nt<false>(smem, C, ld, C, ld, a.dinv + (long)t.j * 128 * 128, 128, 128);
        df_publish(cnt, cnt + DF_ROWH + 2 * t.i + t.half, (unsigned)t.j + 1u, nullptr, 0u);
      }
    } else if (t.half == 2) {
      df_publish(cnt, kd, t.k1, kd + 1, t.k1);
    } else {
      df_publish(cnt, kd + t.half, t.k1, nullptr, 0u);
    }
    df_list_candidates(a, __builtin_amdgcn_readfirstlane(s_gid));
    // (no debugging counters on this path: four fire-and-forget atomic adds per task on ONE cache line -- 80 per microsecond chip-wide,
    //  the rate at which a line saturates -- stood in front of every s_waitcnt vmcnt(0) of the publication: every task on the chip
    //  2-3x slower, sessions j-n of profiles/r05)
    if (TR) df_trace(a, tt, t.i, t.j, t.k0, t.k1, (unsigned)s_q, t.half, t.fin);
  }
}

// On the workers' stream, in front of their launch: returns when the diagonal worker is RUNNING (its CU key is there).  That
// workgroup needs a CU to itself (512 registers per wavefront: the 128 x 128 kernel alone fills 256, and what the loop around
// it keeps alive would go to scratch); launched second it could find every CU taken by workers that wait for it.  Gives up
// after 20 ms (the workers then run into their own time-out if the diagonal worker never comes).
__global__ void dflow_gate_kernel(const unsigned* cnt) {
  const long long t0 = wall_clock64();
  while (df_ld(cnt + DF_KEY) == 0u && wall_clock64() - t0 < 2000000LL) __builtin_amdgcn_s_sleep(8);
}

// the time-out of a wait (not a property of the matrix) as an impossible minor index: compute_finish() of gh_chol.hip tells it apart
__global__ void dflow_check_kernel(const unsigned* cnt, long long* info) {
  if (cnt[DF_ABORT] == 2u && *info == 0) *info = GH_DFLOW_TIMEOUT_INFO;
}

// ------------------------------------------------------------------------------------------------ host side
struct DfDeviceSchedule {
  DfTask* d[DF_NQ] = {};
  unsigned count[DF_NQ] = {};
  uint32_t* cand_ptr = nullptr;
  uint32_t* cand = nullptr;
};
static std::mutex g_df_mutex;
static std::map<std::pair<int, int>, DfDeviceSchedule> g_df_cache;     // (device, nt): never freed (a few MB per size)
static std::map<int, size_t> g_df_total;                               // nt -> tasks of all queues

// debugging aid: the next factorisations record one line per task (see df_trace); single-threaded use
static long g_df_trace_cap = 0;
static unsigned long long* g_df_trace = nullptr;
extern "C" int gh_debug_dflow_trace(int64_t capacity, uint64_t* out, int64_t max_records, int64_t* n_out) {
  if (out && n_out && g_df_trace) {                       // read the last factorisation's records (after a synchronisation)
    unsigned long long used = 0;
    GH_HIP(hipMemcpy(&used, g_df_trace, sizeof(used), hipMemcpyDeviceToHost));
    const int64_t n = std::min<int64_t>(std::min<int64_t>((int64_t)used, g_df_trace_cap), max_records);
    if (n > 0) GH_HIP(hipMemcpy(out, g_df_trace + 1, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *n_out = n;
  } else if (n_out) *n_out = 0;
  if (capacity >= 0 && capacity != g_df_trace_cap) {
    if (g_df_trace) { (void)hipDeviceSynchronize(); (void)hipFree(g_df_trace); g_df_trace = nullptr; }
    g_df_trace_cap = (long)capacity;
  }
  return GH_OK;
}

static size_t df_total_tasks(int nt) {
  std::lock_guard<std::mutex> lk(g_df_mutex);
  auto it = g_df_total.find(nt);
  if (it != g_df_total.end()) return it->second;
  DfSchedule s;
  df_build(nt, s);
  return g_df_total[nt] = s.total();
}
size_t gh_dflow_counter_bytes(int64_t np) { const int nt = (int)(np / 128); return df_words(nt, df_total_tasks(nt)) * sizeof(unsigned); }

// factor the np x np matrix at A in place (lower triangle; np a multiple of 128), dinv[j] = L_jj^-1; counters: at least
// gh_dflow_counter_bytes(np) of device memory that nothing else uses until the streams have passed this call.
// `st`: the stream the matrix was built on -- the workers' launch goes there; `sd`: a second stream for the diagonal
// worker's launch (the two must run side by side); ev[2]: two events of the caller.  On return everything is joined
// on `sd` (the diagonal worker finishes last by construction), where the caller continues.
// ONE dataflow factorisation per device at a time (the caller holds gh_dflow_mutex(device) until it has synchronised):
// the workers of two of them could fill the chip before either diagonal worker is placed.
int gh_dflow_factor(double* A, int64_t ld, int64_t np, double* dinv, long long* d_info, unsigned* counters,
                    hipStream_t st, hipStream_t sd, hipEvent_t* ev) {
  const int nt = (int)(np / 128);
  if (np % 128 || nt <= 0 || nt > 4096) { gh_set_error("dflow: np must be a multiple of 128"); return GH_ERR_BAD_ARG; }
  if (!sd || sd == st || !ev) { gh_set_error("dflow: needs a second stream"); return GH_ERR_BAD_ARG; }
  int dev = 0;
  GH_HIP(hipGetDevice(&dev));
  DfDeviceSchedule ds;
  {
    std::lock_guard<std::mutex> lk(g_df_mutex);
    auto it = g_df_cache.find({dev, nt});
    if (it == g_df_cache.end()) {
      DfSchedule s;
      df_build(nt, s);
      DfDeviceSchedule n;
      for (int q = 0; q < DF_NQ; ++q) {
        n.count[q] = (unsigned)s.q[q].size();
        if (s.q[q].empty()) continue;
        GH_HIP(hipMalloc((void**)&n.d[q], s.q[q].size() * sizeof(DfTask)));
        GH_HIP(hipMemcpy(n.d[q], s.q[q].data(), s.q[q].size() * sizeof(DfTask), hipMemcpyHostToDevice));
      }
      GH_HIP(hipMalloc((void**)&n.cand_ptr, s.cand_ptr.size() * sizeof(uint32_t)));
      GH_HIP(hipMemcpy(n.cand_ptr, s.cand_ptr.data(), s.cand_ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      GH_HIP(hipMalloc((void**)&n.cand, std::max<size_t>(1, s.cand.size()) * sizeof(uint32_t)));
      if (!s.cand.empty()) GH_HIP(hipMemcpy(n.cand, s.cand.data(), s.cand.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      g_df_total[nt] = s.total();
      it = g_df_cache.emplace(std::make_pair(dev, nt), n).first;
    }
    ds = it->second;
  }
  static int ncu = 0;
  if (ncu == 0) {
    hipDeviceProp_t prop;
    GH_HIP(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  DfArgs a;
  a.A = A; a.ld = (long)ld; a.dinv = dinv; a.info = d_info; a.cnt = counters;
  unsigned at = (unsigned)df_off_lists(nt);
  size_t total = 0;
  for (int q = 0; q < DF_NQ; ++q) {
    a.tasks[q] = ds.d[q]; a.count[q] = ds.count[q];
    a.off_done[q] = at; at += ds.count[q];
    a.off_list[q] = at; at += ds.count[q];
    a.qoff[q] = (unsigned)total;
    total += ds.count[q];
  }
  a.qoff[DF_NQ] = (unsigned)total;
  a.cand_ptr = ds.cand_ptr; a.cand = ds.cand;
  a.off_kd = (unsigned)df_off_kd(nt); a.nt = nt;
  a.trace = nullptr; a.trace_cap = 0;
  if (g_df_trace_cap > 0) {
    if (!g_df_trace) GH_HIP(hipMalloc((void**)&g_df_trace, (1 + 4 * (size_t)g_df_trace_cap) * sizeof(unsigned long long)));
    GH_HIP(hipMemsetAsync(g_df_trace, 0, sizeof(unsigned long long), st));
    a.trace = g_df_trace; a.trace_cap = (unsigned)g_df_trace_cap;
  }
  GH_HIP(hipMemsetAsync(counters, 0, df_words(nt, total) * sizeof(unsigned), st));
  GH_HIP(hipEventRecord(ev[0], st));
  GH_HIP(hipStreamWaitEvent(sd, ev[0], 0));
  if (a.trace) hipLaunchKernelGGL(dflow_diag_kernel<true>, dim3(1), dim3(256), 0, sd, a);
  else hipLaunchKernelGGL(dflow_diag_kernel<false>, dim3(1), dim3(256), 0, sd, a);
  GH_HIP(hipGetLastError());
  hipLaunchKernelGGL(dflow_gate_kernel, dim3(1), dim3(1), 0, st, (const unsigned*)counters);
  GH_HIP(hipGetLastError());
  // two workgroups per CU (64 KiB of LDS each); the diagonal worker's CU takes none (its registers are gone)
  static const int nwork = getenv("GEORGE_AMD_DATAFLOW_WORKERS") ? std::max(1, atoi(getenv("GEORGE_AMD_DATAFLOW_WORKERS"))) : 0;
  const unsigned grid = (unsigned)(nwork > 0 ? nwork : 2 * ncu - 2);
  if (total > 0) {
    if (a.trace) hipLaunchKernelGGL(dflow_worker_kernel<true>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dflow_worker_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    GH_HIP(hipGetLastError());
  }
  GH_HIP(hipEventRecord(ev[1], st));
  GH_HIP(hipStreamWaitEvent(sd, ev[1], 0));
  hipLaunchKernelGGL(dflow_check_kernel, dim3(1), dim3(1), 0, sd, (const unsigned*)counters, d_info);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

std::mutex& gh_dflow_mutex(int device) {
  static std::mutex m[64];
  return m[device >= 0 && device < 64 ? device : 0];
}
