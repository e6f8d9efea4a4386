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
// Who does what.  Workgroup 0 is the DIAGONAL worker: for j = 0, 1, ...: L_j,j-1 = A_j,j-1 L_j-1,j-1^-T (the
// sub-diagonal tile), A_jj -= L_j,j-1 L_j,j-1^T, potf2(A_jj) -- the critical path, on a CU of its own (the workgroup
// that shares its CU leaves at once).  Every other workgroup is a WORKER that takes tasks from three queues, in
// priority order:
//   crit : tiles next to the front (rows that the diagonal worker needs within a panel): one k step at a time, as
//          64-row half tiles; and the multiplies by L_jj^-T of those rows;
//   hi   : rows far below the front -- per link j ONE task per half tile: the tile's in-panel k range and the
//          multiply by L_jj^-T (left-looking inside the panel, as the launch chain's rows-below stream) -- and the
//          previous panel's contribution to the next block column, in four pieces as its columns complete;
//   lo   : everything older than the previous panel, 128 x 128 tiles, k ranges of up to 2048 (the bulk of the flops).
// Every queue is cut into BUCKETS with a gate (a counter that must have reached a value: "L_jj^-1 is there"); inside an open
// bucket tasks are handed out by a fetch-and-add ticket -- no compare-and-swap chain: the first form claimed only a RUNNABLE
// head, one claim per ~3.5 us whatever the number of idle workgroups, 8x slower than the launch chain (profiles/r05/
// dataflow.md, session a).  A claimed task whose inputs are not there yet is waited for briefly and then HELD: its owner
// goes on serving the other queues (one held task per queue and workgroup) and starts it when it becomes runnable.
//
// Dependencies are not stored: they follow from a task's fields and three families of monotone counters,
//   D          diagonal steps finished (L_jj^-1 is readable when D > j),
//   rowh[i,h]  final L tiles in half-row (i, h), counted from column 0,
//   kd[t,h]    k steps applied to half-tile (t, h),
// written behind an agent-scope release by whoever finishes a task and polled relaxed, followed by ONE agent-scope
// acquire, by whoever wants to start one (/opt/skills/guides: Guideline 16's recipe, the one the retired persistent
// panel used bit-identically).  Forward progress: the queues are consistent with one topological order of all tasks
// (tests/test_dataflow_schedule.py replays them), lo's tickets are claimed in that order, crit and hi only when
// runnable -- so the earliest unfinished task is always either running or claimable by the next free workgroup, and a
// workgroup that is not resident has claimed nothing.  Every wait gives up after 2 s and raises the abort word.
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
#define DF_NEARF 16          // tiles of block column p in rows < 8 p + NEARF take panel p - 1 one column at a time (queue 1)
#define DF_NQ 5
#define DF_AHEAD 8            // tickets of a ready list that may be out beyond its listed entries (workgroups waiting at their slots)
#define DF_NLIST 2           // queues 0, 1: no buckets, their tasks reach the workers through ready lists (see DfSchedule)
#define DF_NPEEK 3           // queues 0 .. 2 hand out a ticket only for a runnable next task; 3 and 4 in order, their owners wait
#define DF_LO_NEAR 3         // passes of panel q into block columns <= q + DF_LO_NEAR go first (queue 3), the rest after them (queue 4)
#define DF_SCAN 8            // buckets a claim looks into, beyond the used-up ones
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
struct DfBucket {            // tasks [start, start + size) of a queue, claimable once cnt[gate_word] >= gate_val
  uint32_t start, size, gate_word, gate_val;
};

// counter words (unsigned), hot ones on lines of their own
#define DF_D 0
#define DF_ABORT 32
#define DF_KEY 64
#define DF_HEAD 96           // + 32 q
#define DF_STAT 192          // [0] tasks run, [1] idle polls (debug)
#define DF_TAIL 240           // + q: entries of ready list q handed out to producers; + 4 + q: tasks of queue q kept by their producers (never listed); + 8 + q: tickets taken (q < DF_NLIST)
#define DF_ROWH 256
static inline size_t df_off_kd(int nt) { return DF_ROWH + (size_t)((2 * nt + 31) / 32) * 32; }
static inline size_t df_off_next(int nt) { return df_off_kd(nt) + (size_t)((nt * (nt + 1) + 31) / 32) * 32; }   // per-bucket ticket counters
#define DF_LISTED_PER_LINK 320                // bound on the listed tasks (queues 0, 1) per link: checked when the schedule is built
static inline size_t df_off_notes(int nt) { return df_off_next(nt) + 2 * (size_t)nt + 32; }   // (a hi bucket per link, two lo buckets)
// per listed task one "listed" note and one ready-list slot
static inline size_t df_words(int nt) { return df_off_notes(nt) + 2 * (size_t)DF_LISTED_PER_LINK * nt + 64; }

// ------------------------------------------------------------------------------------------------ the schedule (host)
struct DfSchedule {
  std::vector<DfTask> q[DF_NQ];
  std::vector<DfBucket> b[DF_NQ];          // queues DF_NLIST .. : tickets per bucket
  // queues 0 .. DF_NLIST - 1 are LISTED: whoever finishes one of their tasks' inputs checks the task and, when it is runnable,
  // appends it to the queue's ready list.  Producers: task g (global index: queue by queue), then the diagonal worker's
  // multiply of step j (total + j), then its potf2 of step j (total + nt + j); cand[cand_ptr[P] .. cand_ptr[P + 1]) = global
  // indices of the listed tasks that read what P writes.
  std::vector<uint32_t> cand_ptr, cand;
  size_t total() const { size_t n = 0; for (auto& v : q) n += v.size(); return n; }
};
static void df_build(int nt, DfSchedule& s) {
  // key: (bucket, order inside the bucket ...)
  typedef std::tuple<int, int, int, int, int, int> Key;
  std::vector<std::pair<Key, DfTask>> qs[DF_NQ];
  auto task = [](int i, int j, int k0, int k1, int half, int fin) {
    DfTask t; memset(&t, 0, sizeof(t));
    t.i = (uint16_t)i; t.j = (uint16_t)j; t.k0 = (uint16_t)k0; t.k1 = (uint16_t)k1; t.half = (uint8_t)half; t.fin = (uint8_t)fin;
    return t;
  };
  // a task next to the front: one whole tile, or its two halves
  auto near_task = [&](int q, int bucket, int cls, int i, int j, int k0, int k1, int fin) {
    if (DF_HALVES) { for (int h = 0; h < 2; ++h) qs[q].push_back({Key(bucket, cls, i, j, h, 0), task(i, j, k0, k1, h, fin)}); }
    else qs[q].push_back({Key(bucket, cls, i, j, 0, 0), task(i, j, k0, k1, 2, fin)});
  };
  // queue 0 (crit), buckets of link l: 2 l = behind D > l, the multiplies by L_ll^-T of the near rows, then their k = l updates
  // that need only those; 2 l + 1 = the k = l updates of tile column l + 1, whose B operand L(l+1, l) is the diagonal worker's
  // (gate: its row counter) -- for the rows of the panel's own diagonal block ONLY.  Queue 1, the same two buckets per link:
  // the same tasks for the eight rows below (the next diagonal block), and step k = l of the tiles of the NEXT block column in
  // the sixteen rows that will be its near rows.  Those also wait for the lo queues, so in a queue of their own, or a bucket
  // of theirs that stands open would eat the crit queue's scan window (the form before this one: 25 of 37 ms at N = 16384 spent
  // by the diagonal worker waiting for multiplies that were runnable for 2.5 ms).  Queue 2 (hi): per link the far rows' tasks,
  // then the far pieces that end there.  Queues 3 and 4 (lo): one bucket each, always open, tickets in order: the passes of
  // panel q into the next two block columns that still lack it (q + 2, q + 3) before everything further to the right -- the
  // diagonal worker reaches those columns within two panels, the rest has time.  The tiles of the next diagonal blocks
  // (rows < 8 p + 16 of block column p) take their old passes through the hi queue instead: few tasks, and the lo queues run
  // 2-3 ms behind what is runnable when the chip is full -- exactly the tiles the diagonal worker then waits for.
  for (int j = 0; j < nt; ++j) {
    const int p = j / DF_PW, q0 = DF_PW * p;
    for (int i = j; i < nt; ++i) {
      const int kmax = std::max(0, i == j ? j - 1 : j);      // the workers' share of the tile's k range: [0, kmax)
      const bool near_row = i < DF_PW * (p + 1) + DF_NEARX;
      // (1) panels older than the previous one: one pass per panel (K = 1024), in the launch chain's order -- panel by panel,
      //     every tile to the right of the next block column -- so that a tile has all but the previous panel long before its
      //     own panel begins.  (The first forms were lazy: ONE group of passes per block column, K = 2048, while the panel before
      //     it was factored.  The diagonal block of column 12 at N = 16384 then got its six passes one after the other, 1.9 ms,
      //     23 ms after their inputs were final, with the diagonal worker waiting: profiles/r05/dataflow_critpath_session_r_N16384.txt.)
      if (p >= 2) {
        const int wend = std::min(DF_PW * (p - 1), kmax);
        for (int k0 = 0; k0 < wend; k0 += DF_PW)
          if (i < DF_PW * p + DF_NEARF)      // the next diagonal blocks' tiles: LISTED (queue 1) -- they run the moment their rows are final through the panel
            qs[1].push_back({Key(k0 + DF_PW - 1, 2, 0, i, j, 0), task(i, j, k0, std::min(k0 + DF_PW, wend), 2, 0)});
          else {
            // (inside a panel's group: the rows of the next few diagonal blocks first -- they are the first to become near rows)
            const int q_ = k0 / DF_PW, soon = i / DF_PW <= q_ + 4 ? 0 : 1;
            qs[p - q_ <= DF_LO_NEAR ? 3 : 4].push_back({Key(0, q_, soon, p, j, i), task(i, j, k0, std::min(k0 + DF_PW, wend), 2, 0)});
          }
      }
      // (2) the previous panel
      if (p >= 1) {
        const int a0 = DF_PW * (p - 1), a1 = std::min(DF_PW * p, kmax);
        if (i < DF_PW * p + DF_NEARF) {
          for (int k = a0; k < a1; ++k) near_task(1, 2 * k + (j == k + 1 ? 1 : 0), 1, i, j, k, k + 1, 0);
        } else {
          static const int cut[5] = {0, 4, 6, 7, 8};
          for (int c = 0; c < 4; ++c) {
            const int k0 = a0 + cut[c], k1 = std::min(a0 + cut[c + 1], a1);
            if (k1 > k0) qs[2].push_back({Key(k1 - 1, 2, i / DF_PW <= p + 2 ? 0 : 1, j, i, 0), task(i, j, k0, k1, 2, 0)});
          }
        }
      }
      // (3) inside the panel, eagerly for the rows near the front
      if (kmax > q0 && near_row)
        for (int k = q0; k < kmax; ++k) near_task(i < DF_PW * (p + 1) ? 0 : 1, 2 * k + (j == k + 1 ? 1 : 0), 1, i, j, k, k + 1, 0);
      // (4) the multiply by L_jj^-T (rows j + 2 and below; row j + 1 is the diagonal worker's)
      if (i >= j + 2) {
        if (near_row) near_task(i < DF_PW * (p + 1) ? 0 : 1, 2 * j, 0, i, j, j, j, 1);
        else {
          // far rows: the in-panel k range and the multiply as ONE task per HALF tile -- a row's eight tasks of a panel follow
          // one another (36 products), and as whole tiles (23-27 us per product beside a second workgroup on the CU) they took
          // longer than the diagonal worker needs for the panel: rows finished late, the lo queue behind them stood still
          for (int h = 0; h < 2; ++h) qs[2].push_back({Key(j, 0, i, j, h, 0), task(i, j, kmax > q0 ? q0 : j, j, h, 1)});
        }
      }
    }
  }
  for (int q = 0; q < DF_NQ; ++q) {
    std::stable_sort(qs[q].begin(), qs[q].end(), [](const std::pair<Key, DfTask>& a, const std::pair<Key, DfTask>& b) { return a.first < b.first; });
    s.q[q].clear(); s.b[q].clear();
    s.q[q].reserve(qs[q].size());
    int cur = -1;
    for (auto& e : qs[q]) {
      const int bk = std::get<0>(e.first);
      if (q < DF_NLIST) { s.q[q].push_back(e.second); continue; }
      if (bk != cur) {
        cur = bk;
        DfBucket B; B.start = (uint32_t)s.q[q].size(); B.size = 0;
        if (q >= 3) { B.gate_word = DF_D; B.gate_val = 0; }                                          // always open
        else if (q == 2) { B.gate_word = DF_D; B.gate_val = (uint32_t)bk + 1; }                      // D > link
        else if (bk % 2 == 0) { B.gate_word = DF_D; B.gate_val = (uint32_t)(bk / 2) + 1; }           // D > link
        else { B.gate_word = DF_ROWH + 2 * (uint32_t)(bk / 2 + 1); B.gate_val = (uint32_t)(bk / 2) + 1; }   // L(l+1, l) published
        s.b[q].push_back(B);
      }
      s.q[q].push_back(e.second);
      s.b[q].back().size += 1;
    }
  }
  // ---- candidates: every listed task X goes onto the list of each of its inputs' producers (df_ready()'s rule)
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
    for (int q = 0; q < DF_NLIST; ++q)
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
  // order: a producer keeps the FIRST runnable candidate of its list for itself (see df_list_candidates) -- the one that
  // continues its own row's chain: same row first, then the nearest column
  {
    std::vector<const DfTask*> flat;
    for (int q = 0; q < DF_NQ; ++q) for (const DfTask& t : s.q[q]) flat.push_back(&t);
    for (size_t P = 0; P < lists.size(); ++P) {
      const int prow = P < G ? (int)flat[P]->i : -1;
      std::stable_sort(lists[P].begin(), lists[P].end(), [&](uint32_t x, uint32_t y) {
        const DfTask &a = *flat[x], &b = *flat[y];
        const int ka = (int)a.i == prow ? 0 : 1, kb = (int)b.i == prow ? 0 : 1;
        return std::make_tuple(ka, (int)a.j, (int)a.i, (int)a.k0) < std::make_tuple(kb, (int)b.j, (int)b.i, (int)b.k0);
      });
    }
  }
  s.cand_ptr.assign(1, 0u);
  s.cand.clear();
  for (auto& l : lists) { s.cand.insert(s.cand.end(), l.begin(), l.end()); s.cand_ptr.push_back((uint32_t)s.cand.size()); }
}

// the schedule of an nt x nt tile matrix, for the replay in tests/test_dataflow_schedule.py (host only, no device needed):
// counts[q] = tasks of queue q (0 crit, 1 hi, 2 lo); out (when not NULL): rows of 10 ints {queue, i, j, k0, k1, half, fin,
// bucket, gate word, gate value} queue by queue, bucket by bucket, in ticket order, at most max_rows of them (gate word: 0 = D,
// 256 + 2 r + h = rowh[r, h])
extern "C" int gh_debug_dflow_schedule(int32_t nt, int32_t* counts, int32_t* out, int64_t max_rows) {
  if (nt <= 0 || nt > 4096 || !counts) { gh_set_error("dflow_schedule: bad argument"); return GH_ERR_BAD_ARG; }
  DfSchedule s;
  df_build(nt, s);
  int64_t r = 0;
  for (int q = 0; q < DF_NQ; ++q) {
    counts[q] = (int32_t)s.q[q].size();
    if (!out) continue;
    size_t bi = 0;
    for (size_t x = 0; x < s.q[q].size(); ++x) {
      const DfTask& t = s.q[q][x];
      while (bi + 1 < s.b[q].size() && x >= s.b[q][bi + 1].start) ++bi;
      if (r >= max_rows) return GH_OK;
      int32_t* o = out + 10 * r++;
      o[0] = q; o[1] = t.i; o[2] = t.j; o[3] = t.k0; o[4] = t.k1; o[5] = t.half; o[6] = t.fin;
      if (s.b[q].empty()) { o[7] = -1; o[8] = 0; o[9] = 0; }            // a listed queue: no buckets
      else { o[7] = (int32_t)bi; o[8] = (int32_t)s.b[q][bi].gate_word; o[9] = (int32_t)s.b[q][bi].gate_val; }
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
  const DfBucket* buckets[DF_NQ];
  unsigned nb[DF_NQ];            // buckets per queue
  unsigned off_next[DF_NQ];      // cnt + off_next[q] + b: the ticket counter of bucket b of queue q
  unsigned count[DF_NLIST];      // tasks of the listed queues
  unsigned off_note[DF_NLIST];   // cnt + off_note[q] + x: task x of listed queue q is on its ready list
  unsigned off_list[DF_NLIST];   // cnt + off_list[q] + e: entry e of ready list q = task index + 1 (0: not written yet)
  unsigned qoff[DF_NQ + 1];      // global index of the first task of queue q
  const uint32_t* cand_ptr;      // candidates of producer P: cand[cand_ptr[P] .. cand_ptr[P + 1])
  const uint32_t* cand;
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
__device__ __forceinline__ void df_publish(unsigned* w0, unsigned v0, unsigned* w1, unsigned v1) {
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

// After df_publish(): the calling workgroup's threads check the inputs of producer P's candidates (tasks of the listed queues
// that read what P wrote), one candidate per thread, and append the runnable ones to their ready list.  Two producers that
// finish a task's last two inputs at the same time both find it runnable (each has its own counters out -- waited for --
// before it reads the other's); the note's compare-and-swap lets one of them list it.
// `keep` (LDS, or NULL): the workgroup keeps the first runnable candidate of the list for itself instead of listing it -- its
// own row's next task, by the order the host gave the list; *keep = its global index + 1, or 0.
__device__ __forceinline__ void df_list_candidates(const DfArgs& a, unsigned P, unsigned* keep) {
  const unsigned beg = a.cand_ptr[P], end = a.cand_ptr[P + 1];
  if (keep && threadIdx.x == 0) *keep = 0xffffffffu;
  if (beg == end) { if (keep) { __syncthreads(); if (threadIdx.x == 0) *keep = 0u; __syncthreads(); } return; }     // (uniform)
  __syncthreads();
  unsigned* const cnt = a.cnt;
  for (unsigned c0 = beg; c0 < end; c0 += blockDim.x) {
    const unsigned c = c0 + threadIdx.x;
    bool mine = false;
    unsigned g = 0, x = 0;
    int q = 0;
    if (c < end) {
      g = a.cand[c];
      q = g >= a.qoff[1] ? 1 : 0;
      x = g - a.qoff[q];
      unsigned* const note = cnt + a.off_note[q] + x;
      if (df_ld(note) == 0u) {
        const DfTask t = a.tasks[q][x];
        if (df_ready(a, t)) {
          unsigned expect = 0u;
          mine = __hip_atomic_compare_exchange_strong(note, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (keep && c0 == beg) {                                     // (uniform) the first round decides what stays here
      if (mine) __hip_atomic_fetch_min(keep, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
      if (mine && *keep == c) {                                   // kept: not listed -- one list slot less will ever be written
        mine = false;
        (void)__hip_atomic_fetch_add(cnt + DF_TAIL + 4 + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (mine) {
      const unsigned slot = __hip_atomic_fetch_add(cnt + DF_TAIL + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      df_st(cnt + a.off_list[q] + slot, x + 1u);
    }
  }
  if (keep) {
    __syncthreads();
    if (threadIdx.x == 0) *keep = *keep == 0xffffffffu ? 0u : a.cand[*keep] + 1u;
    __syncthreads();
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
  // (through a scalar register: the compiler must SEE that every lane takes the same way out of the caller's loop, or it is
  //  free to split the lanes over two loops -- see the workers' loop)
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
        df_publish(cnt + DF_ROWH + 2 * j, (unsigned)j, cnt + DF_ROWH + 2 * j + 1, (unsigned)j);
        df_list_candidates(a, a.qoff[DF_NQ] + (unsigned)j, nullptr);
        if (TR) { df_trace(a, tt, j, j - 1, 0, 0, 9, 2, 1); tt = wall_clock64(); }
        // (no barrier: the other wavefronts request the first slab and C while wavefront 0 is in the release)
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
      df_publish(cnt + DF_D, (unsigned)(j + 1), nullptr, 0u);
      df_list_candidates(a, a.qoff[DF_NQ] + (unsigned)a.nt + (unsigned)j, nullptr);
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
  __shared__ unsigned s_keep;              // the candidate this workgroup kept for itself after its last task (global index + 1), or 0
  const int tid = threadIdx.x;
  unsigned* const cnt = a.cnt;
  const long ld = a.ld;
  const unsigned mykey = df_cu_key();
  if (tid == 0) s_keep = 0u;
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

  // (lane 0) per queue: a claimed task that waits for its inputs (index into the queue), or none
  unsigned held[DF_NQ], tick[DF_NLIST];
  bool lo_done[DF_NQ - DF_NPEEK];
#pragma unroll
  for (int q = 0; q < DF_NQ - DF_NPEEK; ++q) lo_done[q] = false;
#pragma unroll
  for (int q = 0; q < DF_NQ; ++q) held[q] = 0xffffffffu;
#pragma unroll
  for (int q = 0; q < DF_NLIST; ++q) tick[q] = 0xffffffffu;
  for (;;) {
    if (tid == 0) {
      int st = 0;
      const long long t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        if (df_ld(cnt + DF_ABORT) != 0u) { st = -1; break; }
        bool anyleft = false;
        if (s_keep != 0u) {                                     // the next task of the row this workgroup just worked on
          const unsigned g = s_keep - 1u;
          const int q = g >= a.qoff[1] ? 1 : 0;
          s_task = a.tasks[q][g - a.qoff[q]]; s_gid = g; s_keep = 0u; st = 1 + q;
          break;
        }
        // ---- the listed queues first.  Entries are taken by TICKET (a fetch-and-add on the list's head that always succeeds),
        // up to DF_AHEAD tickets ahead of what is listed: the owner of a ticket polls its own slot -- an address nobody else
        // looks at -- and starts the task the moment a producer writes it.  (Taking the first entry by compare-and-swap, every
        // idle workgroup that saw the tail move fired at the same word: hundreds of failed read-modify-writes per listed task on
        // one memory channel, and every load on the chip that touched that channel queued behind them -- all tasks 2-3x slower,
        // the more the more workgroups were idle: profiles/r05/dataflow_ab_session_n.log, _w.log.)
        {
          // (all the words of both lists in one batch: the loads are independent, the round trip is paid once)
          unsigned tl[DF_NLIST], hd[DF_NLIST], kp[DF_NLIST], sl[DF_NLIST];
#pragma unroll
          for (int q = 0; q < DF_NLIST; ++q) {
            tl[q] = df_ld(cnt + DF_TAIL + q); hd[q] = df_ld(cnt + DF_TAIL + 8 + q); kp[q] = df_ld(cnt + DF_TAIL + 4 + q);
            sl[q] = tick[q] != 0xffffffffu ? df_ld(cnt + a.off_list[q] + tick[q]) : 0u;
          }
#pragma unroll
          for (int q = 0; q < DF_NLIST; ++q) {
            if (st != 0) continue;
            const unsigned lim = a.count[q] - kp[q];                                  // slots that can still be written
            if (tick[q] == 0xffffffffu) {
              if (hd[q] < lim) anyleft = true;
              if (hd[q] < lim && hd[q] < tl[q] + DF_AHEAD) {
                const unsigned tk = __hip_atomic_fetch_add(cnt + DF_TAIL + 8 + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk < lim) { tick[q] = tk; sl[q] = df_ld(cnt + a.off_list[q] + tk); }
              }
            }
            if (tick[q] != 0xffffffffu) {
              const unsigned e = sl[q];
              if (e != 0u) { anyleft = true; s_task = a.tasks[q][e - 1u]; s_gid = a.qoff[q] + e - 1u; tick[q] = 0xffffffffu; st = 1 + q; }
              else if (tick[q] >= lim) tick[q] = 0xffffffffu;   // as many tasks were KEPT by their producers as slots lie behind this one: it stays empty
              else anyleft = true;
            }
          }
        }
        // (a workgroup that waits for a listed entry takes no new bucket tickets: the entry would start late behind a 200-us
        //  pass; what it holds from before it still runs when that becomes runnable -- others may be waiting for it)
        const bool waiting = tick[0] != 0xffffffffu || tick[1] != 0xffffffffu;
#pragma unroll
        for (int q = DF_NLIST; q < DF_NQ; ++q) {
          if (st != 0) continue;
          bool fresh = false;
          if (q >= DF_NPEEK) {
            // the lo queues: ONE bucket, always open, tickets in order -- no hint, no gate, no look: a fetch-and-add
            if (held[q] == 0xffffffffu && !waiting && !lo_done[q - DF_NPEEK]) {
              const unsigned size = a.qoff[q + 1] - a.qoff[q];
              const unsigned tk = size ? __hip_atomic_fetch_add(cnt + a.off_next[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : size;
              if (tk < size) { held[q] = tk; fresh = true; } else lo_done[q - DF_NPEEK] = true;
            }
            if (!lo_done[q - DF_NPEEK]) anyleft = true;
          } else
          if (held[q] == 0xffffffffu && !waiting) {
            // claim: from the first bucket that is not used up on, at most DF_SCAN of them, none behind a closed D gate (those
            // open in order).  crit and hi LOOK before they take a ticket: the next task of the bucket must be runnable -- a
            // ticket for a task whose inputs are far away would keep its owner from the lo queue (or, held, start late behind
            // its owner's lo task: the second form of this kernel lost 25 of 37 ms at N = 16384 that way)
            unsigned* const hint = cnt + DF_HEAD + 32 * q;
            unsigned bk = df_ld(hint);
            const unsigned bk0 = bk;
            bool front = true;                  // every bucket before bk is used up
            for (int scan = 0; bk < a.nb[q] && scan < DF_SCAN; ++bk) {
              const DfBucket B = a.buckets[q][bk];
              unsigned* const next = cnt + a.off_next[q] + bk;
              const unsigned nx = df_ld(next);
              if (nx >= B.size) { if (front) __hip_atomic_fetch_max(hint, bk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
              front = false;
              anyleft = true;
              ++scan;
              if (df_ld(cnt + B.gate_word) < B.gate_val) { if (B.gate_word == DF_D) break; continue; }
              if (q < DF_NPEEK && !df_ready(a, a.tasks[q][B.start + nx])) continue;
              const unsigned tk = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (tk < B.size) { held[q] = B.start + tk; fresh = true; break; }
            }
            (void)bk0;
          }
          if (held[q] != 0xffffffffu) {
            anyleft = true;
            const DfTask t = a.tasks[q][held[q]];
            bool ok = df_ready(a, t);
            if (!ok && fresh && q < DF_NPEEK) {
              // just claimed and not runnable: its inputs are tasks of the same link, in flight -- wait a little before
              // going on with other work (a task that is held while its owner runs a long lo task starts late)
              const long long w0 = wall_clock64();
              const long long lim = 5000;                                     // 50 us
              while (!ok && wall_clock64() - w0 < lim) {
                __builtin_amdgcn_s_sleep(4);
                if (df_ld(cnt + DF_ABORT) != 0u) break;
                ok = df_ready(a, t);
              }
            }
            if (ok) { s_task = t; s_gid = a.qoff[q] + held[q]; held[q] = 0xffffffffu; st = 1 + q; }
          }
        }
        if (st) break;
        if (!anyleft) { st = -1; break; }        // nothing left to claim, nothing held
        if (spins < 4) __builtin_amdgcn_s_sleep(16); else if (spins < 16) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(127);
        if ((++spins & 31u) == 0u && wall_clock64() - t0 > DF_TIMEOUT_TICKS) { df_st(cnt + DF_ABORT, 2u); st = -1; break; }
      }
      if (st > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_state = st;
    }
    __syncthreads();
    // The way out of the loop must be UNIFORM for the compiler too: s_state comes from LDS, a per-lane value as far as it knows,
    // and with a per-lane exit it may restructure this loop into two -- the lanes that never enter the block above circling
    // through the barrier and the task below on their own, lane 0 parked outside for ever.  It did, in the scanner form of this
    // kernel: its first task re-ran ~10^5 times per second, the result never published (profiles/r05/dataflow_lane_split_smoke.log).
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
        df_publish(cnt + DF_ROWH + 2 * t.i, (unsigned)t.j + 1u, cnt + DF_ROWH + 2 * t.i + 1, (unsigned)t.j + 1u);
      } else {
        gh_tile64_nt<false>(smem, C, ld, C, ld, a.dinv + (long)t.j * 128 * 128, 128, 128);
        df_publish(cnt + DF_ROWH + 2 * t.i + t.half, (unsigned)t.j + 1u, nullptr, 0u);
      }
    } else if (t.half == 2) {
      df_publish(kd, t.k1, kd + 1, t.k1);
    } else {
      df_publish(kd + t.half, t.k1, nullptr, 0u);
    }
    df_list_candidates(a, __builtin_amdgcn_readfirstlane(s_gid), &s_keep);
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
  DfBucket* b[DF_NQ] = {};
  unsigned count[DF_NQ] = {};
  unsigned nb[DF_NQ] = {};
  uint32_t* cand_ptr = nullptr;
  uint32_t* cand = nullptr;
};
static std::mutex g_df_mutex;
static std::map<std::pair<int, int>, DfDeviceSchedule> g_df_cache;     // (device, nt): never freed (a few MB per size)

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

size_t gh_dflow_counter_bytes(int64_t np) { return df_words((int)(np / 128)) * sizeof(unsigned); }

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
        n.nb[q] = (unsigned)s.b[q].size();
        if (s.q[q].empty()) continue;
        GH_HIP(hipMalloc((void**)&n.d[q], s.q[q].size() * sizeof(DfTask)));
        GH_HIP(hipMemcpy(n.d[q], s.q[q].data(), s.q[q].size() * sizeof(DfTask), hipMemcpyHostToDevice));
        if (s.b[q].empty()) continue;
        GH_HIP(hipMalloc((void**)&n.b[q], s.b[q].size() * sizeof(DfBucket)));
        GH_HIP(hipMemcpy(n.b[q], s.b[q].data(), s.b[q].size() * sizeof(DfBucket), hipMemcpyHostToDevice));
      }
      if (s.q[0].size() + s.q[1].size() > (size_t)DF_LISTED_PER_LINK * nt) { gh_set_error("dflow: more listed tasks than the counter buffer holds"); return GH_ERR_BAD_ARG; }
      GH_HIP(hipMalloc((void**)&n.cand_ptr, s.cand_ptr.size() * sizeof(uint32_t)));
      GH_HIP(hipMemcpy(n.cand_ptr, s.cand_ptr.data(), s.cand_ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      GH_HIP(hipMalloc((void**)&n.cand, std::max<size_t>(1, s.cand.size()) * sizeof(uint32_t)));
      if (!s.cand.empty()) GH_HIP(hipMemcpy(n.cand, s.cand.data(), s.cand.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
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
  unsigned at = (unsigned)df_off_next(nt);
  unsigned total = 0;
  for (int q = 0; q < DF_NQ; ++q) {
    a.tasks[q] = ds.d[q]; a.buckets[q] = ds.b[q]; a.nb[q] = ds.nb[q]; a.off_next[q] = at;
    at += ds.nb[q];
    a.qoff[q] = total; total += ds.count[q];
  }
  a.qoff[DF_NQ] = total;
  if ((size_t)at > df_off_notes(nt)) { gh_set_error("dflow: bucket counters overflow"); return GH_ERR_BAD_ARG; }
  at = (unsigned)df_off_notes(nt);
  for (int q = 0; q < DF_NLIST; ++q) {
    a.count[q] = ds.count[q];
    a.off_note[q] = at; at += ds.count[q];
    a.off_list[q] = at; at += ds.count[q];
  }
  a.cand_ptr = ds.cand_ptr; a.cand = ds.cand;
  a.off_kd = (unsigned)df_off_kd(nt); a.nt = nt;
  a.trace = nullptr; a.trace_cap = 0;
  if (g_df_trace_cap > 0) {
    if (!g_df_trace) GH_HIP(hipMalloc((void**)&g_df_trace, (1 + 4 * (size_t)g_df_trace_cap) * sizeof(unsigned long long)));
    GH_HIP(hipMemsetAsync(g_df_trace, 0, sizeof(unsigned long long), st));
    a.trace = g_df_trace; a.trace_cap = (unsigned)g_df_trace_cap;
  }
  GH_HIP(hipMemsetAsync(counters, 0, df_words(nt) * sizeof(unsigned), st));
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
