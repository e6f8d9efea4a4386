// gh_mgpu.hip -- the dense GP solve on SEVERAL MI355X behind the C ABI: one process, one host thread
// and two streams per device (panel chain + transfers / trailing updates: one-panel look-ahead), 2-D
// block-cyclic tiles, panels moved with RCCL over xGMI.
//
// Replaces, for a caller that binds include/george_amd.h directly, what BasicSolver.compute /
// dot_solve / apply_inverse do on one host (reference src/george/solvers/basic.py:51-102) on `n_dev`
// GPUs; the multi-PROCESS form of the same algorithm (one rank per GPU under torch.distributed, the
// form bench.py --gpus N launches) is george_amd/distributed.py.  Same partitioning in both: the
// padded matrix is cut into nb x nb tiles, tile (I, J) lives on rank (I mod Pr) * Pc + (J mod Pc),
// every rank BUILDS its own tiles from (kernel, x) on its own GPU, and the factorisation is
// right-looking, one tile column per step:
//
//   P(k)  the owner of the diagonal tile factors it (gh_dev_potrf_block) and sends L_kk and its
//         diagonal-block inverses down its process column; that column TRSMs its panel tiles;
//         the panel travels along every process row, then the tiles each process column needs
//         transposed travel inside that column;
//   U(k)  every rank updates its own trailing tiles (one fp64-MFMA GEMM per local tile column).
//
// Transport.  Every transfer is a broadcast from one rank to the 1-3 other members of its process row
// or column.  GH_MGPU_RCCL (default): grouped ncclSend / ncclRecv on the world communicator of
// ncclCommInitAll -- on the xGMI full mesh the root's g - 1 sends leave over g - 1 different links at
// once (a ring broadcast inside the group would put the whole panel on one link per hop); librccl is
// resolved with dlopen at the first gh_mgpu_create, so single-GPU users of the library do not need
// it (and a process that already holds torch's librccl shares that one).  GH_MGPU_COPY: peer copies
// (hipMemcpyPeerAsync behind events), which also accepts the SAME device several times -- "virtual
// devices": the ownership / ordering logic of an n_dev-rank run can then be exercised on one GPU
// (tests/test_gpu_mgpu.py), the thing RCCL refuses to do.
//
// The O(N^2) triangular sweeps (log-likelihood, K^-1 y) are latency-bound chains of nt small steps;
// their nb-long partial sums and solutions are exchanged through pinned host memory (the ranks are
// threads of one process), two thread barriers per tile row.
#include <dlfcn.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "gh_common.h"
#include "gh_threads.h"
#include <rccl/rccl.h>            // types and enums only: every function is resolved with dlsym

#define T GH_TILE
#define MG_MAX_DEV 16

// ------------------------------------------------------------------------------- RCCL, lazily bound
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  if (g_rccl.lib == nullptr) {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { g_rccl.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?"); return false; }
  }
#define MG_SYM(field, name)                                                            \
  *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, name);                                  \
  if (!g_rccl.field) { g_rccl.why = std::string("librccl.so lacks ") + name; return false; }
  MG_SYM(CommInitAll, "ncclCommInitAll") MG_SYM(CommDestroy, "ncclCommDestroy") MG_SYM(GroupStart, "ncclGroupStart")
  MG_SYM(GroupEnd, "ncclGroupEnd") MG_SYM(Send, "ncclSend") MG_SYM(Recv, "ncclRecv") MG_SYM(AllReduce, "ncclAllReduce")
  MG_SYM(GetErrorString, "ncclGetErrorString")
#undef MG_SYM
  g_rccl.ok = true;
  return true;
}

struct Group {                     // a process row, a process column, or the world
  std::vector<int> members;        // global ranks, ascending
  HostBarrier bar;
  // GH_MGPU_COPY: what the current root offers
  const void* src = nullptr;
  int src_dev = 0;
  hipEvent_t src_ready = nullptr;
};

struct MRank {
  int rank = 0, dev = 0, pr = 0, pc = 0;
  hipStream_t st = nullptr;        // trailing updates, build, sweeps
  hipStream_t sp = nullptr;        // (high priority) the panel chain: potrf, TRSM and every transfer
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;        // GH_MGPU_COPY hand-shake (recorded on sp)
  hipEvent_t ev_panel[2] = {nullptr, nullptr}, ev_bcol = nullptr;      // look-ahead: panel k is here / block column k+1 is up to date
  hipStream_t su[2] = {nullptr, nullptr};                              // two more streams for the per-tile-column GEMMs of U(k): the tail of
  hipEvent_t ev_fan = nullptr, ev_su[2] = {nullptr, nullptr};          // one launch overlaps the head of the next (fenced against st on both sides)
  gh_kernel kern;
  GhBuf A, dinv, Lkk, wrow[2], colp[2], x, yerr, scal, zloc, xloc, va, vb, part, flags;
  long long* d_info = nullptr;
  double* pin = nullptr;           // pinned host staging, 4 * nb doubles
  std::vector<int> rows, cols;     // global tile rows / columns this rank owns, ascending
  int rc = GH_OK;
  std::string err;
  double logdet = 0.0, acc = 0.0;
  long long info = 0;
};
}  // namespace

struct gh_mgpu {
  gh_mgpu_opts opts;
  int W = 1, Pr = 1, Pc = 1;
  int64_t n = 0, nb = 0, nt = 0, ndim = 0;
  std::vector<MRank> ranks;
  std::vector<Group> rowg, colg;   // rowg[r]: the Pc ranks of process row r; colg[c]: the Pr ranks of process column c
  Group world;
  ncclComm_t comms[MG_MAX_DEV];
  bool have_comms = false;
  std::atomic<int> abort{0};
  bool computed = false;
  double logdet = 0.0;
  int64_t info = 0;
  // host-side exchange of the triangular sweeps: parts[k][member] and the solved tiles, nb doubles each
  std::vector<double> parts, zfull, xfull;
  ~gh_mgpu() {
    for (auto& r : ranks) {
      (void)hipSetDevice(r.dev);
      if (r.st) (void)hipStreamSynchronize(r.st);
      if (r.sp) (void)hipStreamSynchronize(r.sp);
      for (GhBuf* b : {&r.A, &r.dinv, &r.Lkk, &r.wrow[0], &r.wrow[1], &r.colp[0], &r.colp[1], &r.x, &r.yerr, &r.scal, &r.zloc, &r.xloc, &r.va,
                       &r.vb, &r.part, &r.flags}) b->release();
      if (r.kern.d_nodes) { (void)hipFree(r.kern.d_nodes); r.kern.d_nodes = nullptr; }
      if (r.d_info) (void)hipFree(r.d_info);
      if (r.pin) (void)hipHostFree(r.pin);
      if (r.ev_ready) (void)hipEventDestroy(r.ev_ready);
      if (r.ev_done) (void)hipEventDestroy(r.ev_done);
      for (hipEvent_t e : {r.ev_panel[0], r.ev_panel[1], r.ev_bcol, r.ev_fan, r.ev_su[0], r.ev_su[1]}) if (e) (void)hipEventDestroy(e);
      for (hipStream_t q : {r.su[0], r.su[1]}) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
      if (r.sp) (void)hipStreamDestroy(r.sp);
      if (r.st) (void)hipStreamDestroy(r.st);
    }
    if (have_comms) for (int i = 0; i < W; ++i) (void)g_rccl.CommDestroy(comms[i]);
  }
};

#define MG_NCCL(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) {                                                                       \
      gh_set_error("RCCL error %d (%s) at %s:%d: %s", (int)r_, g_rccl.GetErrorString(r_), __FILE__, __LINE__, #expr); \
      return GH_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)

namespace {
inline int grank(const gh_mgpu* h, int pr, int pc) { return pr * h->Pc + pc; }
inline int local_index(const std::vector<int>& v, int g) {            // position of global tile g in v (must be there)
  return (int)(std::lower_bound(v.begin(), v.end(), g) - v.begin());
}
inline int first_at_least(const std::vector<int>& v, int g) { return (int)(std::lower_bound(v.begin(), v.end(), g) - v.begin()); }

// ---- broadcast of `count` doubles at `buf` (same address role on every member) from global rank `root`
int mg_bcast(gh_mgpu* h, MRank& r, Group& g, double* buf, size_t count, int root, hipStream_t st) {
  if (g.members.size() <= 1 || count == 0) return GH_OK;
  if (h->opts.transport == GH_MGPU_RCCL) {
    MG_NCCL(g_rccl.GroupStart());
    if (r.rank == root) {
      for (int p : g.members) if (p != root) MG_NCCL(g_rccl.Send(buf, count, ncclDouble, p, h->comms[r.rank], st));
    } else {
      MG_NCCL(g_rccl.Recv(buf, count, ncclDouble, root, h->comms[r.rank], st));
    }
    MG_NCCL(g_rccl.GroupEnd());
    return GH_OK;
  }
  // peer copies: root publishes (pointer, "data ready" event); members pull; root waits for their "done" events
  if (r.rank == root) {
    GH_HIP(hipEventRecord(r.ev_ready, st));
    g.src = buf; g.src_dev = r.dev; g.src_ready = r.ev_ready;
  }
  if (!g.bar.wait()) { gh_set_error("multi-GPU solve aborted (another rank failed)"); return GH_ERR_HIP; }
  if (r.rank != root) {
    GH_HIP(hipStreamWaitEvent(st, g.src_ready, 0));
    if (g.src_dev == r.dev) GH_HIP(hipMemcpyAsync(buf, g.src, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    else GH_HIP(hipMemcpyPeerAsync(buf, r.dev, g.src, g.src_dev, count * sizeof(double), st));
    GH_HIP(hipEventRecord(r.ev_done, st));
  }
  if (!g.bar.wait()) { gh_set_error("multi-GPU solve aborted (another rank failed)"); return GH_ERR_HIP; }
  if (r.rank == root)
    for (int p : g.members) if (p != root) GH_HIP(hipStreamWaitEvent(st, h->ranks[p].ev_done, 0));
  if (!g.bar.wait()) { gh_set_error("multi-GPU solve aborted (another rank failed)"); return GH_ERR_HIP; }   // (events and g.src are reused by the next call)
  return GH_OK;
}
inline int mg_group_start(gh_mgpu* h) { if (h->opts.transport == GH_MGPU_RCCL) MG_NCCL(g_rccl.GroupStart()); return GH_OK; }
inline int mg_group_end(gh_mgpu* h) { if (h->opts.transport == GH_MGPU_RCCL) MG_NCCL(g_rccl.GroupEnd()); return GH_OK; }

__global__ void mg_copy2d_kernel(const double* in, long ldi, double* out, long ldo, long rows, long cols) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const long i = idx / cols, j = idx % cols;
  out[i * ldo + j] = in[i * ldi + j];
}
int mg_copy2d(hipStream_t st, const double* in, long ldi, double* out, long ldo, long rows, long cols) {
  if (rows <= 0 || cols <= 0) return GH_OK;
  const long tot = rows * cols;
  hipLaunchKernelGGL(mg_copy2d_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, ldi, out, ldo, rows, cols);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// run fn(rank) on one thread per rank; collect the first error into the caller's thread
template <typename F>
int mg_run(gh_mgpu* h, F fn) {
  h->abort.store(0);
  // (a run that was aborted -- a matrix that is not positive definite, say -- leaves the arrival count of the barrier it
  //  died in behind: the test of the NOT_PD path re-uses its handle)
  h->world.bar.reset();
  for (auto& g : h->rowg) g.bar.reset();
  for (auto& g : h->colg) g.bar.reset();
  std::vector<std::thread> th;
  for (int i = 0; i < h->W; ++i) {
    th.emplace_back([h, i, &fn]() {
      MRank& r = h->ranks[i];
      r.rc = GH_OK; r.err.clear();
      if (hipSetDevice(r.dev) != hipSuccess) { r.rc = GH_ERR_HIP; r.err = "hipSetDevice failed"; h->abort.store(1); return; }
      const int rc = fn(r);
      if (rc != GH_OK) { r.rc = rc; r.err = gh_last_error(); h->abort.store(1); }
    });
  }
  for (auto& t : th) t.join();
  int first = GH_OK;
  for (auto& r : h->ranks) {
    if (r.rc == GH_OK) continue;
    // (a rank that merely saw the abort flag reports a generic message: prefer a real one)
    if (first == GH_OK || (r.err.find("aborted") == std::string::npos)) {
      first = r.rc;
      gh_set_error("rank %d (device %d): %s", r.rank, r.dev, r.err.c_str());
      if (r.err.find("aborted") == std::string::npos) break;
    }
  }
  return first;
}

// ------------------------------------------------------------------------------------ one rank's work
int rank_setup(gh_mgpu* h, MRank& r, const gh_kernel* k, const double* x, const double* yerr) {
  const int64_t nb = h->nb, nt = h->nt, n = h->n;
  r.rows.clear(); r.cols.clear();
  for (int i = 0; i < nt; ++i) { if (i % h->Pr == r.pr) r.rows.push_back(i); if (i % h->Pc == r.pc) r.cols.push_back(i); }
  const size_t nlr = std::max<size_t>(r.rows.size(), 1), nlc = std::max<size_t>(r.cols.size(), 1);
  // a private copy of the kernel program on this device (a gh_kernel caches ONE device copy)
  if (r.kern.d_nodes) { (void)hipFree(r.kern.d_nodes); r.kern.d_nodes = nullptr; }
  r.kern.nodes = k->nodes; r.kern.ndim = k->ndim; r.kern.size = k->size; r.kern.fast = k->fast; r.kern.device = -1;
  GH_CHECK(r.kern.upload());
  GH_CHECK(r.A.ensure(nlr * nb * nlc * nb * sizeof(double)));
  GH_CHECK(r.dinv.ensure((size_t)nt * (nb / T) * T * T * sizeof(double)));
  GH_CHECK(r.Lkk.ensure((size_t)nb * nb * sizeof(double)));
  for (int q = 0; q < 2; ++q) {                              // panel workspaces, double-buffered for the look-ahead
    GH_CHECK(r.wrow[q].ensure(nlr * nb * nb * sizeof(double)));
    GH_CHECK(r.colp[q].ensure(nlc * nb * nb * sizeof(double)));
  }
  GH_CHECK(r.x.ensure((size_t)n * h->ndim * sizeof(double)));
  GH_CHECK(r.yerr.ensure((size_t)n * sizeof(double)));
  GH_CHECK(r.scal.ensure(64 * sizeof(double)));
  GH_CHECK(r.zloc.ensure(nlc * nb * sizeof(double)));
  GH_CHECK(r.xloc.ensure(nlr * nb * sizeof(double)));
  GH_CHECK(r.va.ensure((size_t)nb * sizeof(double)));
  GH_CHECK(r.vb.ensure((size_t)nb * sizeof(double)));
  GH_CHECK(r.part.ensure((size_t)nb * sizeof(double)));
  GH_CHECK(r.flags.ensure((size_t)(nb / T + 2) * sizeof(unsigned)));
  if (!r.d_info) GH_HIP(hipMalloc((void**)&r.d_info, sizeof(long long)));
  if (r.pin) { (void)hipHostFree(r.pin); r.pin = nullptr; }
  GH_HIP(hipHostMalloc((void**)&r.pin, 4 * (size_t)nb * sizeof(double), hipHostMallocDefault));
  GH_CHECK(gh_to_device(r.x.d(), x, (size_t)n * h->ndim, r.st));
  GH_CHECK(gh_to_device(r.yerr.d(), yerr, (size_t)n, r.st));
  GH_HIP(hipMemsetAsync(r.d_info, 0, sizeof(long long), r.st));
  GH_HIP(hipMemsetAsync(r.scal.p, 0, 64 * sizeof(double), r.st));
  return GH_OK;
}

// Right-looking factorisation with one-panel look-ahead on two streams per rank:
//   sp (high priority): P(k) -- potrf, column TRSM and EVERY transfer of panel k;
//   st                : U(k) -- first the tiles of block column k+1 (event ev_bcol: P(k+1) may start),
//                       then the rest of the trailing update, which runs while P(k+1) is factored and travels.
// Panel workspaces alternate by the parity of k: P(k+2) overwrites what U(k) read, and it is issued
// only after the block-column part of U(k+1), which follows U(k) on st.
int rank_factor(gh_mgpu* h, MRank& r) {
  const int64_t nb = h->nb, nt = h->nt;
  const int Pr = h->Pr, Pc = h->Pc;
  const long ld = (long)std::max<size_t>(r.cols.size(), 1) * nb;
  const int nlr = (int)r.rows.size();
  double* A = r.A.d();
  auto tile = [&](int i, int j) { return A + (long)local_index(r.rows, i) * nb * ld + (long)local_index(r.cols, j) * nb; };
  // ---- build: every rank evaluates its own tiles (lower tile triangle only)
  for (int i : r.rows)
    for (int j : r.cols)
      if (j <= i)
        GH_CHECK(gh_dev_kmat_block(&r.kern, r.x.d(), h->n, (int32_t)h->ndim, r.yerr.d(), (int64_t)i * nb, nb, (int64_t)j * nb, nb,
                                   tile(i, j), ld, r.st));
  // P(k) on stream sp into workspace `buf`
  auto panel = [&](int k, int buf) -> int {
    hipStream_t sp = r.sp;
    const int kr = k % Pr, kc = k % Pc;
    const bool in_col = (r.pc == kc);
    double* dk = r.dinv.d() + (long)k * (nb / T) * T * T;
    double* wrow = r.wrow[buf].d();
    double* colp = r.colp[buf].d();
    if (r.pr == kr && in_col) {
      double* akk = tile(k, k);
      GH_CHECK(gh_dev_potrf_block(akk, ld, nb, dk, (int64_t*)r.d_info, (int64_t)k * nb, sp));
      GH_CHECK(gh_dev_logdet_accum(akk, ld, nb, r.scal.d(), sp));
      GH_CHECK(mg_copy2d(sp, akk, ld, r.Lkk.d(), nb, nb, nb));
    }
    if (k == nt - 1) return GH_OK;
    if (in_col && Pr > 1) {
      GH_CHECK(mg_group_start(h));
      GH_CHECK(mg_bcast(h, r, h->colg[kc], r.Lkk.d(), (size_t)nb * nb, grank(h, kr, kc), sp));
      GH_CHECK(mg_bcast(h, r, h->colg[kc], dk, (size_t)(nb / T) * T * T, grank(h, kr, kc), sp));
      GH_CHECK(mg_group_end(h));
    }
    const int li0 = first_at_least(r.rows, k + 1);
    const long m = (long)(nlr - li0) * nb;
    if (in_col && m > 0) {
      double* pan = A + (long)li0 * nb * ld + (long)local_index(r.cols, k) * nb;
      GH_CHECK(gh_dev_trsm_right(r.Lkk.d(), nb, dk, pan, ld, m, nb, sp));
      GH_CHECK(mg_copy2d(sp, pan, ld, wrow, nb, m, nb));
    }
    if (Pc > 1 && m > 0) GH_CHECK(mg_bcast(h, r, h->rowg[r.pr], wrow, (size_t)m * nb, grank(h, r.pr, kc), sp));
    // column panel: tile row j of the panel, for my tile columns j > k; held (after the row transfer) by process row j % Pr
    const int lc0 = first_at_least(r.cols, k + 1);
    GH_CHECK(mg_group_start(h));
    for (size_t lj = lc0; lj < r.cols.size(); ++lj) {
      const int j = r.cols[lj], src_pr = j % Pr;
      double* pj = colp + (long)lj * nb * nb;
      if (r.pr == src_pr)
        GH_HIP(hipMemcpyAsync(pj, wrow + (long)(local_index(r.rows, j) - li0) * nb * nb, (size_t)nb * nb * sizeof(double),
                              hipMemcpyDeviceToDevice, sp));
      if (Pr > 1) GH_CHECK(mg_bcast(h, r, h->colg[r.pc], pj, (size_t)nb * nb, grank(h, src_pr, r.pc), sp));
    }
    GH_CHECK(mg_group_end(h));
    return GH_OK;
  };
  // U(k) restricted to my tile columns with global index in [jlo, jhi], on stream st, from workspace `buf`
  auto update = [&](int k, int buf, int jlo, int jhi) -> int {
    const int li0 = first_at_least(r.rows, k + 1);
    const size_t l0 = first_at_least(r.cols, jlo);
    size_t l1 = l0;
    while (l1 < r.cols.size() && r.cols[l1] <= jhi) ++l1;
    const bool fan = l1 - l0 >= 3;                               // independent GEMMs dealt over st, su[0], su[1]
    if (fan) {
      GH_HIP(hipEventRecord(r.ev_fan, r.st));
      for (hipStream_t q : r.su) GH_HIP(hipStreamWaitEvent(q, r.ev_fan, 0));
    }
    for (size_t lj = l0; lj < l1; ++lj) {
      const int j = r.cols[lj];
      const int ls = first_at_least(r.rows, j);
      if (ls >= nlr) continue;
      const size_t q = fan ? (lj - l0) % 3 : 0;
      GH_CHECK(gh_dev_gemm_nt(A + (long)ls * nb * ld + (long)lj * nb, ld, r.wrow[buf].d() + (long)(ls - li0) * nb * nb, nb,
                              r.colp[buf].d() + (long)lj * nb * nb, nb, (long)(nlr - ls) * nb, nb, nb, 0, q == 0 ? r.st : r.su[q - 1]));
    }
    if (fan)
      for (int q = 0; q < 2; ++q) { GH_HIP(hipEventRecord(r.ev_su[q], r.su[q])); GH_HIP(hipStreamWaitEvent(r.st, r.ev_su[q], 0)); }
    return GH_OK;
  };
  // ---- factor
  GH_HIP(hipEventRecord(r.ev_bcol, r.st));                       // (the build is complete)
  GH_HIP(hipStreamWaitEvent(r.sp, r.ev_bcol, 0));
  GH_CHECK(panel(0, 0));
  GH_HIP(hipEventRecord(r.ev_panel[0], r.sp));
  for (int k = 0; k < nt; ++k) {
    const int buf = k & 1;
    GH_HIP(hipStreamWaitEvent(r.st, r.ev_panel[buf], 0));        // panel k is factored and here
    if (k == nt - 1) break;
    GH_CHECK(update(k, buf, k + 1, k + 1));                      // block column k+1 first
    GH_HIP(hipEventRecord(r.ev_bcol, r.st));
    GH_HIP(hipStreamWaitEvent(r.sp, r.ev_bcol, 0));
    GH_CHECK(panel(k + 1, buf ^ 1));                             // P(k+1) beside the rest of U(k)
    GH_HIP(hipEventRecord(r.ev_panel[buf ^ 1], r.sp));
    GH_CHECK(update(k, buf, k + 2, (int)nt - 1));
  }
  long long info = 0;
  double ld_part = 0.0;
  GH_HIP(hipMemcpyAsync(&info, r.d_info, sizeof(long long), hipMemcpyDeviceToHost, r.st));
  GH_HIP(hipMemcpyAsync(&ld_part, r.scal.d(), sizeof(double), hipMemcpyDeviceToHost, r.st));
  GH_HIP(hipStreamSynchronize(r.st));
  GH_HIP(hipStreamSynchronize(r.sp));
  r.info = info; r.logdet = ld_part;
  return GH_OK;
}

// z = L^-1 y (into h->zfull) and, when `backward`, x = L^-T z (into h->xfull); r.acc += |z_k|^2 on the owners
int rank_sweeps(gh_mgpu* h, MRank& r, const double* ypad, bool backward) {
  const int64_t nb = h->nb, nt = h->nt;
  const int Pr = h->Pr, Pc = h->Pc;
  const long ld = (long)std::max<size_t>(r.cols.size(), 1) * nb;
  const int nlr = (int)r.rows.size();
  double* A = r.A.d();
  const size_t pstride = (size_t)std::max(Pr, Pc) * nb;              // parts[k][member][nb]
  r.acc = 0.0;
  int* failflag = (int*)((unsigned*)r.flags.p + nb / T);
  auto aborted = [&]() { gh_set_error("multi-GPU solve aborted (another rank failed)"); return GH_ERR_HIP; };
  for (int k = 0; k < nt; ++k) {                                  // ---- forward, left-looking over tile rows
    const int kr = k % Pr, kc = k % Pc;
    double* zk = h->zfull.data() + (size_t)k * nb;
    if (r.pr == kr) {
      const int cend = first_at_least(r.cols, k);                  // my tile columns j < k
      double* mine = h->parts.data() + (size_t)k * pstride + (size_t)r.pc * nb;
      if (cend > 0) {
        GH_CHECK(gh_dev_gemv(A + (long)local_index(r.rows, k) * nb * ld, ld, nb, (int64_t)cend * nb, 0, r.zloc.d(), r.part.d(), 1.0, 0.0, r.st));
        GH_HIP(hipMemcpyAsync(r.pin, r.part.d(), (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipStreamSynchronize(r.st));
        memcpy(mine, r.pin, (size_t)nb * sizeof(double));
      } else {
        memset(mine, 0, (size_t)nb * sizeof(double));
      }
      if (!h->rowg[kr].bar.wait()) return aborted();
      if (r.pc == kc) {                                            // the diagonal owner: w = y_k - sum of the partials, z_k = L_kk^-1 w
        double* w = r.pin + nb;
        for (int64_t t = 0; t < nb; ++t) {
          double s = 0.0;
          for (int c = 0; c < Pc; ++c) s += h->parts[(size_t)k * pstride + (size_t)c * nb + t];     // fixed order: reproducible
          w[t] = ypad[(size_t)k * nb + t] - s;
        }
        GH_HIP(hipMemcpyAsync(r.va.d(), w, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, r.st));
        const long lk = local_index(r.rows, k), lck = local_index(r.cols, k);
        GH_CHECK(gh_dev_trsv_lower(A + lk * nb * ld + lck * nb, ld, r.dinv.d() + (long)k * (nb / T) * T * T, nb, r.va.d(), r.vb.d(), r.flags.p, r.st));
        int failed = 0;
        GH_HIP(hipMemcpyAsync(r.pin + 2 * nb, r.vb.d(), (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipMemcpyAsync(&failed, failflag, sizeof(int), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipStreamSynchronize(r.st));
        if (failed) { gh_set_error("forward solve: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
        memcpy(zk, r.pin + 2 * nb, (size_t)nb * sizeof(double));
        for (int64_t t = 0; t < nb; ++t) r.acc += zk[t] * zk[t];
      }
    }
    if (r.pc == kc) {                                              // process column kc keeps z_k for the tile rows to come
      if (!h->colg[kc].bar.wait()) return aborted();
      GH_HIP(hipMemcpyAsync(r.zloc.d() + (long)local_index(r.cols, k) * nb, zk, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, r.st));
      GH_HIP(hipStreamSynchronize(r.st));
    }
  }
  if (!backward) return GH_OK;
  if (!h->world.bar.wait()) return aborted();                      // (parts[] is reused)
  for (int k = (int)nt - 1; k >= 0; --k) {                         // ---- backward: x_k = L_kk^-T (z_k - sum_{i>k} L_ik^T x_i)
    const int kr = k % Pr, kc = k % Pc;
    double* xk = h->xfull.data() + (size_t)k * nb;
    if (r.pc == kc) {
      const int li0 = first_at_least(r.rows, k + 1);               // my tile rows i > k
      double* mine = h->parts.data() + (size_t)k * pstride + (size_t)r.pr * nb;
      if (li0 < nlr) {
        GH_CHECK(gh_dev_gemv(A + (long)li0 * nb * ld + (long)local_index(r.cols, k) * nb, ld, (int64_t)(nlr - li0) * nb, nb, 1,
                             r.xloc.d() + (long)li0 * nb, r.part.d(), 1.0, 0.0, r.st));
        GH_HIP(hipMemcpyAsync(r.pin, r.part.d(), (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipStreamSynchronize(r.st));
        memcpy(mine, r.pin, (size_t)nb * sizeof(double));
      } else {
        memset(mine, 0, (size_t)nb * sizeof(double));
      }
      if (!h->colg[kc].bar.wait()) return aborted();
      if (r.pr == kr) {
        double* w = r.pin + nb;
        for (int64_t t = 0; t < nb; ++t) {
          double s = 0.0;
          for (int p = 0; p < Pr; ++p) s += h->parts[(size_t)k * pstride + (size_t)p * nb + t];
          w[t] = h->zfull[(size_t)k * nb + t] - s;
        }
        GH_HIP(hipMemcpyAsync(r.va.d(), w, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, r.st));
        const long lk = local_index(r.rows, k), lck = local_index(r.cols, k);
        GH_CHECK(gh_dev_trsv_lower_t(A + lk * nb * ld + lck * nb, ld, r.dinv.d() + (long)k * (nb / T) * T * T, nb, r.va.d(), r.vb.d(), r.flags.p, r.st));
        int failed = 0;
        GH_HIP(hipMemcpyAsync(r.pin + 2 * nb, r.vb.d(), (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipMemcpyAsync(&failed, failflag, sizeof(int), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipStreamSynchronize(r.st));
        if (failed) { gh_set_error("backward solve: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
        memcpy(xk, r.pin + 2 * nb, (size_t)nb * sizeof(double));
      }
    }
    if (r.pr == kr) {                                              // process row kr keeps x_k for the tile columns to come
      if (!h->rowg[kr].bar.wait()) return aborted();
      GH_HIP(hipMemcpyAsync(r.xloc.d() + (long)local_index(r.rows, k) * nb, xk, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, r.st));
      GH_HIP(hipStreamSynchronize(r.st));
    }
  }
  return GH_OK;
}
}  // namespace

// =============================================================================================== ABI
extern "C" int gh_mgpu_create(const gh_mgpu_opts* opts, gh_mgpu** out) {
  if (!opts || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  if (opts->n_dev < 1 || opts->n_dev > MG_MAX_DEV) { gh_set_error("n_dev must be 1..%d", MG_MAX_DEV); return GH_ERR_BAD_ARG; }
  if (opts->nb < 0 || opts->nb % T) { gh_set_error("nb must be a multiple of 128"); return GH_ERR_BAD_ARG; }
  if (opts->transport != GH_MGPU_RCCL && opts->transport != GH_MGPU_COPY) { gh_set_error("unknown transport"); return GH_ERR_BAD_ARG; }
  const int ndev_box = gh_device_count();
  if (ndev_box <= 0) { gh_set_error("no HIP device available: the george_amd solver needs an MI355X"); return GH_ERR_HIP; }
  const int W = opts->n_dev;
  for (int i = 0; i < W; ++i) {
    if (opts->devices[i] < 0 || opts->devices[i] >= ndev_box) { gh_set_error("devices[%d] = %d: this box has %d", i, opts->devices[i], ndev_box); return GH_ERR_BAD_ARG; }
    if (opts->transport == GH_MGPU_RCCL)
      for (int j = 0; j < i; ++j)
        if (opts->devices[j] == opts->devices[i]) { gh_set_error("device %d listed twice: RCCL needs one rank per device (GH_MGPU_COPY accepts virtual devices)", opts->devices[i]); return GH_ERR_BAD_ARG; }
  }
  int Pr = opts->pr, Pc = opts->pc;
  if (Pr <= 0 || Pc <= 0) {                                   // as square as n_dev allows, Pr <= Pc (1x2, 2x2, 2x4)
    Pr = 1;
    for (int p = 1; p * p <= W; ++p) if (W % p == 0) Pr = p;
    Pc = W / Pr;
  }
  if (Pr * Pc != W) { gh_set_error("grid %d x %d does not hold %d devices", Pr, Pc, W); return GH_ERR_BAD_ARG; }
  gh_mgpu* h = new gh_mgpu();
  h->opts = *opts; h->W = W; h->Pr = Pr; h->Pc = Pc;
  h->ranks.resize(W);
  h->rowg = std::vector<Group>(Pr); h->colg = std::vector<Group>(Pc);
  for (int i = 0; i < W; ++i) {
    MRank& r = h->ranks[i];
    r.rank = i; r.dev = opts->devices[i]; r.pr = i / Pc; r.pc = i % Pc;
    h->rowg[r.pr].members.push_back(i); h->colg[r.pc].members.push_back(i); h->world.members.push_back(i);
    int plo = 0, phi = 0;                                       // numerically lowest value = highest priority
    if (hipSetDevice(r.dev) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
    gh_prime_device(r.dev);                  // (gh_common.h: the null stream must have seen a launch before the first stream is made)
    if (hipSetDevice(r.dev) != hipSuccess || hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&r.sp, hipStreamNonBlocking, phi) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_panel[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_panel[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_bcol, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&r.su[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&r.su[1], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_fan, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_su[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_su[1], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); delete h; gh_set_error("stream / event creation failed on device %d", opts->devices[i]); return GH_ERR_HIP;
    }
  }
  for (auto* gs : {&h->rowg, &h->colg}) for (auto& g : *gs) { g.bar.n = (int)g.members.size(); g.bar.abort = &h->abort; }
  h->world.bar.n = W; h->world.bar.abort = &h->abort;
  if (opts->transport == GH_MGPU_RCCL) {
    if (!rccl_load()) { delete h; gh_set_error("RCCL unavailable: %s", g_rccl.why.c_str()); return GH_ERR_HIP; }
    ncclResult_t rc = g_rccl.CommInitAll(h->comms, W, opts->devices);
    if (rc != ncclSuccess) { delete h; gh_set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(rc)); return GH_ERR_HIP; }
    h->have_comms = true;
    // self-check of the communicator: all-reduce of (rank + 1) must give W (W + 1) / 2 on every rank
    std::vector<double> got(W, 0.0);
    int rcs = mg_run(h, [&](MRank& r) -> int {
      double v = (double)(r.rank + 1);
      GH_CHECK(r.scal.ensure(64 * sizeof(double)));
      GH_HIP(hipMemcpyAsync(r.scal.p, &v, sizeof(double), hipMemcpyHostToDevice, r.st));
      MG_NCCL(g_rccl.AllReduce(r.scal.p, (double*)r.scal.p + 1, 1, ncclDouble, ncclSum, h->comms[r.rank], r.st));
      GH_HIP(hipMemcpyAsync(&got[r.rank], (double*)r.scal.p + 1, sizeof(double), hipMemcpyDeviceToHost, r.st));
      GH_HIP(hipStreamSynchronize(r.st));
      return GH_OK;
    });
    if (rcs != GH_OK) { delete h; return rcs; }
    for (int i = 0; i < W; ++i)
      if (got[i] != 0.5 * W * (W + 1)) { delete h; gh_set_error("RCCL self-check failed on rank %d: all-reduce gave %g, expected %g", i, got[i], 0.5 * W * (W + 1)); return GH_ERR_HIP; }
  }
  *out = h;
  return GH_OK;
}

extern "C" void gh_mgpu_destroy(gh_mgpu* h) { delete h; }
extern "C" int64_t gh_mgpu_info(const gh_mgpu* h) { return h ? h->info : 0; }
extern "C" int gh_mgpu_grid(const gh_mgpu* h, int32_t* pr, int32_t* pc, int32_t* nb) {
  if (!h) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (pr) *pr = h->Pr;
  if (pc) *pc = h->Pc;
  if (nb) *nb = (int32_t)h->nb;
  return GH_OK;
}

extern "C" int gh_mgpu_compute(gh_mgpu* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim, const double* yerr,
                               double* logdet_out) {
  if (!h || !k || !x || !yerr || n <= 0) { gh_set_error("bad argument to compute"); return GH_ERR_BAD_ARG; }
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  h->computed = false; h->info = 0;
  h->n = n; h->ndim = ndim;
  h->nb = h->opts.nb > 0 ? h->opts.nb : (n >= 24576 ? 1024 : 512);
  h->nt = (n + h->nb - 1) / h->nb;
  const size_t pstride = (size_t)std::max(h->Pr, h->Pc) * h->nb;
  h->parts.assign((size_t)h->nt * pstride, 0.0);
  h->zfull.assign((size_t)h->nt * h->nb, 0.0);
  h->xfull.assign((size_t)h->nt * h->nb, 0.0);
  int rc = mg_run(h, [&](MRank& r) -> int {
    GH_CHECK(rank_setup(h, r, k, x, yerr));
    return rank_factor(h, r);
  });
  if (rc != GH_OK) return rc;
  double tot = 0.0;
  long long bad = 0;
  for (auto& r : h->ranks) {                                   // fixed rank order: reproducible
    tot += r.logdet;
    if (r.info > 0 && (bad == 0 || r.info < bad)) bad = r.info;
  }
  if (bad != 0) {
    h->info = bad;
    gh_set_error("%lld-th leading minor of the array is not positive definite", bad);
    return GH_ERR_NOT_PD;
  }
  h->logdet = tot;
  h->computed = true;
  if (logdet_out) *logdet_out = tot;
  return GH_OK;
}

static int mg_need(gh_mgpu* h) {
  if (!h) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (!h->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  return GH_OK;
}

extern "C" int gh_mgpu_dot_solve(gh_mgpu* h, const double* y, double* out) {
  GH_CHECK(mg_need(h));
  if (!y || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  std::vector<double> ypad((size_t)h->nt * h->nb, 0.0);
  memcpy(ypad.data(), y, (size_t)h->n * sizeof(double));
  GH_CHECK(mg_run(h, [&](MRank& r) -> int { return rank_sweeps(h, r, ypad.data(), false); }));
  double acc = 0.0;
  for (auto& r : h->ranks) acc += r.acc;
  *out = acc;
  return GH_OK;
}

extern "C" int gh_mgpu_solve(gh_mgpu* h, const double* b, int64_t nrhs, double* out) {
  GH_CHECK(mg_need(h));
  if (nrhs < 0 || (nrhs > 0 && (!b || !out))) { gh_set_error("bad argument to solve"); return GH_ERR_BAD_ARG; }
  std::vector<double> ypad((size_t)h->nt * h->nb, 0.0);
  for (int64_t c = 0; c < nrhs; ++c) {                        // (n, nrhs) row-major in and out; one right-hand side per sweep pair
    for (int64_t i = 0; i < h->n; ++i) ypad[(size_t)i] = b[i * nrhs + c];
    GH_CHECK(mg_run(h, [&](MRank& r) -> int { return rank_sweeps(h, r, ypad.data(), true); }));
    for (int64_t i = 0; i < h->n; ++i) out[i * nrhs + c] = h->xfull[(size_t)i];
  }
  return GH_OK;
}
