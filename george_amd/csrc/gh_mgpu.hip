// gh_mgpu.hip -- the dense GP solve on SEVERAL MI355X behind the C ABI: one process, one host thread and
// three streams per device, block-cyclic tiles, panels moved with RCCL over xGMI.
//
// Replaces, for a caller that binds include/george_amd.h directly, the whole BasicSolver protocol
// (reference src/george/solvers/basic.py:51-121: compute, log-determinant, apply_inverse, dot_solve,
// apply_sqrt, get_inverse) and the predictive mean / variance / covariance of GP.predict
// (src/george/gp.py:482-545) on `n_dev` GPUs; the multi-PROCESS form of the same algorithm (one rank per
// GPU under torch.distributed, the form bench.py --gpus N launches) is george_amd/distributed.py.
//
// Partitioning.  The padded matrix is cut into nb x nb tiles; tile (I, J) lives on the rank at
// (prow(I), J mod Pc) of a Pr x Pc grid, and every rank BUILDS its own tiles from (kernel, x).
// DEFAULT GRID: Pr = n_dev, Pc = 1 -- whole tile rows per rank, dealt in "snake" order
// (prow(I) = I mod 2Pr folded back: 0 1 .. Pr-1 Pr-1 .. 1 0) so that every rank holds the same number
// of lower-triangle tiles.  Why not the square-ish grid of a switched network: on the xGMI full mesh
// every pair of GPUs has its own link, so what a step costs is the CHAIN
//    potrf(k) -> L_kk to the others -> TRSM of panel k -> what block column k+1 needs -> potrf(k+1),
// and with whole tile rows per rank the only things that travel on that chain are two nb x nb tiles
// (L_kk + its diagonal-block inverses, and panel tile k+1), sent to all peers over 7 different links
// at once; the TRSM and the block-column update are split over ALL ranks.  A 2 x 4 grid puts
// (N - k nb) / 2 x nb doubles of row panel on ONE link inside the chain at every step (8.6 GB per
// process row at N = 65536: 140-170 ms of the 234 ms a 6x speed-up allows; profiles/r04/scale_model.md).
// pr / pc in gh_mgpu_opts still select any grid (2-D block-cyclic, plain cyclic rows).
//
// Schedule (right-looking, one tile column per step, one-panel look-ahead), three streams per rank:
//   sp (high priority) the chain: potrf, L_kk broadcast, TRSM, row panel (Pc > 1), panel tile k+1 sent ahead;
//   sg                 the bulk: the rest of the column panel gathered (all links, its own communicator);
//   st                 the updates: block column k+1 first (then P(k+1) may start), then everything else as one staircase launch.
//
// Transport.  Every transfer is a broadcast from one rank to the other members of its process row or
// column.  GH_MGPU_RCCL (default): grouped ncclSend / ncclRecv on a world communicator of
// ncclCommInitAll -- on the full mesh the root's g - 1 sends leave over g - 1 different links at once;
// two communicators (chain, bulk) so that a gather in flight never delays the next chain transfer;
// librccl is resolved with dlopen at the first gh_mgpu_create (a process that already holds torch's
// librccl shares that one).  GH_MGPU_COPY: peer copies behind events, which also accepts the SAME
// device several times -- "virtual devices": ownership, ordering and hand-over logic of an n_dev-rank
// run on one GPU (tests/test_gpu_mgpu.py), the thing RCCL refuses to do.
//
// Solves are device-resident left-looking tile sweeps on the sharded factor for any number of
// right-hand sides (GEMMs on the matrix pipe; one right-hand side: gemv + the chained 128-block
// substitution kernel of gh_chol.hip); what crosses between ranks is one nb x R tile per tile row.
#include <dlfcn.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "gh_common.h"
#include "gh_threads.h"
#include "../../include/george_amd_debug.h"
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
  bool shared_device_ok = false;   // the loaded library is the test stand-in: ranks may share a device
  std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  if (g_rccl.lib == nullptr) {
    // GEORGE_AMD_RCCL_LIB=<path>: that library instead (tests/mock_rccl: a stand-in that checks every send against its
    // receive on virtual ranks; a library that exports ncclMockSharedDeviceOk lifts "one rank per device")
    if (const char* env = getenv("GEORGE_AMD_RCCL_LIB")) {
      g_rccl.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
      if (!g_rccl.lib) { g_rccl.why = std::string("GEORGE_AMD_RCCL_LIB: ") + (dlerror() ? dlerror() : "?"); return false; }
      g_rccl.shared_device_ok = dlsym(g_rccl.lib, "ncclMockSharedDeviceOk") != nullptr;
    }
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (g_rccl.lib) break;
      g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
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
  // GH_MGPU_COPY: what the current root offers (bcast) / what every member offers (reduce)
  const void* src[MG_MAX_DEV] = {nullptr};
  int src_dev[MG_MAX_DEV] = {0};
  hipEvent_t src_ready[MG_MAX_DEV] = {nullptr};
  int slot(int rank) const { return (int)(std::lower_bound(members.begin(), members.end(), rank) - members.begin()); }
};

struct TraceEv { int step, phase; hipEvent_t a, b; double units; };

struct MRank {
  int rank = 0, dev = 0, pr = 0, pc = 0;
  hipStream_t st = nullptr;        // updates, build, sweeps
  hipStream_t sp = nullptr;        // (high priority) the chain: potrf, TRSM, the small transfers
  hipStream_t sg = nullptr;        // the bulk gather of the column panel
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;        // GH_MGPU_COPY hand-shake
  hipEvent_t ev_fast[2] = {nullptr, nullptr};              // panel k: row panel + tile k+1 are here (block column k+1 may be updated)
  hipEvent_t ev_panel[2] = {nullptr, nullptr};             // panel k: the whole column panel is here
  hipEvent_t ev_bcol = nullptr, ev_rest = nullptr;
  gh_kernel kern;
  GhBuf A, dinv, Lkk, wrow[2], colp[2], nxt[2], x, yerr, scal, flags;
  GhBuf zc, xr, wk, part, red, rhs, acc;                   // sweeps: Z for my tile columns, X for my tile rows, work tile, partial, reduce slots
  long long* d_info = nullptr;
  double* h_back = nullptr;        // pinned, 2 doubles: [0] the failure word's bits, [1] this rank's log-det part (end of a factorisation)
  std::vector<int> rows, cols;     // global tile rows / columns this rank owns, ascending
  int rc = GH_OK;
  std::string err;
  double logdet = 0.0, acc_host = 0.0;
  long long info = 0;
  std::vector<TraceEv> trace;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  hipEvent_t next_ev() {
    if (ev_used == ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev_pool.push_back(e); }
    return ev_pool[ev_used++];
  }
};
}  // namespace

enum { MG_PH_POTRF = 0, MG_PH_TRSM = 1, MG_PH_BCOL = 2, MG_PH_REST = 3, MG_PH_LKK = 4, MG_PH_ROWX = 5, MG_PH_AHEAD = 6, MG_PH_GATHER = 7, MG_PH_BUILD = 8 };

struct gh_mgpu {
  gh_mgpu_opts opts;
  int W = 1, Pr = 1, Pc = 1;
  bool snake = false, chain_only = false, trace_on = false;
  int64_t n = 0, nb = 0, nt = 0, ndim = 0;
  std::vector<MRank> ranks;
  std::vector<Group> rowg, colg;   // rowg[r]: the Pc ranks of process row r; colg[c]: the Pr ranks of process column c
  Group world;
  ncclComm_t comms[MG_MAX_DEV], comms_b[MG_MAX_DEV];       // chain / bulk communicators
  bool have_comms = false, have_comms_b = false;
  bool one_comm = false;           // the bulk gather shares the chain's stream and communicator (gh_mgpu_create: why)
  std::atomic<int> abort{0};
  std::atomic<int> dead{0};        // mg_drain gave up on a stream: kernels may still run, nothing is synchronised or freed any more
  std::mutex turn;                 // trace mode: one rank's compute phase at a time (contention-free durations on virtual devices)
  bool computed = false;
  double logdet = 0.0;
  int64_t info = 0;
  std::vector<double> trace_rows;  // (rank, step, phase, ms, units) of the last traced compute()
  int prow(int i) const { if (!snake) return i % Pr; const int t = i % (2 * Pr); return t < Pr ? t : 2 * Pr - 1 - t; }
  int pcol(int j) const { return j % Pc; }
  ~gh_mgpu() {
    if (dead.load()) return;       // (see mg_drain)
    for (auto& r : ranks) {
      (void)hipSetDevice(r.dev);
      for (hipStream_t q : {r.st, r.sp, r.sg}) if (q) (void)hipStreamSynchronize(q);
      for (GhBuf* b : {&r.A, &r.dinv, &r.Lkk, &r.wrow[0], &r.wrow[1], &r.colp[0], &r.colp[1], &r.nxt[0], &r.nxt[1], &r.x, &r.yerr, &r.scal,
                       &r.flags, &r.zc, &r.xr, &r.wk, &r.part, &r.red, &r.rhs, &r.acc}) b->release();
      if (r.kern.d_nodes) { (void)hipFree(r.kern.d_nodes); r.kern.d_nodes = nullptr; }
      if (r.d_info) (void)hipFree(r.d_info);
      if (r.h_back) (void)hipHostFree(r.h_back);
      for (hipEvent_t e : {r.ev_ready, r.ev_done, r.ev_fast[0], r.ev_fast[1], r.ev_panel[0], r.ev_panel[1], r.ev_bcol, r.ev_rest}) if (e) (void)hipEventDestroy(e);
      for (hipEvent_t e : r.ev_pool) (void)hipEventDestroy(e);
      for (hipStream_t q : {r.sg, r.sp, r.st}) if (q) (void)hipStreamDestroy(q);
    }
    if (have_comms) for (int i = 0; i < W; ++i) (void)g_rccl.CommDestroy(comms[i]);
    if (have_comms_b) for (int i = 0; i < W; ++i) (void)g_rccl.CommDestroy(comms_b[i]);
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
inline int mg_aborted() { gh_set_error("multi-GPU solve aborted (another rank failed)"); return GH_ERR_HIP; }

// ---- the end of a factorisation: wait for a stream, but not for ever.  With RCCL the transfer kernels of two ranks wait
//      for each other ON THE DEVICE; should they ever do so crosswise (gh_mgpu_create: two communicators in flight) no HIP
//      call returns an error -- the streams just never drain.  GEORGE_AMD_MGPU_TIMEOUT_S (default 900; 0: wait for ever)
//      bounds the wait; past it the solver is marked dead (its device memory is abandoned: nothing can be freed under
//      kernels that still run) and the caller gets an error that names the way out.
int mg_drain(gh_mgpu* h, MRank& r, hipStream_t q) {
  static const double limit = [] { const char* e = getenv("GEORGE_AMD_MGPU_TIMEOUT_S"); return e ? atof(e) : 900.0; }();
  if (limit <= 0.0 || h->opts.transport != GH_MGPU_RCCL || h->W == 1) { GH_HIP(hipStreamSynchronize(q)); return GH_OK; }
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(q);
    if (e == hipSuccess) return GH_OK;
    if (e != hipErrorNotReady) { GH_HIP(e); }
    if (h->dead.load()) break;
    if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spin > 20000 ? 1000 : 50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
  }
  (void)hipGetLastError();
  h->dead.store(1);
  gh_set_error("multi-GPU factorisation: a stream of rank %d (device %d) did not drain within %g s%s; this solver is dead -- make a new one "
               "(GH_MGPU_ONE_COMM / one_comm=True keeps one communicator in flight at a time)", r.rank, r.dev, limit,
               h->one_comm ? "" : " with the chain and the bulk gather on two communicators");
  return GH_ERR_HIP;
}

// ---- broadcast of `count` doubles at `buf` (same address role on every member) from global rank `root`;
//      `cs`: the communicator set (chain or bulk) -- calls on one set are issued in the same order by every member
int mg_bcast(gh_mgpu* h, MRank& r, Group& g, double* buf, size_t count, int root, hipStream_t st, ncclComm_t* cs) {
  if (g.members.size() <= 1 || count == 0) return GH_OK;
  if (h->opts.transport == GH_MGPU_RCCL) {
    MG_NCCL(g_rccl.GroupStart());
    if (r.rank == root) {
      for (int p : g.members) if (p != root) MG_NCCL(g_rccl.Send(buf, count, ncclDouble, p, cs[r.rank], st));
    } else {
      MG_NCCL(g_rccl.Recv(buf, count, ncclDouble, root, cs[r.rank], st));
    }
    MG_NCCL(g_rccl.GroupEnd());
    return GH_OK;
  }
  // peer copies: root publishes (pointer, "data ready" event); members pull; root waits for their "done" events
  if (r.rank == root) {
    GH_HIP(hipEventRecord(r.ev_ready, st));
    g.src[0] = buf; g.src_dev[0] = r.dev; g.src_ready[0] = r.ev_ready;
  }
  if (!g.bar.wait()) return mg_aborted();
  if (r.rank != root) {
    GH_HIP(hipStreamWaitEvent(st, g.src_ready[0], 0));
    if (g.src_dev[0] == r.dev) GH_HIP(hipMemcpyAsync(buf, g.src[0], count * sizeof(double), hipMemcpyDeviceToDevice, st));
    else GH_HIP(hipMemcpyPeerAsync(buf, r.dev, g.src[0], g.src_dev[0], count * sizeof(double), st));
    GH_HIP(hipEventRecord(r.ev_done, st));
  }
  if (!g.bar.wait()) return mg_aborted();
  if (r.rank == root)
    for (int p : g.members) if (p != root) GH_HIP(hipStreamWaitEvent(st, h->ranks[p].ev_done, 0));
  if (!g.bar.wait()) return mg_aborted();                              // (events and g.src are reused by the next call)
  return GH_OK;
}
// ---- the members' `count` doubles at `part` collected on `root`: slot s of `red` = member s's (the root's own included)
int mg_collect(gh_mgpu* h, MRank& r, Group& g, const double* part, size_t count, int root, double* red, hipStream_t st, ncclComm_t* cs) {
  if (count == 0) return GH_OK;
  const int me = g.slot(r.rank);
  if (g.members.size() <= 1) {
    GH_HIP(hipMemcpyAsync(red, part, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    return GH_OK;
  }
  if (h->opts.transport == GH_MGPU_RCCL) {
    if (r.rank == root) GH_HIP(hipMemcpyAsync(red + (size_t)me * count, part, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    MG_NCCL(g_rccl.GroupStart());
    if (r.rank == root) {
      for (size_t s = 0; s < g.members.size(); ++s)
        if (g.members[s] != root) MG_NCCL(g_rccl.Recv(red + s * count, count, ncclDouble, g.members[s], cs[r.rank], st));
    } else {
      MG_NCCL(g_rccl.Send(part, count, ncclDouble, root, cs[r.rank], st));
    }
    MG_NCCL(g_rccl.GroupEnd());
    return GH_OK;
  }
  GH_HIP(hipEventRecord(r.ev_ready, st));
  g.src[me] = part; g.src_dev[me] = r.dev; g.src_ready[me] = r.ev_ready;
  if (!g.bar.wait()) return mg_aborted();
  if (r.rank == root) {
    for (size_t s = 0; s < g.members.size(); ++s) {
      if (g.members[s] != root) GH_HIP(hipStreamWaitEvent(st, g.src_ready[s], 0));
      if (g.src_dev[s] == r.dev) GH_HIP(hipMemcpyAsync(red + s * count, g.src[s], count * sizeof(double), hipMemcpyDeviceToDevice, st));
      else GH_HIP(hipMemcpyPeerAsync(red + s * count, r.dev, g.src[s], g.src_dev[s], count * sizeof(double), st));
    }
    GH_HIP(hipEventRecord(r.ev_done, st));
  }
  if (!g.bar.wait()) return mg_aborted();
  if (r.rank != root) GH_HIP(hipStreamWaitEvent(st, h->ranks[root].ev_done, 0));       // `part` may be overwritten after this
  if (!g.bar.wait()) return mg_aborted();
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
// w[i] = b[i] - sum_s red[s * count + i], slots added in index order (fixed order: reproducible); b may be NULL (0)
__global__ void mg_sub_sum_kernel(const double* b, const double* red, int nslots, long count, double* w) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int q = 0; q < nslots; ++q) s += red[(long)q * count + i];
  w[i] = (b ? b[i] : 0.0) - s;
}
int mg_sub_sum(hipStream_t st, const double* b, const double* red, int nslots, long count, double* w) {
  hipLaunchKernelGGL(mg_sub_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, b, red, nslots, count, w);
  GH_HIP(hipGetLastError());
  return GH_OK;
}
// acc[c] += sum_i v[i * ld + c]^2  and  mu[c] += sum_i v[i * ld + c] * z[i]   (one lane per column; rows in order: reproducible)
__global__ void mg_colacc_kernel(const double* v, long ld, long rows, long cols, const double* z, double* mu, double* sq) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double a = 0.0, b = 0.0;
  for (long i = 0; i < rows; ++i) { const double t = v[i * ld + c]; a += t * z[i]; b += t * t; }
  mu[c] += a; sq[c] += b;
}
__global__ void mg_sumsq_kernel(const double* v, long count, double* out) {      // out[0] += sum v^2, one workgroup, fixed order
  __shared__ double sh[256];
  double a = 0.0;
  for (long i = threadIdx.x; i < count; i += 256) a += v[i] * v[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] += sh[0];
}
__global__ void mg_sticky_flag_kernel(const unsigned* flag, unsigned* sticky) { if (threadIdx.x == 0 && *flag) *sticky = 1u; }
__global__ void mg_eye_tile_kernel(double* w, long ld, long rows, long cols, long grow0, long gcol0) {   // w = I[grow0.., gcol0..]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const long i = idx / cols, j = idx % cols;
  w[i * ld + j] = (grow0 + i == gcol0 + j) ? 1.0 : 0.0;
}

// run fn(rank) on one thread per rank; collect the first error into the caller's thread
// (threads are created per call: ~0.1 ms per rank and call, against sweeps of milliseconds)
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
  for (int i = 0; i < nt; ++i) { if (h->prow(i) == r.pr) r.rows.push_back(i); if (h->pcol(i) == r.pc) r.cols.push_back(i); }
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
    GH_CHECK(r.nxt[q].ensure((size_t)nb * nb * sizeof(double)));
  }
  GH_CHECK(r.x.ensure((size_t)n * h->ndim * sizeof(double)));
  GH_CHECK(r.yerr.ensure((size_t)n * sizeof(double)));
  GH_CHECK(r.scal.ensure(64 * sizeof(double)));
  GH_CHECK(r.flags.ensure((size_t)(nb / T + 2) * sizeof(unsigned)));
  if (!r.d_info) GH_HIP(hipMalloc((void**)&r.d_info, sizeof(long long)));
  if (!r.h_back) GH_HIP(hipHostMalloc((void**)&r.h_back, 2 * sizeof(double), hipHostMallocDefault));
  GH_CHECK(gh_to_device(r.x.d(), x, (size_t)n * h->ndim, r.st));
  GH_CHECK(gh_to_device(r.yerr.d(), yerr, (size_t)n, r.st));
  GH_HIP(hipMemsetAsync(r.d_info, 0, sizeof(long long), r.st));
  GH_HIP(hipMemsetAsync(r.scal.p, 0, 64 * sizeof(double), r.st));
  return GH_OK;
}

// Right-looking factorisation with one-panel look-ahead on three streams per rank (see the head of the file).
// Panel workspaces alternate by the parity of k: P(k+2) overwrites what U(k) read, and it is issued only after the
// block-column part of U(k+1), which follows U(k) on st.
int rank_factor(gh_mgpu* h, MRank& r) {
  const int64_t nb = h->nb, nt = h->nt;
  const int Pr = h->Pr, Pc = h->Pc;
  const long ld = (long)std::max<size_t>(r.cols.size(), 1) * nb;
  const int nlr = (int)r.rows.size();
  double* A = r.A.d();
  const bool tracing = h->trace_on;
  r.trace.clear(); r.ev_used = 0;
  auto tile = [&](int i, int j) { return A + (long)local_index(r.rows, i) * nb * ld + (long)local_index(r.cols, j) * nb; };
  // a phase of the schedule: `body` enqueues it on `s`.  Trace mode brackets it with timing events, and COMPUTE phases
  // (potrf, TRSM, the updates) additionally run one at a time over all ranks and to completion (h->turn): with every
  // rank on the same physical device ("virtual devices") the durations are then those of a rank alone on its GPU.
  auto phase = [&](int step, int ph, hipStream_t s, double units, auto&& body) -> int {
    if (!tracing) return body();
    const bool compute = (ph <= MG_PH_REST || ph == MG_PH_BUILD) && h->opts.transport == GH_MGPU_COPY;   // (RCCL: a rank inside the turnstile may wait for a send its peer has not issued)
    std::unique_lock<std::mutex> lk(h->turn, std::defer_lock);
    if (compute) lk.lock();
    hipEvent_t a = r.next_ev(), b = r.next_ev();
    if (!a || !b) { gh_set_error("event creation failed"); return GH_ERR_HIP; }
    GH_HIP(hipEventRecord(a, s));
    GH_CHECK(body());
    GH_HIP(hipEventRecord(b, s));
    r.trace.push_back(TraceEv{step, ph, a, b, units});
    if (compute) GH_HIP(hipStreamSynchronize(s));
    return GH_OK;
  };
  // ---- build: every rank evaluates its own tiles (lower tile triangle only)
  {
    double bytes = 0.0;
    for (int i : r.rows) for (int j : r.cols) if (j <= i) bytes += 8.0 * (double)nb * nb;
    GH_CHECK(phase(-1, MG_PH_BUILD, r.st, bytes, [&]() -> int {
      for (int i : r.rows)
        for (int j : r.cols)
          if (j <= i)
            GH_CHECK(gh_dev_kmat_block(&r.kern, r.x.d(), h->n, (int32_t)h->ndim, r.yerr.d(), (int64_t)i * nb, nb, (int64_t)j * nb, nb,
                                       tile(i, j), ld, r.st));
      return GH_OK;
    }));
  }
  // P(k), chain part, on stream sp into workspace `buf`: ends with "row panel + panel tile k+1 are here" (ev_fast[buf])
  auto panel = [&](int k, int buf) -> int {
    hipStream_t sp = r.sp;
    if (h->abort.load()) return mg_aborted();
    const int kr = h->prow(k), kc = h->pcol(k);
    const bool in_col = (r.pc == kc);
    double* dk = r.dinv.d() + (long)k * (nb / T) * T * T;
    double* wrow = r.wrow[buf].d();
    if (r.pr == kr && in_col) {
      double* akk = tile(k, k);
      GH_CHECK(phase(k, MG_PH_POTRF, sp, (double)nb * nb * nb / 3.0, [&]() -> int {
        GH_CHECK(gh_dev_potrf_block(akk, ld, nb, dk, (int64_t*)r.d_info, (int64_t)k * nb, sp));
        GH_CHECK(gh_dev_logdet_accum(akk, ld, nb, r.scal.d(), sp));
        return mg_copy2d(sp, akk, ld, r.Lkk.d(), nb, nb, nb);
      }));
    }
    if (k == nt - 1) { GH_HIP(hipEventRecord(r.ev_fast[buf], sp)); return GH_OK; }
    if (in_col && Pr > 1) {
      GH_CHECK(phase(k, MG_PH_LKK, sp, 8.0 * ((double)nb * nb + (double)(nb / T) * T * T), [&]() -> int {
        GH_CHECK(mg_group_start(h));
        GH_CHECK(mg_bcast(h, r, h->colg[kc], r.Lkk.d(), (size_t)nb * nb, grank(h, kr, kc), sp, h->comms));
        GH_CHECK(mg_bcast(h, r, h->colg[kc], dk, (size_t)(nb / T) * T * T, grank(h, kr, kc), sp, h->comms));
        return mg_group_end(h);
      }));
    }
    const int li0 = first_at_least(r.rows, k + 1);
    const long m = (long)(nlr - li0) * nb;
    if (in_col && m > 0) {
      double* pan = A + (long)li0 * nb * ld + (long)local_index(r.cols, k) * nb;
      GH_CHECK(phase(k, MG_PH_TRSM, sp, (double)m * nb * nb, [&]() -> int {
        GH_CHECK(gh_dev_trsm_right(r.Lkk.d(), nb, dk, pan, ld, m, nb, sp));
        return mg_copy2d(sp, pan, ld, wrow, nb, m, nb);
      }));
    }
    if (Pc > 1 && m > 0)
      GH_CHECK(phase(k, MG_PH_ROWX, sp, 8.0 * (double)m * nb, [&]() -> int {
        return mg_bcast(h, r, h->rowg[r.pr], wrow, (size_t)m * nb, grank(h, r.pr, kc), sp, h->comms); }));
    // panel tile k+1 (tile row k+1 of the panel), sent ahead to the process column that owns block column k+1:
    // all the next panel waits for is that block column, and its update needs this ONE tile of the column panel
    const int nc = h->pcol(k + 1), src_pr = h->prow(k + 1);
    if (r.pc == nc) {
      GH_CHECK(phase(k, MG_PH_AHEAD, sp, 8.0 * (double)nb * nb, [&]() -> int {
        if (r.pr == src_pr)
          GH_HIP(hipMemcpyAsync(r.nxt[buf].d(), wrow + (long)(local_index(r.rows, k + 1) - li0) * nb * nb, (size_t)nb * nb * sizeof(double),
                                hipMemcpyDeviceToDevice, sp));
        if (Pr > 1) GH_CHECK(mg_bcast(h, r, h->colg[nc], r.nxt[buf].d(), (size_t)nb * nb, grank(h, src_pr, nc), sp, h->comms));
        return GH_OK;
      }));
    }
    GH_HIP(hipEventRecord(r.ev_fast[buf], sp));
    return GH_OK;
  };
  // P(k), bulk part, on stream sg: tile row j of the panel for my tile columns j > k+1, held (after the row transfer)
  // by process row prow(j); ends with "the whole column panel is here" (ev_panel[buf])
  auto gather = [&](int k, int buf) -> int {
    // (one_comm: the bulk gather goes out on the CHAIN's stream and communicator, after the chain's transfers of the step, in
    //  program order -- the same on every rank; see gh_mgpu_create)
    hipStream_t sg = h->one_comm ? r.sp : r.sg;
    ncclComm_t* const bulk_comms = h->one_comm ? h->comms : h->comms_b;
    GH_HIP(hipStreamWaitEvent(sg, r.ev_fast[buf], 0));
    if (k < nt - 1 && !h->chain_only) {
      const int li0 = first_at_least(r.rows, k + 1);
      double* wrow = r.wrow[buf].d();
      double* colp = r.colp[buf].d();
      const int lc0 = first_at_least(r.cols, k + 2);
      double bytes = 0.0;
      for (size_t lj = lc0; lj < r.cols.size(); ++lj) bytes += 8.0 * (double)nb * nb;
      if (bytes > 0.0)
        GH_CHECK(phase(k, MG_PH_GATHER, sg, bytes, [&]() -> int {
          GH_CHECK(mg_group_start(h));
          for (size_t lj = lc0; lj < r.cols.size(); ++lj) {
            const int j = r.cols[lj], spr = h->prow(j);
            double* pj = colp + (long)lj * nb * nb;
            if (r.pr == spr)
              GH_HIP(hipMemcpyAsync(pj, wrow + (long)(local_index(r.rows, j) - li0) * nb * nb, (size_t)nb * nb * sizeof(double),
                                    hipMemcpyDeviceToDevice, sg));
            if (Pr > 1) GH_CHECK(mg_bcast(h, r, h->colg[r.pc], pj, (size_t)nb * nb, grank(h, spr, r.pc), sg, bulk_comms));
          }
          return mg_group_end(h);
        }));
    }
    GH_HIP(hipEventRecord(r.ev_panel[buf], sg));
    return GH_OK;
  };
  // U(k) restricted to my tile columns with global index in [jlo, jhi], on stream st, from workspace `buf`.
  // ahead: [jlo, jhi] is block column k+1 alone, served by the tile that travelled ahead -- one GEMM over all my rows below it.
  // Else ONE STAIRCASE GEMM over all my tile rows: C[i, jlo..min(i, jhi)] -= W_i P[jlo..]^T for every local row i -- my rows are
  // contiguous in A and in wrow, the column-panel tiles of consecutive local columns are contiguous in colp, and each row reaches
  // as far as its own diagonal tile (gh_dev_gemm_nt_stair).  (Round 4, first form: one GEMM per tile COLUMN -- 39 TFLOP/s per rank
  // at nb = 512; second form: one GEMM per tile row dealt over three streams -- 62 TFLOP/s: 8 launches of ~1.3 chip-fulls each per
  // step at N = 65536 on 8 ranks; profiles/r04/scale_model.md.)
  auto update = [&](int k, int buf, int jlo, int jhi, bool ahead) -> int {
    const int li0 = first_at_least(r.rows, k + 1);
    const size_t l0 = first_at_least(r.cols, jlo);
    double flops = 0.0;
    for (size_t lj = l0; lj < r.cols.size() && r.cols[lj] <= jhi; ++lj) {
      const int ls = first_at_least(r.rows, r.cols[lj]);
      if (ls < nlr) flops += 2.0 * (double)(nlr - ls) * nb * nb * nb;
    }
    if (flops == 0.0) return GH_OK;
    return phase(k, ahead ? MG_PH_BCOL : MG_PH_REST, r.st, flops, [&]() -> int {
      if (ahead) {
        const int ls = first_at_least(r.rows, jlo);
        return gh_dev_gemm_nt(A + (long)ls * nb * ld + (long)l0 * nb, ld, r.wrow[buf].d() + (long)(ls - li0) * nb * nb, nb, r.nxt[buf].d(), nb,
                              (long)(nlr - ls) * nb, nb, nb, 0, r.st);
      }
      // every tile row of mine that reaches column jlo, as ONE staircase launch: row li takes my columns jlo .. min(i, jhi)
      const int lr0 = first_at_least(r.rows, jlo);
      std::vector<int64_t> width;
      int first = -1;
      for (int li = lr0; li < nlr; ++li) {
        const int i = r.rows[li];
        size_t l1 = l0;
        while (l1 < r.cols.size() && r.cols[l1] <= std::min(i, jhi)) ++l1;
        if (l1 == l0) continue;                                     // (widths grow with li: empty rows come first)
        if (first < 0) first = li;
        width.push_back((int64_t)(l1 - l0) * nb);
      }
      if (first < 0) return GH_OK;
      GH_CHECK(gh_dev_gemm_nt_stair(A + (long)first * nb * ld + (long)l0 * nb, ld, r.wrow[buf].d() + (long)(first - li0) * nb * nb, nb,
                                    r.colp[buf].d() + (long)l0 * nb * nb, nb, nb, (int32_t)width.size(), width.data(), nb, r.st));
      return GH_OK;
    });
  };
  // ---- factor
  GH_HIP(hipEventRecord(r.ev_bcol, r.st));                       // (the build is complete)
  GH_HIP(hipStreamWaitEvent(r.sp, r.ev_bcol, 0));
  GH_CHECK(panel(0, 0));
  GH_CHECK(gather(0, 0));
  for (int k = 0; k < nt; ++k) {
    const int buf = k & 1;
    GH_HIP(hipStreamWaitEvent(r.st, r.ev_fast[buf], 0));         // row panel + panel tile k+1 are here
    if (k == nt - 1) break;
    GH_CHECK(update(k, buf, k + 1, k + 1, true));                // block column k+1 first
    GH_HIP(hipEventRecord(r.ev_bcol, r.st));
    GH_HIP(hipStreamWaitEvent(r.sp, r.ev_bcol, 0));
    GH_CHECK(panel(k + 1, buf ^ 1));                             // the chain of P(k+1) beside the rest of U(k)
    GH_CHECK(gather(k + 1, buf ^ 1));
    GH_HIP(hipStreamWaitEvent(r.st, r.ev_panel[buf], 0));        // the whole column panel of step k
    if (!h->chain_only) GH_CHECK(update(k, buf, k + 2, (int)nt - 1, false));
  }
  // Drain FIRST, copy afterwards.  A device-to-host copy into pageable memory blocks inside hipMemcpyAsync until the stream
  // has reached it (the HODLR solver's pinned result block exists for that reason), so with the copies in front the rank
  // thread sat in the copy -- not in mg_drain's bounded poll -- in exactly the crossed-communicator hang the time-out is
  // for, and a drain that gave up returned with copies still aimed at this stack frame.  Now: the bounded drains of the
  // chain and gather streams, then of the main stream, then two small copies into a PINNED block that lives in the rank.
  for (hipStream_t q : {r.sp, r.sg, r.st}) GH_CHECK(mg_drain(h, r, q));
  GH_HIP(hipMemcpyAsync(r.h_back, r.d_info, sizeof(long long), hipMemcpyDeviceToHost, r.st));
  GH_HIP(hipMemcpyAsync(r.h_back + 1, r.scal.d(), sizeof(double), hipMemcpyDeviceToHost, r.st));
  GH_CHECK(mg_drain(h, r, r.st));
  long long info = 0;
  memcpy(&info, r.h_back, sizeof(long long));
  r.info = info; r.logdet = r.h_back[1];
  return GH_OK;
}

// ------------------------------------------------------------------------------------------ tile sweeps
// Left-looking sweeps over tile rows on the sharded factor, R right-hand sides at a time, device resident.
//   forward  (tile row k, ascending):  the ranks of process row prow(k) form  sum_{j<k, j mine} L[k,j] Z_j  from the Z tiles they
//            keep for their tile columns; the partials are collected on the diagonal owner, which solves
//            Z_k = L_kk^-1 (B_k - sum) and sends Z_k down process column pcol(k) (the ranks that hold column-k tiles);
//   backward (descending): the mirror image along process columns with L[i,k]^T X_i, X_k sent along process row prow(k).
// Rp = 1: gemv + the chained 128-block substitution kernel; Rp a multiple of 128: GEMMs on the matrix pipe + blocked substitution.
// What `fill` puts into r.wk on the diagonal owner is tile k of the right-hand side; `take` sees the solved tile there.
struct Sweep {
  int64_t Rp = 1;                  // columns (1, or a multiple of 128)
  bool forward = true, backward = false;
  int k_first = 0;                 // forward: tile rows before this one have a zero right-hand side AND a zero solution
};

int sweep_buffers(gh_mgpu* h, MRank& r, int64_t Rp) {
  const int64_t nb = h->nb;
  const size_t nlr = std::max<size_t>(r.rows.size(), 1), nlc = std::max<size_t>(r.cols.size(), 1);
  GH_CHECK(r.zc.ensure(nlc * nb * Rp * sizeof(double)));
  GH_CHECK(r.xr.ensure(nlr * nb * Rp * sizeof(double)));
  GH_CHECK(r.wk.ensure(2 * (size_t)nb * Rp * sizeof(double)));
  GH_CHECK(r.part.ensure((size_t)nb * Rp * sizeof(double)));
  GH_CHECK(r.red.ensure((size_t)std::max(h->Pr, h->Pc) * nb * Rp * sizeof(double)));
  GH_CHECK(r.rhs.ensure((size_t)nb * Rp * sizeof(double)));
  GH_HIP(hipMemsetAsync((unsigned*)r.flags.p + nb / T + 1, 0, sizeof(unsigned), r.st));      // the sweep's sticky time-out word
  return GH_OK;
}

// W (nb x Rp, in place) <- L_kk^-1 W or L_kk^-T W for the factored diagonal tile at `lkk` (row pitch ld)
int tile_solve(MRank& r, const double* lkk, long ld, const double* dk, int64_t nb, double* w, int64_t Rp, bool trans, hipStream_t st) {
  if (Rp == 1) {
    double* z = w + nb;                                           // (r.wk holds two tiles: the chained kernel wants w != z)
    if (!trans) GH_CHECK(gh_dev_trsv_lower(lkk, ld, dk, nb, w, z, r.flags.p, st));
    else GH_CHECK(gh_dev_trsv_lower_t(lkk, ld, dk, nb, w, z, r.flags.p, st));
    GH_HIP(hipMemcpyAsync(w, z, (size_t)nb * sizeof(double), hipMemcpyDeviceToDevice, st));
    // (the kernel's time-out flag is cleared by the next launch: fold it into the word behind it, read once per sweep)
    hipLaunchKernelGGL(mg_sticky_flag_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)r.flags.p + nb / T, (unsigned*)r.flags.p + nb / T + 1);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  const int64_t nt = nb / T;
  if (!trans) {
    for (int64_t j = 0; j < nt; ++j) {
      double* wj = w + j * T * Rp;
      GH_CHECK(gh_dev_gemm(wj, Rp, dk + j * T * T, T, wj, Rp, T, Rp, T, 1.0, 0.0, GH_GEMM_B_NMAJOR, st));                 // W_j <- L_jj^-1 W_j
      if (j + 1 < nt)
        GH_CHECK(gh_dev_gemm(w + (j + 1) * T * Rp, Rp, lkk + (j + 1) * T * ld + j * T, ld, wj, Rp, (nt - j - 1) * T, Rp, T, -1.0, 1.0,
                             GH_GEMM_B_NMAJOR, st));                                                                        // rows below
    }
  } else {
    for (int64_t j = nt - 1; j >= 0; --j) {
      double* wj = w + j * T * Rp;
      GH_CHECK(gh_dev_gemm(wj, Rp, dk + j * T * T, T, wj, Rp, T, Rp, T, 1.0, 0.0, GH_GEMM_A_MMAJOR | GH_GEMM_B_NMAJOR, st));   // W_j <- L_jj^-T W_j
      if (j > 0)
        GH_CHECK(gh_dev_gemm(w, Rp, lkk + j * T * ld, ld, wj, Rp, j * T, Rp, T, -1.0, 1.0, GH_GEMM_A_MMAJOR | GH_GEMM_B_NMAJOR, st));   // rows above
    }
  }
  return GH_OK;
}

template <typename Fill, typename Take>
int rank_sweep(gh_mgpu* h, MRank& r, const Sweep& sw, Fill fill, Take take) {
  const int64_t nb = h->nb, nt = h->nt, Rp = sw.Rp;
  const long ld = (long)std::max<size_t>(r.cols.size(), 1) * nb;
  const int nlr = (int)r.rows.size();
  double* A = r.A.d();
  hipStream_t st = r.st;
  const size_t tile_elems = (size_t)nb * Rp;
  auto dk = [&](int k) { return r.dinv.d() + (long)k * (nb / T) * T * T; };
  if (sw.forward) {
    for (int k = sw.k_first; k < nt; ++k) {
      if (h->abort.load()) return mg_aborted();
      const int kr = h->prow(k), kc = h->pcol(k), root = grank(h, kr, kc);
      if (r.pr == kr) {
        const int c0 = first_at_least(r.cols, sw.k_first), cend = first_at_least(r.cols, k);        // my tile columns in [k_first, k)
        const bool have = cend > c0;
        Group& g = h->rowg[kr];
        if (g.members.size() > 1 || have) {
          if (have) {
            const double* lrow = A + (long)local_index(r.rows, k) * nb * ld + (long)c0 * nb;
            if (Rp == 1) GH_CHECK(gh_dev_gemv(lrow, ld, nb, (int64_t)(cend - c0) * nb, 0, r.zc.d() + (size_t)c0 * nb, r.part.d(), 1.0, 0.0, st));
            else GH_CHECK(gh_dev_gemm(r.part.d(), Rp, lrow, ld, r.zc.d() + (size_t)c0 * tile_elems, Rp, nb, Rp, (int64_t)(cend - c0) * nb, 1.0, 0.0,
                                      GH_GEMM_B_NMAJOR, st));
          } else {
            GH_HIP(hipMemsetAsync(r.part.p, 0, tile_elems * sizeof(double), st));
          }
          GH_CHECK(mg_collect(h, r, g, r.part.d(), tile_elems, root, r.red.d(), st, h->comms));
        }
        if (r.rank == root) {
          GH_CHECK(fill(k, r.rhs.d()));                                                       // B_k
          if (g.members.size() > 1 || have) GH_CHECK(mg_sub_sum(st, r.rhs.d(), r.red.d(), (int)g.members.size(), (long)tile_elems, r.wk.d()));
          else GH_HIP(hipMemcpyAsync(r.wk.p, r.rhs.p, tile_elems * sizeof(double), hipMemcpyDeviceToDevice, st));
          GH_CHECK(tile_solve(r, A + (long)local_index(r.rows, k) * nb * ld + (long)local_index(r.cols, k) * nb, ld, dk(k), nb, r.wk.d(), Rp, false, st));
          if (!sw.backward) GH_CHECK(take(k, r.wk.d()));
        }
      }
      if (r.pc == kc) {                                            // process column kc keeps Z_k for the tile rows to come
        double* zk = r.zc.d() + (size_t)local_index(r.cols, k) * tile_elems;
        if (r.rank == root) GH_HIP(hipMemcpyAsync(zk, r.wk.p, tile_elems * sizeof(double), hipMemcpyDeviceToDevice, st));
        GH_CHECK(mg_bcast(h, r, h->colg[kc], zk, tile_elems, root, st, h->comms));
      }
    }
  }
  if (sw.backward) {
    for (int k = (int)nt - 1; k >= 0; --k) {
      if (h->abort.load()) return mg_aborted();
      const int kr = h->prow(k), kc = h->pcol(k), root = grank(h, kr, kc);
      if (r.pc == kc) {
        const int li0 = first_at_least(r.rows, k + 1);               // my tile rows i > k
        const bool have = li0 < nlr;
        Group& g = h->colg[kc];
        if (g.members.size() > 1 || have) {
          if (have) {
            const double* lcol = A + (long)li0 * nb * ld + (long)local_index(r.cols, k) * nb;
            if (Rp == 1) GH_CHECK(gh_dev_gemv(lcol, ld, (int64_t)(nlr - li0) * nb, nb, 1, r.xr.d() + (size_t)li0 * nb, r.part.d(), 1.0, 0.0, st));
            else GH_CHECK(gh_dev_gemm(r.part.d(), Rp, lcol, ld, r.xr.d() + (size_t)li0 * tile_elems, Rp, nb, Rp, (int64_t)(nlr - li0) * nb, 1.0, 0.0,
                                      GH_GEMM_A_MMAJOR | GH_GEMM_B_NMAJOR, st));
          } else {
            GH_HIP(hipMemsetAsync(r.part.p, 0, tile_elems * sizeof(double), st));
          }
          GH_CHECK(mg_collect(h, r, g, r.part.d(), tile_elems, root, r.red.d(), st, h->comms));
        }
        if (r.rank == root) {
          const double* zk = sw.forward ? r.zc.d() + (size_t)local_index(r.cols, k) * tile_elems : nullptr;
          if (!sw.forward) { GH_CHECK(fill(k, r.rhs.d())); zk = r.rhs.d(); }
          if (g.members.size() > 1 || have) GH_CHECK(mg_sub_sum(st, zk, r.red.d(), (int)g.members.size(), (long)tile_elems, r.wk.d()));
          else GH_HIP(hipMemcpyAsync(r.wk.p, zk, tile_elems * sizeof(double), hipMemcpyDeviceToDevice, st));
          GH_CHECK(tile_solve(r, A + (long)local_index(r.rows, k) * nb * ld + (long)local_index(r.cols, k) * nb, ld, dk(k), nb, r.wk.d(), Rp, true, st));
          GH_CHECK(take(k, r.wk.d()));
        }
      }
      if (r.pr == kr) {                                            // process row kr keeps X_k for the tile columns to come
        double* xk = r.xr.d() + (size_t)local_index(r.rows, k) * tile_elems;
        if (r.rank == root) GH_HIP(hipMemcpyAsync(xk, r.wk.p, tile_elems * sizeof(double), hipMemcpyDeviceToDevice, st));
        GH_CHECK(mg_bcast(h, r, h->rowg[kr], xk, tile_elems, root, st, h->comms));
      }
    }
  }
  GH_HIP(hipStreamSynchronize(st));
  return GH_OK;
}

// the chained substitution kernels give up after 2 s without their predecessor; the flag sits behind the block flags
int sweep_check_flag(gh_mgpu* h, MRank& r) {
  int failed = 0;
  GH_HIP(hipMemcpy(&failed, (unsigned*)r.flags.p + h->nb / T + 1, sizeof(int), hipMemcpyDeviceToHost));
  if (failed) { gh_set_error("triangular sweep: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
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
    if (opts->transport == GH_MGPU_RCCL && !(getenv("GEORGE_AMD_RCCL_LIB") && rccl_load() && g_rccl.shared_device_ok))
      for (int j = 0; j < i; ++j)
        if (opts->devices[j] == opts->devices[i]) { gh_set_error("device %d listed twice: RCCL needs one rank per device (GH_MGPU_COPY accepts virtual devices)", opts->devices[i]); return GH_ERR_BAD_ARG; }
  }
  int Pr = opts->pr, Pc = opts->pc;
  if (Pr <= 0 || Pc <= 0) { Pr = W; Pc = 1; }                  // whole tile rows per rank (head of the file: why)
  if (Pr * Pc != W) { gh_set_error("grid %d x %d does not hold %d devices", Pr, Pc, W); return GH_ERR_BAD_ARG; }
  gh_mgpu* h = new gh_mgpu();
  h->opts = *opts; h->W = W; h->Pr = Pr; h->Pc = Pc;
  h->snake = (Pc == 1 && Pr > 1) && !(opts->flags & GH_MGPU_PLAIN_CYCLIC);
  h->chain_only = (opts->flags & GH_MGPU_CHAIN_ONLY) != 0;
  h->trace_on = (opts->flags & GH_MGPU_TRACE) != 0;
  h->ranks.resize(W);
  h->rowg = std::vector<Group>(Pr); h->colg = std::vector<Group>(Pc);
  for (int i = 0; i < W; ++i) {
    MRank& r = h->ranks[i];
    r.rank = i; r.dev = opts->devices[i]; r.pr = i / Pc; r.pc = i % Pc;
    h->rowg[r.pr].members.push_back(i); h->colg[r.pc].members.push_back(i); h->world.members.push_back(i);
    int plo = 0, phi = 0;                                       // numerically lowest value = highest priority
    if (hipSetDevice(r.dev) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
    gh_prime_device(r.dev);                  // (gh_common.h: the null stream must have seen a launch before the first stream is made)
    bool ok = hipSetDevice(r.dev) == hipSuccess && hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithPriority(&r.sp, hipStreamNonBlocking, phi) == hipSuccess &&
              hipStreamCreateWithFlags(&r.sg, hipStreamNonBlocking) == hipSuccess;
    for (hipEvent_t* e : {&r.ev_ready, &r.ev_done, &r.ev_fast[0], &r.ev_fast[1], &r.ev_panel[0], &r.ev_panel[1], &r.ev_bcol, &r.ev_rest})
      ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); delete h; gh_set_error("stream / event creation failed on device %d", opts->devices[i]); return GH_ERR_HIP; }
  }
  for (auto* gs : {&h->rowg, &h->colg}) for (auto& g : *gs) { g.bar.n = (int)g.members.size(); g.bar.abort = &h->abort; }
  h->world.bar.n = W; h->world.bar.abort = &h->abort;
  if (opts->transport == GH_MGPU_RCCL) {
    if (!rccl_load()) { delete h; gh_set_error("RCCL unavailable: %s", g_rccl.why.c_str()); return GH_ERR_HIP; }
    ncclResult_t rc = g_rccl.CommInitAll(h->comms, W, opts->devices);
    if (rc != ncclSuccess) { delete h; gh_set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(rc)); return GH_ERR_HIP; }
    h->have_comms = true;
    if (W > 1) {
      rc = g_rccl.CommInitAll(h->comms_b, W, opts->devices);
      if (rc != ncclSuccess) { delete h; gh_set_error("ncclCommInitAll (bulk communicator) failed: %s", g_rccl.GetErrorString(rc)); return GH_ERR_HIP; }
      h->have_comms_b = true;
    } else {
      h->comms_b[0] = h->comms[0];                               // (a world of one never transfers)
    }
    // TWO communicators are driven at once (chain(k+2) on sp, the bulk gather of step k+1 on sg) and nothing orders them
    // across ranks: if sp and sg share a hardware queue on one rank and not on another, rank A may dispatch the gather's
    // transfer kernel in front of the chain's while rank B does the opposite -- each then waits for the peer's kernel that is
    // queued behind the other: a hang (the multi-communicator hazard of the NCCL documentation).  So: on every rank, does a
    // one-workgroup kernel on sg complete while a long grid runs on sp, and the other way round (the probe of gh_chol.hip)?
    // If not on ANY rank -- or GH_MGPU_ONE_COMM asks for it -- the gather goes out on sp with the chain's communicator, in
    // program order, the same on every rank.  (What it costs: the gather of step k+1 no longer overlaps the chain of k+2.)
    h->one_comm = (opts->flags & GH_MGPU_ONE_COMM) != 0;
    if (W > 1 && !h->one_comm) {
      for (int i = 0; i < W && !h->one_comm; ++i) {
        MRank& r = h->ranks[i];
        if (hipSetDevice(r.dev) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (!gh_streams_dispatch_independently(r.sp, r.sg)) h->one_comm = true;
      }
    }
    // self-check of both communicators: all-reduce of (rank + 1) must give W (W + 1) / 2 on every rank
    for (ncclComm_t* cs : {h->comms, h->comms_b}) {
      std::vector<double> got(W, 0.0);
      int rcs = mg_run(h, [&](MRank& r) -> int {
        double v = (double)(r.rank + 1);
        GH_CHECK(r.scal.ensure(64 * sizeof(double)));
        GH_HIP(hipMemcpyAsync(r.scal.p, &v, sizeof(double), hipMemcpyHostToDevice, r.st));
        MG_NCCL(g_rccl.AllReduce(r.scal.p, (double*)r.scal.p + 1, 1, ncclDouble, ncclSum, cs[r.rank], r.st));
        GH_HIP(hipMemcpyAsync(&got[r.rank], (double*)r.scal.p + 1, sizeof(double), hipMemcpyDeviceToHost, r.st));
        GH_HIP(hipStreamSynchronize(r.st));
        return GH_OK;
      });
      if (rcs != GH_OK) { delete h; return rcs; }
      for (int i = 0; i < W; ++i)
        if (got[i] != 0.5 * W * (W + 1)) { delete h; gh_set_error("RCCL self-check failed on rank %d: all-reduce gave %g, expected %g", i, got[i], 0.5 * W * (W + 1)); return GH_ERR_HIP; }
    }
  }
  *out = h;
  return GH_OK;
}

extern "C" void gh_mgpu_destroy(gh_mgpu* h) { delete h; }
extern "C" int64_t gh_mgpu_info(const gh_mgpu* h) { return h ? h->info : 0; }
extern "C" int gh_mgpu_comm_mode(const gh_mgpu* h) {
  if (!h) return -1;
  return h->opts.transport != GH_MGPU_RCCL ? 0 : (h->one_comm ? 1 : 2);
}
int gh_mgpu_grid(const gh_mgpu* h, int32_t* pr, int32_t* pc, int32_t* nb) {
  if (!h) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (pr) *pr = h->Pr;
  if (pc) *pc = h->Pc;
  if (nb) *nb = (int32_t)h->nb;
  return GH_OK;
}
extern "C" int gh_mgpu_owner(const gh_mgpu* h, int64_t tile_row, int64_t tile_col) {
  if (!h || tile_row < 0 || tile_col < 0) return -1;
  return grank(h, h->prow((int)tile_row), h->pcol((int)tile_col));
}

extern "C" int gh_mgpu_compute(gh_mgpu* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim, const double* yerr,
                               double* logdet_out) {
  if (!h || !k || !x || !yerr || n <= 0) { gh_set_error("bad argument to compute"); return GH_ERR_BAD_ARG; }
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  if (h->dead.load()) { gh_set_error("this multi-GPU solver timed out earlier (mg_drain) and is dead: make a new one"); return GH_ERR_HIP; }
  h->computed = false; h->info = 0;
  h->n = n; h->ndim = ndim;
  h->nb = h->opts.nb > 0 ? h->opts.nb : (n >= 24576 ? 1024 : 512);
  h->nt = (n + h->nb - 1) / h->nb;
  int rc = mg_run(h, [&](MRank& r) -> int {
    GH_CHECK(rank_setup(h, r, k, x, yerr));
    // every rank has its buffers (or the run is over) BEFORE any of them enqueues a transfer: a rank that failed here --
    // GH_ERR_NOMEM on a GPU somebody else is using, say -- must not leave the others blocked in a receive for ever
    if (!h->world.bar.wait()) return mg_aborted();
    return rank_factor(h, r);
  });
  if (rc != GH_OK) return rc;
  h->trace_rows.clear();
  if (h->trace_on)
    for (auto& r : h->ranks) {
      (void)hipSetDevice(r.dev);
      for (auto& e : r.trace) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) { (void)hipGetLastError(); ms = -1.f; }
        for (double v : {(double)r.rank, (double)e.step, (double)e.phase, (double)ms, e.units}) h->trace_rows.push_back(v);
      }
    }
  double tot = 0.0;
  long long bad = 0;
  for (auto& r : h->ranks) {                                   // fixed rank order: reproducible
    tot += r.logdet;
    if (r.info > 0 && (bad == 0 || r.info < bad)) bad = r.info;
  }
  if (h->chain_only) {                                         // (a timing run: the trailing updates were skipped, the numbers mean nothing)
    if (logdet_out) *logdet_out = tot;
    return GH_OK;
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

extern "C" int gh_mgpu_get_trace(const gh_mgpu* h, double* out, int64_t max_rows, int64_t* n_rows) {
  if (!h || !n_rows) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  const int64_t have = (int64_t)h->trace_rows.size() / 5;
  *n_rows = have;
  if (out && max_rows > 0) memcpy(out, h->trace_rows.data(), (size_t)std::min(have, max_rows) * 5 * sizeof(double));
  return GH_OK;
}

static int mg_need(gh_mgpu* h) {
  if (!h) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (!h->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  return GH_OK;
}

// copy rows [row0, row0 + nb) x columns [c0, c0 + nc) of the host matrix b (n x ldb) into the device tile w (nb x Rp), zero padded
static int mg_load_tile(gh_mgpu* h, MRank& r, const double* b, int64_t ldb, int64_t row0, int64_t c0, int64_t nc, double* w, int64_t Rp) {
  const int64_t nb = h->nb, vr = std::max<int64_t>(0, std::min<int64_t>(nb, h->n - row0));
  if (vr < nb || nc < Rp) GH_HIP(hipMemsetAsync(w, 0, (size_t)nb * Rp * sizeof(double), r.st));
  if (vr > 0)
    GH_HIP(hipMemcpy2DAsync(w, Rp * sizeof(double), b + row0 * ldb + c0, ldb * sizeof(double), nc * sizeof(double), vr, hipMemcpyHostToDevice, r.st));
  return GH_OK;
}
static int mg_store_tile(gh_mgpu* h, MRank& r, const double* w, int64_t Rp, double* out, int64_t ldo, int64_t row0, int64_t c0, int64_t nc) {
  const int64_t vr = std::max<int64_t>(0, std::min<int64_t>(h->nb, h->n - row0));
  if (vr > 0)
    GH_HIP(hipMemcpy2DAsync(out + row0 * ldo + c0, ldo * sizeof(double), w, Rp * sizeof(double), nc * sizeof(double), vr, hipMemcpyDeviceToHost, r.st));
  return GH_OK;
}
#define MG_RCHUNK 2048            // right-hand sides per sweep (buffers: (N / Pc + N / Pr) x chunk doubles per rank)

extern "C" int gh_mgpu_dot_solve(gh_mgpu* h, const double* y, double* out) {
  GH_CHECK(mg_need(h));
  if (!y || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  // y^T K^-1 y = || L^-1 y ||^2 : the forward sweep only (basic.py:102 does both)
  GH_CHECK(mg_run(h, [&](MRank& r) -> int {
    GH_CHECK(sweep_buffers(h, r, 1));
    GH_CHECK(r.acc.ensure(8 * sizeof(double)));
    GH_HIP(hipMemsetAsync(r.acc.p, 0, 8 * sizeof(double), r.st));
    Sweep sw; sw.Rp = 1;
    GH_CHECK(rank_sweep(h, r, sw,
        [&](int k, double* w) -> int { return mg_load_tile(h, r, y, 1, (int64_t)k * h->nb, 0, 1, w, 1); },
        [&](int, const double* z) -> int {
          hipLaunchKernelGGL(mg_sumsq_kernel, dim3(1), dim3(256), 0, r.st, z, (long)h->nb, r.acc.d());
          GH_HIP(hipGetLastError());
          return GH_OK;
        }));
    GH_HIP(hipMemcpy(&r.acc_host, r.acc.p, sizeof(double), hipMemcpyDeviceToHost));
    return sweep_check_flag(h, r);
  }));
  double acc = 0.0;
  for (auto& r : h->ranks) acc += r.acc_host;                  // fixed rank order
  *out = acc;
  return GH_OK;
}

extern "C" int gh_mgpu_solve(gh_mgpu* h, const double* b, int64_t nrhs, double* out) {
  GH_CHECK(mg_need(h));
  if (nrhs < 0 || (nrhs > 0 && (!b || !out))) { gh_set_error("bad argument to solve"); return GH_ERR_BAD_ARG; }
  if (gh_is_device_ptr(b) || gh_is_device_ptr(out)) { gh_set_error("gh_mgpu_solve takes host pointers"); return GH_ERR_BAD_ARG; }
  std::vector<double> tmp;
  const double* src = b;
  if (b == out && nrhs > 1) { tmp.assign(b, b + (size_t)h->n * nrhs); src = tmp.data(); }      // (tiles are stored while later ones are still read)
  for (int64_t c0 = 0; c0 < nrhs; c0 += MG_RCHUNK) {
    const int64_t nc = std::min<int64_t>(MG_RCHUNK, nrhs - c0), Rp = nrhs == 1 ? 1 : gh_round_up(nc, T);
    GH_CHECK(mg_run(h, [&](MRank& r) -> int {
      GH_CHECK(sweep_buffers(h, r, Rp));
      Sweep sw; sw.Rp = Rp; sw.backward = true;
      GH_CHECK(rank_sweep(h, r, sw,
          [&](int k, double* w) -> int { return mg_load_tile(h, r, src, nrhs, (int64_t)k * h->nb, c0, nc, w, Rp); },
          [&](int k, const double* xk) -> int { return mg_store_tile(h, r, xk, Rp, out, nrhs, (int64_t)k * h->nb, c0, nc); }));
      return Rp == 1 ? sweep_check_flag(h, r) : GH_OK;
    }));
  }
  return GH_OK;
}

// basic.py:116-121: K^-1 = cho_solve(factor, I), here in column chunks through the sharded sweeps; the identity tile of
// a chunk is made on the diagonal owner, and tile rows above the chunk's first column have nothing to solve going forward
extern "C" int gh_mgpu_get_inverse(gh_mgpu* h, double* out) {
  GH_CHECK(mg_need(h));
  if (!out || gh_is_device_ptr(out)) { gh_set_error("gh_mgpu_get_inverse takes a host pointer"); return GH_ERR_BAD_ARG; }
  const int64_t n = h->n;
  for (int64_t c0 = 0; c0 < n; c0 += MG_RCHUNK) {
    const int64_t nc = std::min<int64_t>(MG_RCHUNK, n - c0), Rp = gh_round_up(nc, T);
    GH_CHECK(mg_run(h, [&](MRank& r) -> int {
      GH_CHECK(sweep_buffers(h, r, Rp));
      Sweep sw; sw.Rp = Rp; sw.backward = true; sw.k_first = (int)(c0 / h->nb);
      // (tile columns before k_first keep a zero Z: the backward sweep reads every Z tile)
      for (size_t lj = 0; lj < r.cols.size() && r.cols[lj] < sw.k_first; ++lj)
        GH_HIP(hipMemsetAsync(r.zc.d() + lj * (size_t)h->nb * Rp, 0, (size_t)h->nb * Rp * sizeof(double), r.st));
      return rank_sweep(h, r, sw,
          [&](int k, double* w) -> int {
            const long tot = (long)h->nb * Rp;
            hipLaunchKernelGGL(mg_eye_tile_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, r.st, w, (long)Rp, (long)h->nb, (long)Rp,
                               (long)k * h->nb, (long)c0);
            GH_HIP(hipGetLastError());
            return GH_OK;
          },
          [&](int k, const double* xk) -> int { return mg_store_tile(h, r, xk, Rp, out, n, (int64_t)k * h->nb, c0, nc); });
    }));
  }
  return GH_OK;
}

// basic.py:104-114: out = r @ U, U = L^T, i.e. out[:, tile j] = sum_{k <= j} r[:, tile k] L[j, k]^T -- no sweep: every tile row of L
// works alone; the ranks of its process row form partial products over their tile columns, collected on the diagonal owner
extern "C" int gh_mgpu_apply_sqrt(gh_mgpu* h, const double* rin, int64_t nrows, double* out) {
  GH_CHECK(mg_need(h));
  if (!rin || !out || nrows <= 0) { gh_set_error("bad argument to apply_sqrt"); return GH_ERR_BAD_ARG; }
  if (gh_is_device_ptr(rin) || gh_is_device_ptr(out)) { gh_set_error("gh_mgpu_apply_sqrt takes host pointers"); return GH_ERR_BAD_ARG; }
  const int64_t n = h->n, nb = h->nb;
  for (int64_t s0 = 0; s0 < nrows; s0 += MG_RCHUNK) {
    const int64_t ns = std::min<int64_t>(MG_RCHUNK, nrows - s0), Sp = gh_round_up(ns, T);
    GH_CHECK(mg_run(h, [&](MRank& r) -> int {
      const long ld = (long)std::max<size_t>(r.cols.size(), 1) * nb;
      const size_t nlc = std::max<size_t>(r.cols.size(), 1);
      hipStream_t st = r.st;
      // the sample columns of my tile columns: (Sp x nlc nb), zero padded
      GH_CHECK(r.zc.ensure((size_t)Sp * nlc * nb * sizeof(double)));
      GH_CHECK(r.part.ensure((size_t)Sp * nb * sizeof(double)));
      GH_CHECK(r.red.ensure((size_t)h->Pc * Sp * nb * sizeof(double)));
      GH_CHECK(r.wk.ensure((size_t)Sp * nb * sizeof(double)));
      GH_HIP(hipMemsetAsync(r.zc.p, 0, (size_t)Sp * nlc * nb * sizeof(double), st));
      for (size_t lj = 0; lj < r.cols.size(); ++lj) {
        const int64_t c0 = (int64_t)r.cols[lj] * nb, vc = std::max<int64_t>(0, std::min<int64_t>(nb, n - c0));
        if (vc > 0)
          GH_HIP(hipMemcpy2DAsync(r.zc.d() + lj * nb, nlc * nb * sizeof(double), rin + s0 * n + c0, n * sizeof(double), vc * sizeof(double), ns,
                                  hipMemcpyHostToDevice, st));
      }
      for (int j = 0; j < (int)h->nt; ++j) {
        const int jr = h->prow(j), jc = h->pcol(j), root = grank(h, jr, jc);
        if (r.pr != jr) continue;
        Group& g = h->rowg[jr];
        const int cend = first_at_least(r.cols, j);
        const double* lrow = r.A.d() + (long)local_index(r.rows, j) * nb * ld;
        double* dst = (g.members.size() > 1) ? r.part.d() : r.wk.d();
        bool any = false;
        if (cend > 0) {
          GH_CHECK(gh_dev_gemm(dst, nb, r.zc.d(), (int64_t)nlc * nb, lrow, ld, Sp, nb, (int64_t)cend * nb, 1.0, 0.0, 0, st));
          any = true;
        }
        if (r.rank == root) {                                      // the diagonal tile: lower-triangular in (n, k)
          const int lcj = local_index(r.cols, j);
          GH_CHECK(gh_dev_gemm(dst, nb, r.zc.d() + (size_t)lcj * nb, (int64_t)nlc * nb, lrow + (long)lcj * nb, ld, Sp, nb, nb, 1.0, any ? 1.0 : 0.0,
                               GH_GEMM_KHI_COL, st));
          any = true;
        }
        if (!any) GH_HIP(hipMemsetAsync(dst, 0, (size_t)Sp * nb * sizeof(double), st));
        if (g.members.size() > 1) {
          GH_CHECK(mg_collect(h, r, g, r.part.d(), (size_t)Sp * nb, root, r.red.d(), st, h->comms));
          if (r.rank == root) {
            GH_CHECK(mg_sub_sum(st, nullptr, r.red.d(), (int)g.members.size(), (long)Sp * nb, r.wk.d()));     // wk = -sum
          }
        }
        if (r.rank == root) {
          const int64_t c0 = (int64_t)j * nb, vc = std::max<int64_t>(0, std::min<int64_t>(nb, n - c0));
          if (vc > 0) {
            GH_HIP(hipMemcpy2DAsync(out + s0 * n + c0, n * sizeof(double), r.wk.d(), nb * sizeof(double), vc * sizeof(double), ns, hipMemcpyDeviceToHost, st));
            GH_HIP(hipStreamSynchronize(st));
            if (g.members.size() > 1)                              // (collected as a negated sum)
              for (int64_t q = 0; q < ns; ++q) for (int64_t c = 0; c < vc; ++c) out[(s0 + q) * n + c0 + c] = -out[(s0 + q) * n + c0 + c];
          }
        }
      }
      GH_HIP(hipStreamSynchronize(st));
      return GH_OK;
    }));
  }
  return GH_OK;
}

// GP.predict (gp.py:482-545) on the sharded factor: V = L^-1 K(x, xs) by the forward sweep (K(x, xs) is built tile by tile on the
// diagonal owners), z = L^-1 r;  mu = V^T z,  var = k(xs, xs) - colsum(V^2),  cov = K(xs, xs) - V^T V  -- accumulated per rank over
// its diagonal tiles and added in rank order on the host.  Host pointers.
extern "C" int gh_mgpu_predict(gh_mgpu* h, gh_kernel* k, const double* rvec, const double* xs, int64_t m, double* mu, double* var, double* cov) {
  GH_CHECK(mg_need(h));
  if (!k || !rvec || !xs || !mu || m <= 0) { gh_set_error("bad argument to predict"); return GH_ERR_BAD_ARG; }
  if (k->ndim != h->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  const int64_t n = h->n, nb = h->nb, W = h->W;
  if (cov && m > MG_RCHUNK) { gh_set_error("gh_mgpu_predict: the full covariance is offered up to %d test points", MG_RCHUNK); return GH_ERR_BAD_ARG; }
  // z = L^-1 r, every tile kept on the host (n doubles)
  std::vector<double> z((size_t)h->nt * nb, 0.0);
  GH_CHECK(mg_run(h, [&](MRank& r) -> int {
    GH_CHECK(sweep_buffers(h, r, 1));
    Sweep sw; sw.Rp = 1;
    GH_CHECK(rank_sweep(h, r, sw,
        [&](int kk, double* w) -> int { return mg_load_tile(h, r, rvec, 1, (int64_t)kk * nb, 0, 1, w, 1); },
        [&](int kk, const double* zk) -> int {
          GH_HIP(hipMemcpyAsync(z.data() + (size_t)kk * nb, zk, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, r.st));
          return GH_OK;
        }));
    return sweep_check_flag(h, r);
  }));
  for (int64_t c0 = 0; c0 < m; c0 += MG_RCHUNK) {
    const int64_t nc = std::min<int64_t>(MG_RCHUNK, m - c0), Rp = gh_round_up(nc, T);
    std::vector<double> pmu((size_t)W * Rp, 0.0), psq((size_t)W * Rp, 0.0), pcov;
    if (cov) pcov.assign((size_t)W * Rp * Rp, 0.0);
    GH_CHECK(mg_run(h, [&](MRank& r) -> int {
      hipStream_t st = r.st;
      GH_CHECK(sweep_buffers(h, r, Rp));
      // this rank's copy of the kernel (hyper-parameters as given NOW: predict may be asked with another kernel object)
      gh_kernel kloc;
      kloc.nodes = k->nodes; kloc.ndim = k->ndim; kloc.size = k->size; kloc.fast = k->fast; kloc.device = -1;
      GH_CHECK(kloc.upload());
      GhBuf xsd, zd, acc, cv;
      GH_CHECK(xsd.ensure((size_t)nc * h->ndim * sizeof(double)));
      GH_CHECK(gh_to_device(xsd.d(), xs + c0 * h->ndim, (size_t)nc * h->ndim, st));
      GH_CHECK(zd.ensure((size_t)nb * sizeof(double)));
      GH_CHECK(acc.ensure(2 * (size_t)Rp * sizeof(double)));
      GH_HIP(hipMemsetAsync(acc.p, 0, 2 * (size_t)Rp * sizeof(double), st));
      if (cov) { GH_CHECK(cv.ensure((size_t)Rp * Rp * sizeof(double))); GH_HIP(hipMemsetAsync(cv.p, 0, (size_t)Rp * Rp * sizeof(double), st)); }
      Sweep sw; sw.Rp = Rp;
      GH_CHECK(rank_sweep(h, r, sw,
          [&](int kk, double* w) -> int {                           // K(x[tile kk], xs[chunk]); rows past n are zero
            const int64_t row0 = (int64_t)kk * nb, vr = std::max<int64_t>(0, std::min<int64_t>(nb, n - row0));
            return gh_launch_kmat(&kloc, r.x.d() + std::min(row0, n - 1) * h->ndim, vr, xsd.d(), nc, nullptr, w, Rp, nb, Rp, 0, 0, false, false, st);
          },
          [&](int kk, const double* vk) -> int {
            GH_HIP(hipMemcpyAsync(zd.p, z.data() + (size_t)kk * nb, (size_t)nb * sizeof(double), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(mg_colacc_kernel, dim3((unsigned)((Rp + 255) / 256)), dim3(256), 0, st, vk, (long)Rp, (long)nb, (long)Rp, zd.d(),
                               acc.d(), acc.d() + Rp);
            GH_HIP(hipGetLastError());
            if (cov) GH_CHECK(gh_dev_gemm(cv.d(), Rp, vk, Rp, vk, Rp, Rp, Rp, nb, 1.0, 1.0, GH_GEMM_A_MMAJOR | GH_GEMM_B_NMAJOR, st));
            return GH_OK;
          }));
      GH_HIP(hipMemcpy(pmu.data() + (size_t)r.rank * Rp, acc.p, (size_t)Rp * sizeof(double), hipMemcpyDeviceToHost));
      GH_HIP(hipMemcpy(psq.data() + (size_t)r.rank * Rp, acc.d() + Rp, (size_t)Rp * sizeof(double), hipMemcpyDeviceToHost));
      if (cov) GH_HIP(hipMemcpy(pcov.data() + (size_t)r.rank * Rp * Rp, cv.p, (size_t)Rp * Rp * sizeof(double), hipMemcpyDeviceToHost));
      // k(xs, xs) and K(xs, xs) come from rank 0
      if (r.rank == 0) {
        if (var) {
          GhBuf kd;
          GH_CHECK(kd.ensure((size_t)nc * sizeof(double)));
          GH_CHECK(gh_launch_kdiag(&kloc, xsd.d(), xsd.d(), nc, kd.d(), st));
          GH_CHECK(gh_from_device(var + c0, kd.d(), (size_t)nc, st));
        }
        if (cov) {
          GhBuf kk;
          GH_CHECK(kk.ensure((size_t)Rp * Rp * sizeof(double)));
          GH_CHECK(gh_launch_kmat(&kloc, xsd.d(), nc, xsd.d(), nc, nullptr, kk.d(), Rp, Rp, Rp, 0, 0, true, false, st));
          GH_HIP(hipMemcpy2DAsync(cov, m * sizeof(double), kk.d(), Rp * sizeof(double), nc * sizeof(double), nc, hipMemcpyDeviceToHost, st));
        }
        GH_HIP(hipStreamSynchronize(st));
      }
      return GH_OK;
    }));
    for (int64_t c = 0; c < nc; ++c) {
      double a = 0.0, b = 0.0;
      for (int64_t q = 0; q < W; ++q) { a += pmu[(size_t)q * Rp + c]; b += psq[(size_t)q * Rp + c]; }     // fixed rank order
      mu[c0 + c] = a;
      if (var) var[c0 + c] -= b;
    }
    if (cov)
      for (int64_t i = 0; i < nc; ++i)
        for (int64_t j = 0; j < nc; ++j) {
          double a = 0.0;
          for (int64_t q = 0; q < W; ++q) a += pcov[(size_t)q * Rp * Rp + (size_t)i * Rp + j];
          cov[i * m + j] -= a;
        }
  }
  return GH_OK;
}
