// gh_common.h -- shared host-side plumbing for libgeorge_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/george_amd.h"
#include "gh_eval.h"

#define GH_TILE 128             // tile edge of every blocked kernel; device matrices are padded to it

void gh_set_error(const char* fmt, ...);

#define GH_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      gh_set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_),   \
                   __FILE__, __LINE__, #expr);                                         \
      return GH_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define GH_CHECK(expr)                 \
  do {                                 \
    int rc_ = (expr);                  \
    if (rc_ != GH_OK) return rc_;      \
  } while (0)

static inline int64_t gh_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// true when p is device (HBM) memory
bool gh_is_device_ptr(const void* p);

// Cache of released device blocks for TRANSIENT buffers.  hipMalloc / hipFree cost 20-100 us each
// and hipFree synchronises the device; the HODLR compute() makes ~180 short-lived allocations
// (per-level job tables, batched-inverse scratch), 6 of its 28 ms.  Pooled buffers hand their block
// back to the cache instead; a later request takes the smallest cached block that fits (and is not
// more than 4x too large).  Only for buffers whose users all run on ONE stream (a block may be
// reused while the previous user's kernels are still queued -- stream order then protects it).
void* gh_pool_acquire(size_t bytes, size_t* capacity);     // nullptr on allocation failure
void gh_pool_release(void* p, size_t capacity);
void gh_pool_trim();                                       // really free every cached block of the current device
size_t gh_pool_parked_bytes();                             // released blocks parked in the current device's cache

// RAII device buffer
struct GhBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool pooled = false;           // release into / acquire from the block cache above
  ~GhBuf() { release(); }
  void release() {
    if (!p) return;
    if (pooled) gh_pool_release(p, bytes); else (void)hipFree(p);
    p = nullptr; bytes = 0;
  }
  int ensure(size_t nbytes) {
    if (nbytes <= bytes && p) return GH_OK;
    release();
    if (pooled) {
      size_t cap = 0;
      p = gh_pool_acquire(nbytes ? nbytes : 8, &cap);
      if (!p) { gh_set_error("hipMalloc of %zu bytes failed", nbytes); return GH_ERR_NOMEM; }
      bytes = cap;
      return GH_OK;
    }
    hipError_t e = hipMalloc(&p, nbytes ? nbytes : 8);
    if (e != hipSuccess) {                                  // (the block cache may hold tens of GB of released blocks: drop them, once)
      (void)hipGetLastError();
      gh_pool_trim();
      e = hipMalloc(&p, nbytes ? nbytes : 8);
    }
    if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); gh_set_error("hipMalloc of %zu bytes failed: %s", nbytes, hipGetErrorString(e)); return GH_ERR_NOMEM; }
    bytes = nbytes;
    return GH_OK;
  }
  double* d() const { return (double*)p; }
};
struct GhPooledBuf : GhBuf { GhPooledBuf() { pooled = true; } };

// copy `count` doubles from src (host or device) into device memory dst
int gh_to_device(double* dst, const double* src, size_t count, hipStream_t st);
// copy `count` doubles from device src into dst (host or device)
int gh_from_device(double* dst, const double* src, size_t count, hipStream_t st);

// --------------------------------------------------------------- kernel handle
struct gh_kernel {
  std::vector<GhNode> nodes;   // postfix program (host copy)
  int ndim = 0;
  int size = 0;                // full parameter count
  int device = -1;
  GhNode* d_nodes = nullptr;   // device copy (lazily uploaded per device)
  GhFast fast;                 // affine single-leaf form a + b*F(r2), fast.ok != 0 when it exists
  ~gh_kernel() { if (d_nodes) (void)hipFree(d_nodes); }
  int upload();                // ensure d_nodes valid on the current device
};

// ----------------------------------------------------------------- launchers
// (all device pointers; sizes need not be tile multiples unless noted)
// out[r*ldo + c] = k(x1[r], x2[c]) for r < n1, c < n2; rows/cols up to (rows_p, cols_p) are
// zero-filled (identity when `sym`).  sym: x1 == x2, evaluate with ordered arguments
// (k(x_min, x_max), mirroring kernel_interface.cpp:62-77), add yerr[r]^2 on the diagonal when
// yerr != NULL; lower_only: build only 128-tiles on/below the diagonal.
int gh_launch_kmat(const gh_kernel* k, const double* x1, int64_t n1, const double* x2, int64_t n2,
                   const double* yerr, double* out, int64_t ldo, int64_t rows_p, int64_t cols_p,
                   int64_t row0, int64_t col0, bool sym, bool lower_only, hipStream_t st);
int gh_launch_kdiag(const gh_kernel* k, const double* x1, const double* x2, int64_t n, double* out, hipStream_t st);
int gh_launch_kgrad(const gh_kernel* k, const uint32_t* which_host, const double* x1, int64_t n1,
                    const double* x2, int64_t n2, bool sym, double* out, hipStream_t st);
int gh_launch_kxgrad(const gh_kernel* k, int which_arg, const double* x1, int64_t n1,
                     const double* x2, int64_t n2, double* out, hipStream_t st);
// grad[p] = sum_{i>=j} w_ij (alpha_i alpha_j - Kinv[i][j]) dK_ij/dtheta_p, w = 1/2 on the diagonal, 1 below
// (== 1/2 sum_ij A_ij dK_ij/dtheta_p, gp.py:437,465-466); diagA[i] = alpha_i^2 - Kinv[i][i].
int gh_launch_kgrad_reduce(const gh_kernel* k, const uint32_t* which_host, const double* x, int64_t n,
                           const double* alpha, const double* kinv, int64_t ld, double* grad_dev /* size */,
                           double* diagA /* n or NULL */, GhBuf& scratch, hipStream_t st);

// fp64 GEMM family on the MFMA pipe:  C = beta*C + alpha * op(A) * op(B)^T-ish.  See gh_gemm.hip.
struct GhGemm {
  double* C; int64_t ldc;
  const double* A; int64_t lda;   // a_km: A(m,k) at A[m*lda + k];  else at A[k*lda + m]
  const double* B; int64_t ldb;   // b_km: B(n,k) at B[n*ldb + k];  else at B[k*ldb + n]
  int64_t M, N, K;                // M, N multiples of 128; K multiple of 16
  double alpha, beta;
  bool a_km, b_km;
  bool lower;                     // square C: only tiles with tile_row >= tile_col
  bool klo_max;                   // k starts at max(row0, col0) of the tile (operands lower-triangular in k)
  bool khi_col;                   // k ends at col0 + 128            (B lower-triangular: B(n,k) = 0 for k > n)
  bool khi_row;                   // k ends at row0 + 128            (A lower-triangular: A(m,k) = 0 for k > m)
  bool small_lds;                 // keep to <= 32 KiB of LDS per workgroup: the launch runs beside a SYRK that owns every CU
  // "staircase" C (gh_dev_gemm_nt_stair): M = stair_n * stair_rows; row group g (stair_rows rows) takes columns [0, stair_cols[g]),
  // non-decreasing in g; N = the widest group.  One launch for all groups; stair_n == 0: a plain rectangle / trapezoid.
  int stair_n = 0;
  int64_t stair_rows = 0;
  const int64_t* stair_cols = nullptr;       // host array
};
#define GH_GEMM_STAIR_MAX 128
int gh_launch_gemm(const GhGemm& g, hipStream_t st);
// Process-wide streams of a device, shared by every solver handle (gh_chol.hip): q[0] main (blocking,
// normal priority), q[1..3] non-blocking high-priority.  nullptr where creation failed.  Never destroyed.
// Why shared: HIP maps streams onto a few hardware queues; every further handle with streams of its
// own moved the others' onto different queues and a dense compute() at N <= 16384 ran 20-40 % slower
// with a second handle (or two application streams) alive (scripts/dev/queue_pattern.py) -- and a
// CU-masked stream takes ~1 s to create.  (Per-handle streams were an environment switch until round 4.)
bool gh_shared_streams(int device, hipStream_t q[4]);
// Once per device and process, BEFORE the library creates its first stream there: one empty kernel on the null stream
// and a device synchronisation.  Measured (scripts/dev/no_torch_step.py, stream_order_probe.py): when this library's
// streams are the first thing a process creates on the device -- any george user who does not import torch first -- the
// panel chain of every mid-size factorisation runs at half speed (N = 8192: 13.0 instead of 7.0 ms per
// compute()+log_likelihood(), N = 4096 4.5 instead of 2.7, N = 16384 40 instead of 30); once the null stream has had a
// launch first (what `torch.zeros(1, device="cuda")` happens to do) it does not.  The pairwise overlap of the streams
// is the same in both cases (gh_debug_stream_overlap), so it is not queue sharing; which hardware queue the runtime
// hands out first is not visible from here.  GEORGE_AMD_NO_NULL_PRIME=1 skips it (A/B).
void gh_prime_device(int device);
// measured when the process-wide streams of `device` were made: does the main stream share a dispatcher with the chain,
// rows-below or near stream?  (gh_chol.hip, factor_lookahead_deep: where a look-ahead factorisation is joined)
bool gh_shared_main_crowded(int device);
// do one-workgroup kernels of one stream complete while a grid far larger than the chip runs on the other, both ways round?
// (streams of the current device; ~8 ms)
bool gh_streams_dispatch_independently(hipStream_t a, hipStream_t b);
hipStream_t gh_shared_masked_stream(int device, int reserve_cus);
bool gh_use_mfma();               // false when GEORGE_AMD_NO_MFMA=1 (VALU validation path)

