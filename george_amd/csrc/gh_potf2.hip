// gh_potf2.hip -- the critical-path kernel of the blocked factorisation: one workgroup takes a
// 128x128 diagonal block to its Cholesky factor L AND to L^-1 (which turns every TRSM of the
// solver into an MFMA GEMM).  Also the leaf kernel of the HODLR solver (batched: blockIdx.x).
//
// The lower triangle of the block lives in LDS in PACKED row-major form (128*129/2 doubles =
// 64.5 KiB; 76 KiB with the eight packed 16x16 diagonal inverses and the spare slots).  Packing is
// what lets this kernel be PLACED beside the trailing SYRK during look-ahead: a SYRK workgroup holds
// 64 KiB of LDS, so a CU with one SYRK workgroup still has room for this one, whereas the first,
// unpacked 129-KiB tile had to wait for a whole CU to drain (1.7 ms average under SYRK).  Placed is
// not the same as fast: beside SYRK wavefronts the kernel runs several times slower, which is why
// small matrices keep 32 CUs free of SYRK work (gh_chol.hip, trailing_stream).
//
// The algorithm is in gh_potf2_body.h: 33 us per block -- DPP-broadcast diagonal step that yields the
// 16x16 inverses for free, one barrier per 16-column step with X^T tiles recomputed into MFMA operand
// registers, register-chained doubling products.  (Its 82-us predecessor -- one-wavefront diagonal step
// by v_readlane broadcasts, per-row substitution, separate 16x16 inverses, doubling through an LDS
// scratch -- is retired: scripts/dev/arms/gh_potf2_body_v1.h.)
// The very first version (scalar rank-1 updates, 3 barriers per column, column-wise inverse) took
// 553 us per block and was half of compute() at N = 16384; it is kept in gh_chol.hip
// (`potf2_inv_kernel`) as the validation arm GEORGE_AMD_POTF2=simple.
#include <stdlib.h>
#include <string.h>
#include "gh_potf2_body.h"

// blockIdx.x selects the block of a batch (stride_a / stride_d doubles apart; 0 for the single
// diagonal block of the dense factorisation): the HODLR leaves are factored and inverted this way.
// 75 KB of LDS: two workgroups per CU in a batched launch, and room beside a 64-KB SYRK workgroup.
__global__ __launch_bounds__(256, 2) void potf2_inv_mfma_kernel(double* A, long lda, double* dinv,
                                                             long long* info, long long base,
                                                             long stride_a, long stride_d) {
  A += (long)blockIdx.x * stride_a;
  dinv += (long)blockIdx.x * stride_d;
  __shared__ double s[GH_POTF2_S_DOUBLES];
  __shared__ double dscr[GH_POTF2_D_DOUBLES];   // the eight 16x16 diagonal inverses, packed
  __shared__ int fail_at;
  (void)gh_potf2::potf2_body(A, lda, dinv, info, base, s, dscr, &fail_at);
}
int gh_launch_potf2_batched(double* A, int64_t lda, int64_t stride_a, double* dinv, int64_t stride_d, long long* info,
                            int nbatch, hipStream_t st) {
  if (nbatch <= 0) return GH_OK;
  hipLaunchKernelGGL(potf2_inv_mfma_kernel, dim3((unsigned)nbatch), dim3(256), 0, st,
                     A, (long)lda, dinv, info, 0LL, (long)stride_a, (long)stride_d);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

int gh_launch_potf2_mfma(double* A, int64_t lda, double* dinv, long long* info, long long base, hipStream_t st) {
  hipLaunchKernelGGL(potf2_inv_mfma_kernel, dim3(1), dim3(256), 0, st, A, (long)lda, dinv, info,
                     base, 0L, 0L);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The HODLR leaf kernel (round 6): block b of a batch -> its INVERSE K_b^-1 = L^-T L^-1 (full symmetric 128 x 128, in place) and
// log|K_b|, in ONE launch.  The HODLR solver needs nothing else of a leaf; as four launches -- this factorisation with L and L^-1
// written out in full (zeros included), a read-back of L's diagonal for the log-determinant, a batched transpose of L^-1 and a batched
// product -- the 2048 leaves of C4 moved 1.8 GB through the L2s (profiles/r06/traffic_C4_N262144.json: 0.69 + 0.03 + 0.54 + 0.54)
// and held CUs for 0.27 ms of stream time in a phase that is bound by CU-time.  Here L^-1 never leaves LDS: potf2_body<false>
// stops with its packed lower triangle in s; W = L^-T L^-1, W(ti, tj) = sum_{tk >= ti} Linv(tk, ti)^T Linv(tk, tj) over 16 x 16 tiles
// (ti >= tj), is 480 matrix instructions shared by the four wavefronts (~4 us); log|K_b| = -2 sum_i log Linv(i, i).
// KERN: the block is not read from A but evaluated from the leaf's points (GhPotf2Kern): the HODLR leaf build -- a launch that wrote
// 268 MB for the next one to read back -- disappears for kernels of the a + b F(r^2) form.
struct GhLeafRange { int start, size; long off; };       // (= LeafDesc of gh_hodlr.hip)
template <bool KERN>
__global__ __launch_bounds__(256, 2) void potf2_kinv_kernel(double* A, long lda, long stride_a, double* logdet, long long* info,
                                                            GhFast fast, const double* x, const double* yerr, int nd, const GhLeafRange* leaves) {
  using namespace gh_potf2;
  A += (long)blockIdx.x * stride_a;
  __shared__ double s[GH_POTF2_S_DOUBLES];
  __shared__ double dscr[GH_POTF2_D_DOUBLES];
  __shared__ int fail_at;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool ok;
  if (KERN) {
    const GhLeafRange lf = leaves[blockIdx.x];
    const GhPotf2Kern src{fast, x + (long)lf.start * nd, yerr + lf.start, nd, lf.size};
    ok = potf2_body<false, GhPotf2Kern>(A, lda, nullptr, info, (long long)blockIdx.x * 128, s, dscr, &fail_at, src);
  } else {
    ok = potf2_body<false>(A, lda, nullptr, info, (long long)blockIdx.x * 128, s, dscr, &fail_at);
  }
  if (!ok) {
    if (tid == 0) logdet[blockIdx.x] = 0.0;
    return;
  }
  __syncthreads();
  if (wave == 3) {                               // log|K| = 2 sum log L_ii = -2 sum log Linv_ii (fixed order: reproducible)
    double v = log(s[rowbase(lane) + lane]) + log(s[rowbase(lane + 64) + lane + 64]);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) logdet[blockIdx.x] = -2.0 * v;
  }
  const int fr = lane & 15, fq = lane >> 4;
  int e = 0;
  for (int ti = 7; ti >= 0; --ti)                // (tile rows with the most k-tiles last: the short tiles fill the tail)
    for (int tj = 0; tj <= ti; ++tj, ++e) {
      if ((e & 3) != wave) continue;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int tk = ti; tk < 8; ++tk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int R = 16 * tk + 4 * q + fq;    // the k index of this lane: row R of L^-1
          const int ca = 16 * ti + fr, cb = 16 * tj + fr;
          const double a = s[rowbase(R) + ca], b = s[rowbase(R) + cb];       // (a column beyond the diagonal: the next row's data, masked)
          acc = mma(ca <= R ? a : 0.0, cb <= R ? b : 0.0, acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + fq + 4 * r, j = 16 * tj + fr;
        A[(long)i * lda + j] = acc[r];
        if (ti != tj) A[(long)j * lda + i] = acc[r];
      }
    }
}
// (every block's slot is read in full before it is overwritten: the loads of potf2_body precede its first barrier)
int gh_launch_potf2_kinv_batched(double* A, int64_t lda, int64_t stride_a, double* logdet, long long* info, int nbatch, hipStream_t st) {
  if (nbatch <= 0) return GH_OK;
  GhFast none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(potf2_kinv_kernel<false>, dim3((unsigned)nbatch), dim3(256), 0, st, A, (long)lda, (long)stride_a, logdet, info,
                     none, (const double*)nullptr, (const double*)nullptr, 0, (const GhLeafRange*)nullptr);
  GH_HIP(hipGetLastError());
  return GH_OK;
}
// ... with the blocks evaluated inside the kernel: block b = K(x[leaves[b].start ...], same) + diag(yerr^2), identity-padded to 128
// (`leaves`: device array of {start, size, off}; fast.ok required)
int gh_launch_potf2_kinv_kernel_batched(double* A, int64_t lda, int64_t stride_a, double* logdet, long long* info, int nbatch,
                                        const GhFast& fast, const double* x, const double* yerr, int nd, const void* leaves, hipStream_t st) {
  if (nbatch <= 0) return GH_OK;
  hipLaunchKernelGGL(potf2_kinv_kernel<true>, dim3((unsigned)nbatch), dim3(256), 0, st, A, (long)lda, (long)stride_a, logdet, info,
                     fast, x, yerr, nd, (const GhLeafRange*)leaves);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

