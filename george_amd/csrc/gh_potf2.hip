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
