// Where do the ~80 us of one 128x128 potf2+inverse go?  Builds the library's own body
// (gh_potf2_body.h) with phase stamps (s_memrealtime, 10 ns ticks) and prints the mean per phase.
//   hipcc --offload-arch=gfx950 -O3 -I george_amd/csrc scripts/dev/potf2_phases.hip -o scripts/dev/potf2_phases
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
__device__ unsigned long long g_stamp[192];
#define GH_POTF2_STAMP(k) do { if (threadIdx.x == 0) { g_stamp[(k)] = wall_clock64(); g_stamp[64 + (k)] = clock64(); } } while (0)
#ifdef FINE
#define GH_POTF2_STAMP2(k) do { if (threadIdx.x == 0) { g_stamp[64 + (k)] = clock64(); } } while (0)
#endif
#ifdef V1
#include "gh_potf2_body_v1.h"
#else
#include "gh_potf2_body.h"
#endif
__global__ __launch_bounds__(256, 2) void k(double* A, long lda, double* dinv, long long* info) {
#ifdef V1
  __shared__ double s[128 * 129 / 2];
  __shared__ double inv16[8 * 16 * 17];
  __shared__ double rdiag[128];
  __shared__ int fail_at;
  (void)gh_potf2_v1::potf2_body<32>(A, lda, dinv, info, 0LL, s, inv16, rdiag, &fail_at);
#else
  __shared__ double s[GH_POTF2_S_DOUBLES];
  __shared__ double dscr[GH_POTF2_D_DOUBLES];
  __shared__ int fail_at;
  (void)gh_potf2::potf2_body(A, lda, dinv, info, 0LL, s, dscr, &fail_at);
#endif
}
int main() {
  const int T = 128; const long lda = 16384;
  std::vector<double> h(T * lda, 0.0);
  for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) h[i * lda + j] = exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.01 : 0.0);
  double *A, *D; long long* info;
  hipMalloc(&A, T * lda * 8); hipMalloc(&D, T * T * 8); hipMalloc(&info, 8); hipMemset(info, 0, 8);
  const int reps = 50; double acc[64] = {0};
  unsigned long long st[192];
  for (int r = 0; r < reps + 5; ++r) {
    hipMemcpy(A, h.data(), T * lda * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, A, lda, D, info);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    if (r < 5) continue;
    for (int q = 0; q < 64; ++q) acc[q] += (double)st[q] * 0.01;   // us
  }
  for (int q = 0; q < 64; ++q) acc[q] /= reps;
  long long inf; hipMemcpy(&inf, info, 8, hipMemcpyDeviceToHost);
  printf("info %lld\n", inf);
  {  // the factor and its inverse against a host Cholesky (long double accumulation)
    std::vector<double> L(T * T, 0.0), Li(T * T, 0.0), g(T * lda), gi(T * T);
    for (int j = 0; j < T; ++j) {
      long double d = h[j * lda + j];
      for (int k = 0; k < j; ++k) d -= (long double)L[j * T + k] * L[j * T + k];
      L[j * T + j] = (double)sqrtl(d);
      for (int i = j + 1; i < T; ++i) {
        long double v = h[i * lda + j];
        for (int k = 0; k < j; ++k) v -= (long double)L[i * T + k] * L[j * T + k];
        L[i * T + j] = (double)(v / L[j * T + j]);
      }
    }
    for (int c = 0; c < T; ++c)
      for (int i = c; i < T; ++i) {
        long double v = (i == c) ? 1.0L : 0.0L;
        for (int k = c; k < i; ++k) v -= (long double)L[i * T + k] * Li[k * T + c];
        Li[i * T + c] = (double)(v / L[i * T + i]);
      }
    hipMemcpy(g.data(), A, T * lda * 8, hipMemcpyDeviceToHost);
    hipMemcpy(gi.data(), D, T * T * 8, hipMemcpyDeviceToHost);
    double eL = 0, eI = 0, mL = 0, mI = 0, up = 0;
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) {
      if (j <= i) { eL = fmax(eL, fabs(g[i * lda + j] - L[i * T + j])); mL = fmax(mL, fabs(L[i * T + j]));
                    eI = fmax(eI, fabs(gi[i * T + j] - Li[i * T + j])); mI = fmax(mI, fabs(Li[i * T + j])); }
      else up = fmax(up, fmax(fabs(g[i * lda + j]), fabs(gi[i * T + j])));
    }
    printf("max |L - L_host| %.3e (max |L| %.3e)   max |Linv - host| %.3e (max %.3e)   max |upper| %.3e\n", eL, mL, eI, mI, up);
  }
  printf("load            %7.2f us\n", acc[1] - acc[0]);
  printf("phase 1         %7.2f us\n", acc[2] - acc[1]);
#ifdef V1
  double b = 0, c1 = 0, dg = 0, c2 = 0;
  for (int jb = 0; jb < 7; ++jb) {
    b += acc[11 + 4 * jb] - acc[10 + 4 * jb]; c1 += acc[12 + 4 * jb] - acc[11 + 4 * jb];
    dg += acc[13 + 4 * jb] - acc[12 + 4 * jb];
    double nxt = jb < 6 ? acc[10 + 4 * (jb + 1)] : acc[2];
    c2 += nxt - acc[12 + 4 * jb];
    printf("   step %d: rows %5.2f  col %5.2f  diag %5.2f  (diag||rest %5.2f)\n", jb, acc[11 + 4 * jb] - acc[10 + 4 * jb],
           acc[12 + 4 * jb] - acc[11 + 4 * jb], acc[13 + 4 * jb] - acc[12 + 4 * jb], nxt - acc[12 + 4 * jb]);
  }
  printf("   first diag   %7.2f us; rows %5.2f col %5.2f diag %5.2f diag||rest %5.2f\n", acc[10] - acc[1], b, c1, dg, c2);
#else
  for (int p = 0; p < 8; ++p) {
    double nxt = p < 7 ? acc[10 + 4 * (p + 1)] : acc[2];
    printf("   pass %d: wavefront 0: X + diagonal tile %5.2f  diagonal step %5.2f  wait at the barrier %5.2f\n", p,
           acc[11 + 4 * p] - acc[10 + 4 * p], acc[12 + 4 * p] - acc[11 + 4 * p], nxt - acc[12 + 4 * p]);
  }
#endif
  printf("factor -> HBM   %7.2f us\n", acc[3] - acc[2]);
  printf("8 diag inverses %7.2f us\n", acc[4] - acc[3]);
  printf("doubling        %7.2f us\n", acc[5] - acc[4]);
  printf("L^-1 -> HBM     %7.2f us\n", acc[6] - acc[5]);
  printf("total           %7.2f us\n", acc[6] - acc[0]);
#ifdef FINE
  printf("last diagonal step: start -> column 0 updates %llu ticks; column 15 done -> end %llu ticks; columns %llu ticks\n",
         st[64 + 48] - st[64 + 39], st[64 + 40] - st[64 + 79], st[64 + 79] - st[64 + 48]);
  for (int j = 0; j < 11; ++j) printf("column %2d: %d updates %llu ticks; to next column's updates %llu\n", j, 2 * (15 - j), st[64 + 49 + 2 * j] - st[64 + 48 + 2 * j], st[64 + 50 + 2 * j] - st[64 + 49 + 2 * j]);
#endif
  printf("s_memtime ticks: whole kernel %llu (%.0f per us); diagonal step of stage 3: %llu ticks\n", st[64 + 6] - st[64 + 0],
         (double)(st[64 + 6] - st[64 + 0]) / ((double)(st[6] - st[0]) * 0.01), st[64 + 12 + 16] - st[64 + 11 + 16]);
  return 0;
}
