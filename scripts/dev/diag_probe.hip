// The sixteen column steps of the 16x16 diagonal step (gh_potf2_body.h, diag16) timed by s_memtime:
// one wavefront alone, and wavefront 0 of a 256-thread workgroup whose other wavefronts wait at a barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "gh_potf2_body.h"
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k(double* out, long long* cyc, double a0) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  long long t0 = 0, t1 = 0;
  if (threadIdx.x < 64) {
    double v[16], w[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { v[q] = (q < i) ? a0 * 0.01 * (q + 1) : (q == i ? 2.0 + a0 : 0.0); w[q] = (q == i) ? 1.0 : 0.0; }
    int bad = -1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7" ::: "memory");
    t0 = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gh_potf2::diag16<MODE>(v, w);
    asm volatile("s_nop 7" ::: "memory");
    t1 = clock64();
    double sum = bad;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += v[q] + w[q];
    out[threadIdx.x] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; }
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 64);
  long long h = 0;
#define RUN(NT, MODE, what) do { for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k<NT, MODE>), dim3(1), dim3(NT), 0, 0, out, cyc, 1.0); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-70s %lld ticks\n", what, h); } while (0)
  RUN(64, 0, "one wavefront alone, full step");
  RUN(256, 0, "wavefront 0 of four (others at the barrier), full step");
  RUN(64, 2, "without the 240 DPP updates");
  RUN(64, 4, "without the 1/sqrt chains");
  RUN(64, 6, "neither (16 x 2 multiplications)");
  return 0;
}
