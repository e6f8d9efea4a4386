// Empirical lane-map probe for v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A lane x one-hot B lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0;
      const double b = (lane == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) out[la * 64 + lb] = lane;
    }
}
int main() {
  int* d; int h[4096];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0xff, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A%02d:", la);
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
