// TEST INPUT for tests/test_kernel_gate.py -- never linked into the product.
// The product's half-slab pipelined tile (gh_tile128_nt_sp: inline-asm ds_read_b128 + hand-counted s_waitcnt) forced
// into 128 VGPRs (launch bounds of 1024 threads), where its 128 accumulator registers alone fill the file: the compiler must spill, and a spilled or
// copied destination of an inline-asm LDS read is exactly what george_amd/csrc/check_kernels.py has to refuse.
#include "../../george_amd/csrc/gh_gemm_tile.h"

__global__ __launch_bounds__(1024)
void spilled_sp_kernel(double* C, const double* A, const double* B, long K) {
  extern __shared__ __align__(1024) double sm[];
  gh_tile128_nt_sp<true>(sm, C, 128, A, K, B, K, K);
}
