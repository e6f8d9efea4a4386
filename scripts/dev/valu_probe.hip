// Issue / latency cost of the instruction kinds of the 16x16 diagonal step, one wavefront, s_memtime.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/valu_probe.hip -o scripts/dev/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c0 = 1.0, c1 = 2.0, c2 = 3.0, c3 = 4.0, c4 = 5.0, c5 = 6.0, c6 = 7.0, c7 = 8.0;
  long long t[24]; int n = 0;
  const long long w0 = wall_clock64();
#define TICK() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory"); t[n++] = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
  TICK();
  // T1: 256 independent v_fma_f64 (8 accumulators)
  asm volatile(".rept 32\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
               "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  // T2: 256 dependent v_fma_f64
  asm volatile(".rept 256\n v_fma_f64 %0, %1, %2, %0\n .endr" : "+v"(c0) : "v"(a), "v"(b));
  TICK();
  // T3: 256 v_readlane_b32, results unused
  asm volatile(".rept 32\n v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %0, 6\n"
               "v_readlane_b32 s24, %0, 7\n v_readlane_b32 s25, %0, 8\n v_readlane_b32 s26, %0, 9\n v_readlane_b32 s27, %0, 10\n .endr"
               :: "v"(__double2loint(a)) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  TICK();
  // T4: 128 x (2 readlane, s_nop 1, fma with the SGPR pair), 4 independent accumulators
  asm volatile(".rept 32\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n s_nop 1\n v_fma_f64 %0, %6, s[20:21], %0\n"
               "v_readlane_b32 s22, %4, 4\n v_readlane_b32 s23, %5, 4\n s_nop 1\n v_fma_f64 %1, %6, s[22:23], %1\n"
               "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 5\n s_nop 1\n v_fma_f64 %2, %6, s[24:25], %2\n"
               "v_readlane_b32 s26, %4, 6\n v_readlane_b32 s27, %5, 6\n s_nop 1\n v_fma_f64 %3, %6, s[26:27], %3\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b)
               : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  TICK();
  // T5: 64 x the 1/sqrt chain (rsq, mul, fma, mul || fma, fma), each chain feeding the next
  asm volatile(".rept 64\n v_rsq_f64 %1, %0\n v_mul_f64 %2, %0, 0.5\n v_mul_f64 %2, %2, %1\n v_fma_f64 %2, -%2, %1, 0.5\n v_mul_f64 %3, %1, %2\n"
               "v_fma_f64 %2, %2, %4, 1.0\n v_fma_f64 %0, %3, %2, %1\n .endr"
               : "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(b));
  TICK();
  // T6: 256 dependent v_mul_f64
  asm volatile(".rept 256\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(c1) : "v"(b));
  TICK();
  // T7: 256 v_readlane pairs each followed by a dependent use through the SAME SGPR pair (the compiler's first form)
  asm volatile(".rept 128\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s21, %2, 3\n s_nop 1\n v_fma_f64 %0, %3, s[20:21], %0\n .endr"
               : "+v"(c2) : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b) : "s20", "s21");
  TICK();
  // T8: 256 independent v_mul_f64 (4 registers)
  asm volatile(".rept 64\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n .endr"
               : "+v"(c0), "+v"(c3), "+v"(c5), "+v"(c6) : "v"(b));
  TICK();
  // T9: 256 v_add_f32 dependent (reference point for a plain 32-bit VALU op)
  { float f = (float)a; asm volatile(".rept 256\n v_add_f32 %0, %0, %0\n .endr" : "+v"(f)); c7 += f; }
  TICK();
  // T10: 64 x v_rsq_f64 dependent on itself
  asm volatile(".rept 64\n v_rsq_f64 %0, %0\n .endr" : "+v"(c4));
  TICK();
  // T11: 256 v_fmac_f64_dpp row_newbcast, 8 accumulators
  asm volatile("s_nop 1\n .rept 32\n v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %4, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %6, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  // T12: 256 dependent v_fmac_f64_dpp (the accumulator is also the broadcast source of the next one)
  asm volatile("s_nop 1\n .rept 256\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n .endr" : "+v"(c0) : "v"(b));
  TICK();
  // T13: 64 x v_mov_b64_dpp
  asm volatile("s_nop 1\n .rept 64\n v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n .endr" : "+v"(c1) : "v"(a));
  TICK();
  // T14: 64 x 4 dependent MFMA f64 16x16x4 (one accumulator)
  { typedef double v4d __attribute__((ext_vector_type(4))); v4d acc = {c0, c1, c2, c3};
    asm volatile(".rept 256\n v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n .endr\n s_nop 7\n s_nop 7" : "+v"(acc) : "v"(a), "v"(b));
    TICK();
    v4d acc2 = {c4, c5, c6, c7};
    // T15: 256 MFMAs, two accumulators alternating
    asm volatile(".rept 128\n v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1\n .endr\n s_nop 7\n s_nop 7" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b));
    TICK();
    c0 += acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + acc2[2] + acc2[3]; }
  // T16: as T11 with the negated src1
  asm volatile("s_nop 1\n .rept 32\n v_fmac_f64_dpp %0, %8, -%9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, -%9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %2, %8, -%9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, -%9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %4, %8, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, -%9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %6, %8, -%9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, -%9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  // T17: src0 == src1
  asm volatile("s_nop 1\n .rept 32\n v_fmac_f64_dpp %0, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %2, %8, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %4, %8, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %6, %8, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %8 row_newbcast:10 row_mask:0xf bank_mask:0xf\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  // T18: 256 plain v_fmac_f64 (VOP2, no DPP), 8 accumulators
  asm volatile(".rept 32\n v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
               "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  // T19: 32 distinct accumulator-free pattern: fmac_dpp alternating two sources like the diagonal step (v / w)
  asm volatile("s_nop 1\n .rept 32\n v_fmac_f64_dpp %0, %8, -%8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, -%9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %2, %8, -%8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, -%9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %4, %8, -%8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, -%9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
               "v_fmac_f64_dpp %6, %8, -%8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, -%9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n .endr"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  TICK();
  const long long w1 = wall_clock64();
  out[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0) { for (int q = 0; q < n; ++q) cyc[q] = t[q]; cyc[30] = w1 - w0; cyc[31] = t[n - 1] - t[0]; }
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8);
  long long h[32];
  const char* name[] = {"256 independent v_fma_f64", "256 dependent v_fma_f64", "256 v_readlane_b32", "128 x (2 readlane, s_nop 1, fma) 4 accumulators",
                        "64 x 1/sqrt chain (7 instr)", "256 dependent v_mul_f64", "128 x (2 readlane, s_nop 1, fma) 1 accumulator", "256 independent v_mul_f64",
                        "256 dependent v_add_f32", "64 dependent v_rsq_f64", "256 v_fmac_f64_dpp row_newbcast (8 acc)", "256 dependent v_fmac_f64_dpp", "64 v_mov_b64_dpp", "256 dependent MFMA f64 16x16x4", "256 MFMA f64 16x16x4, 2 accumulators", "256 fmac_dpp, negated src1", "256 fmac_dpp, src0 == src1", "256 v_fmac_f64 (no DPP)", "256 fmac_dpp in the v/w pattern of the diagonal step"};
  const int cnt[] = {256, 256, 256, 128, 64, 256, 128, 256, 256, 64, 256, 256, 64, 256, 256, 256, 256, 256, 256};
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  }
  for (int q = 0; q < 19; ++q) printf("%-52s %7lld ticks  %6.1f per item\n", name[q], h[q + 1] - h[q], (double)(h[q + 1] - h[q]) / cnt[q]);
  printf("s_memtime ticks per microsecond: %.1f (wall clock 100 MHz)\n", (double)h[31] / ((double)h[30] * 0.01));
  return 0;
}
