// gh_gemm_tile.h -- device-side pieces of the fp64 MFMA GEMM family that more than one translation unit uses:
// the LDS-DMA operand path of gh_gemm.hip's kernels (XOR-swizzled slab images, the k assignment every k-major x k-major
// product shares) and, built from it, a one-tile product as a DEVICE FUNCTION (gh_tile128_nt_sp: the HODLR leaf stage's
// batched products, gh_hodlr.hip).
//
// A tile's bits must not depend on which kernel computed it: gh_tile128_nt_sp follows gemm_f64_mfma_dma_sp slab for slab and
// instruction for instruction; accumulators start from -C and the write-back is -acc (exact).  (gh_tile128_nt / gh_tile64_nt,
// the round-4 loop as device functions for the retired dataflow factorisation: scripts/dev/arms/dataflow_r05/.)
#pragma once
#include "gh_common.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 16
#define LS 18

// which k (of the 16 of a slab) lane group fk feeds into k-step kk of a k-major x k-major product: element offset inside the row's
// XOR-swizzled image.  k = {2 fk, 2 fk + 1, 2 fk + 8, 2 fk + 9}[kk]: the lane's four values sit in two 16-byte pieces, so the
// main kernel reads a slab's fragments with 16 ds_read_b128 per wavefront (round 4; the natural assignment k = 4 kk + fk needs
// 32 ds_read_b64: 68.0 -> 69.1 TFLOP/s on the SYRK shape, -1.3 ... -1.6 % on seven shapes, profiles/r04/gemm_pair_ab.md).
// EVERY k-major x k-major kernel must use the same assignment (a tile's bits must not depend on which of them computed it:
// tests compare schedules bit for bit); products with an m-major operand keep k = 4 kk + fk on both sides.
#define GH_KM_OFFK(kk, fk, sw) (((((fk) + 4 * ((kk) >> 1)) ^ (sw)) * 2) + ((kk) & 1))
typedef __attribute__((address_space(3))) void gh_lds_void;
typedef const __attribute__((address_space(1))) void gh_glb_void;

template <bool KM>
struct DmaOperand {
  // source of instruction i = ubase (wave-uniform: lives in SGPRs, advanced by a scalar add per slab) + voff[i] (this lane's byte
  // offset, loop-invariant).  Round 4: one 64-bit VGPR pointer per instruction (the first form) cost 8 v_lshl_add_u64 per slab
  // and wavefront plus a v_readfirstlane per instruction for the LDS address in M0 -- vector instructions that queue behind the
  // fp64 matrix instructions: 69.3 -> 70.3 TFLOP/s with scalar M0, -> 71.2 with the buffer form (profiles/r04/gemm_dma_addr_ab.md).
  const char* ubase;
  unsigned voff[4];
  long step;                // doubles to advance per slab
  int f0, f1;               // fragment read offsets (doubles), see frag()
  int offk[4];
  int off2[2];              // k-major x k-major launches: the lane's two 16-byte pieces, see frag2()

  __device__ __forceinline__ void init(const double* base, long ld, long r0, long kbeg, int wave, int lane, int wsub) {
    const int fr = lane & 15, fk = lane >> 4;
    if (KM) {
      ubase = (const char*)(base + r0 * ld + kbeg);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        voff[i] = (unsigned)(((long)r * ld + (((lane & 7) ^ ((r >> 1) & 7)) * 2)) * 8);
      }
      step = BK;
      const int sw = (fr >> 1) & 7;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) offk[kk] = (((kk * 2 + (fk >> 1)) ^ sw) * 2) + (fk & 1);
      off2[0] = ((fk ^ sw) * 2); off2[1] = (((fk + 4) ^ sw) * 2);
      f0 = (wsub * 64 + fr) * BK;
      f1 = 0;
    } else {
      ubase = (const char*)(base + kbeg * ld + r0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = wave * 4 + i;
        voff[i] = (unsigned)(((long)k * ld + ((lane ^ ((k & 1) << 3)) * 2)) * 8);
      }
      step = BK * ld;
      const int ix = fk & 1;
      f0 = fk * 128 + wsub * 64 + fr + 16 * ix;      // even 16-row groups
      f1 = fk * 128 + wsub * 64 + fr - 16 * ix;      // odd 16-row groups
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) offk[kk] = kk * 512;
    }
  }
  // fragment (k-step kk, 16-row group i) of the slab image at `s`
  __device__ __forceinline__ double frag(const double* s, int kk, int i) const {
    if (KM) return s[f0 + i * 16 * BK + offk[kk]];
    return s[((i & 1) ? f1 : f0) + i * 16 + offk[kk]];
  }
  // k-major image only: pieces fk and fk + 4 of row (16 i + fr) as ONE 16-byte LDS read each -- k-steps 2q and 2q + 1 of the lane.
  // The k index a lane feeds into k-step kk is then {2 fk, 2 fk + 1, 2 fk + 8, 2 fk + 9}[kk] instead of 4 kk + fk: any
  // assignment works as long as both operands use the same one (the instruction sums over its four k), and with this one a
  // slab costs a wavefront 16 ds_read_b128 instead of 32 ds_read_b64.
  __device__ __forceinline__ double2 frag2(const double* s, int q, int i) const {
    return *reinterpret_cast<const double2*>(s + f0 + i * 16 * BK + off2[q]);
  }
};

// buffer_load_dwordx4 ... offen lds: resource descriptor in SGPRs (base = the wave-uniform slab pointer, advanced by a scalar add),
// the lane's byte offset as the 32-bit voffset -- not one vector instruction per DMA.  (global_load_lds with an SGPR base: hipcc
// still forms a 64-bit vector address per instruction inside the loop, one v_lshl_add_u64 each.)
#define GH_DMA_ISSUE(op, sbuf)                                                                        \
  {                                                                                                   \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)op.ubase, 0, 0x7fffffff, 0x00020000); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (gh_lds_void*)((sbuf) + dst + i_ * 128), 16, (int)op.voff[i_], 0, 0, 0); \
    op.ubase += op.step * 8;                                                                          \
  }


// ---------------------------------------------------------------------------------------------
// The pieces of the half-slab software-pipelined slab loop (gemm_f64_mfma_dma_sp in gh_gemm.hip explains it): fragment reads as
// inline asm (hipcc would order them behind a vmcnt(0) once an LDS-DMA is in flight), hand-written counted waits that name the
// registers they guard, and one k-step of 16 matrix instructions.  LDS byte offsets are immediates: buffers A0 | A1 | B0 | B1
// of 16 KiB each behind ONE base address.
typedef double v2d __attribute__((ext_vector_type(2)));
#define GH_SP_READ8(x0, x1, x2, x3, y0, y1, y2, y3, pa, pb, BO)                                                     \
  asm volatile("ds_read_b128 %0, %8 offset:%10\n\tds_read_b128 %1, %8 offset:%11\n\t"                               \
               "ds_read_b128 %2, %8 offset:%12\n\tds_read_b128 %3, %8 offset:%13\n\t"                               \
               "ds_read_b128 %4, %9 offset:%10\n\tds_read_b128 %5, %9 offset:%11\n\t"                               \
               "ds_read_b128 %6, %9 offset:%12\n\tds_read_b128 %7, %9 offset:%13"                                   \
               : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)               \
               : "v"(pa), "v"(pb), "n"((BO)), "n"((BO) + 2048), "n"((BO) + 4096), "n"((BO) + 6144))
#define GH_SP_WAIT(cnt, x0, x1, x2, x3, y0, y1, y2, y3)                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                        \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3))
#define GH_SP_MFMA(x, y, c)                                                                                         \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                \
      acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i_][c], y[j_][c], acc[i_][j_], 0, 0, 0);

// ---------------------------------------------------------------------------------------------
// One 128 x 128 tile on the calling 256-thread workgroup:  ACC: C -= A B^T (accumulators start from -C, store-only
// write-back), else C = A B^T (C may be A: every slab of A has been read when the first element of C is stored).
// A, B: 128 x K, k contiguous (lda, ldb even, bases 16-byte aligned), K a multiple of 32.  sm: 8192 doubles of LDS, 1 KiB
// aligned.  Every wavefront must have passed a barrier since its last read of sm; on return all stores have been ISSUED
// (not waited for) and every wavefront has passed the loop's last barrier.
// gh_tile128_nt_sp: the half-slab pipelined loop of gemm_f64_mfma_dma_sp (same k-steps in the same order: same bits).
// THE INVARIANT (enforced by the build: check_kernels.py): a kernel that inlines this function must not use
// scratch memory.  An inline-asm LDS read's destination is, to the compiler, written when the asm is issued -- under register
// pressure it stores that value to scratch at once (tests/kernel_gate/spilled_variant.hip shows exactly that: scratch_store of
// the destinations in front of the s_waitcnt) while the hardware still owes the register its data.  Round 5 met this as a
// `Memory access fault` when the function was inlined into the spilling workers of the (since retired) dataflow factorisation.
template <bool ACC>
__device__ __forceinline__ void gh_tile128_nt_sp(double* sm, double* C, long ldc, const double* A, long lda,
                                              const double* B, long ldb, long K) {
  double* const sA0 = sm; double* const sA1 = sm + BM * BK;
  double* const sB0 = sm + 2 * BM * BK; double* const sB1 = sm + 2 * BM * BK + BN * BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  v4d acc[4][4];
  const long nk = K / BK;
  DmaOperand<true> oa, ob;
  oa.init(A, lda, 0, 0, wave, lane, wm);
  ob.init(B, ldb, 0, 0, wave, lane, wn);
  const int dst = wave * 4 * 128;
  GH_DMA_ISSUE(oa, sA0) GH_DMA_ISSUE(ob, sB0)
  double* const cbase = C + (long)(wm * 64 + fk) * ldc + wn * 64 + fr;
  if (ACC) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j][r] = -1.0 * cbase[(long)(i * 16 + 4 * r) * ldc + j * 16];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();
  {
    // the half-slab pipelined loop of gemm_f64_mfma_dma_sp (round 5; an EVEN number of slabs): same k-steps in the same order, same bits
    const unsigned lbase = (unsigned)(unsigned long)(gh_lds_void*)sm;
    const unsigned pa0 = lbase + (unsigned)(oa.f0 + oa.off2[0]) * 8u, pa1 = lbase + (unsigned)(oa.f0 + oa.off2[1]) * 8u;
    const unsigned pb0 = lbase + 2u * BM * BK * 8u + (unsigned)(ob.f0 + ob.off2[0]) * 8u;
    const unsigned pb1 = lbase + 2u * BM * BK * 8u + (unsigned)(ob.f0 + ob.off2[1]) * 8u;
    constexpr int B1 = BM * BK * 8;
    v2d a01[4], b01[4], a23[4], b23[4];
    GH_DMA_ISSUE(oa, sA1) GH_DMA_ISSUE(ob, sB1)
    GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, 0);
    for (long kt = 0; kt < nk; kt += 2) {
      const bool more = kt + 2 < nk;
      GH_SP_READ8(a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3], pa1, pb1, 0);
      GH_SP_WAIT(8, a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3]);
      GH_SP_MFMA(a01, b01, 0) GH_SP_MFMA(a01, b01, 1)
      __builtin_amdgcn_sched_barrier(0);
      GH_SP_WAIT(0, a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3]);
      __syncthreads();
      if (more) { GH_DMA_ISSUE(oa, sA0) GH_DMA_ISSUE(ob, sB0) }
      GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, B1);
      __builtin_amdgcn_sched_barrier(0);
      GH_SP_MFMA(a23, b23, 0) GH_SP_MFMA(a23, b23, 1)
      __builtin_amdgcn_sched_barrier(0);
      GH_SP_READ8(a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3], pa1, pb1, B1);
      GH_SP_WAIT(8, a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3]);
      GH_SP_MFMA(a01, b01, 0) GH_SP_MFMA(a01, b01, 1)
      __builtin_amdgcn_sched_barrier(0);
      GH_SP_WAIT(0, a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3]);
      // (also behind the LAST slab: the contract -- every wavefront has passed a barrier after its last read of sm --
      //  is what lets the caller's next tile start its DMA at once)
      __syncthreads();
      if (more) {
        if (kt + 3 < nk) { GH_DMA_ISSUE(oa, sA1) GH_DMA_ISSUE(ob, sB1) }
        GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      GH_SP_MFMA(a23, b23, 0) GH_SP_MFMA(a23, b23, 1)
      __builtin_amdgcn_sched_barrier(0);
    }
    // (a loop that never ran -- K = 0 -- would leave the prologue's eight reads outstanding while the epilogue reuses their
    //  registers; after a loop that did run nothing is outstanding and this costs nothing.  Found by check_kernels.py.)
    GH_SP_WAIT(0, a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3]);
  }
  const double alpha = ACC ? -1.0 : 1.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* const crow = cbase + (long)(i * 16 + 4 * r) * ldc;
#pragma unroll
      for (int j = 0; j < 4; ++j) crow[j * 16] = alpha * acc[i][j][r];
    }
}
