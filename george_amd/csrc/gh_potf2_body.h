// gh_potf2_body.h -- the 128x128 Cholesky + inverse of gh_potf2.hip as a device function (second form;
// the first one, 82 us per block, is retired: scripts/dev/arms/gh_potf2_body_v1.h).
//
// Where the 82 us of the first form went (scripts/dev/potf2_phases.hip, profiles/r02/potf2_phases_v1.txt):
// 8 x 3.3 us for the one-wavefront 16x16 diagonal steps, 7 x 2.2 us for row substitution + tile column
// + their barriers, 6.5 us for the eight 16x16 inverses, 19.4 us for the doubling products (two
// barrier-separated products per chunk through an LDS scratch), 14 us of HBM load / store phases run
// to completion one after the other.  This form removes whole phases instead of tuning them:
//
//  * the 16x16 diagonal step produces the block's INVERSE in the same instruction stream: lane i
//    (of every 16-lane row) holds row i of the block AND column i of D^-1, and the recurrence
//    `v[j] *= 1/sqrt(d);  v[k] -= L_kj v[j]` is the Cholesky column step for the former and forward
//    substitution for the latter, with the SAME multipliers L_kj -- broadcast INSIDE the FMA by DPP
//    row_newbcast (v_fmac_f64_dpp), not through SGPRs (v_readlane + hazard + FMA: 22 cycles per
//    update against 5).  No separate inverse phase, and the rows below become X = A D^-T on the
//    matrix pipe instead of a 16-deep dependent substitution per thread;
//  * one barrier per 16-column step instead of three: every wavefront recomputes the X^T tiles it
//    needs (4 MFMAs each) straight into MFMA operand registers -- the f64 16x16x4 result of
//    X^T = D^-1 A^T has exactly the lane layout both operands of C -= X X^T want, if the k index of
//    that product is taken as (lane >> 4) + 4 r -- and wavefront 0 runs the critical chain
//    (X of the next tile row, next diagonal tile, next diagonal step) on its own while wavefronts
//    1-3 update everything else (straight-line code per wavefront, software-pipelined by hand);
//  * 1/sqrt(d) from v_rsq_f64 and ONE third-order step (4 dependent operations instead of 12);
//  * the doubling products X = -B^-1 (C A^-1) chained through registers the same way (the result
//    tile column of C A^-1 is the B operand of the second product): no scratch, one barrier between
//    reading C and overwriting it;
//  * HBM phases overlapped: wavefront 0 fetches the first diagonal tile itself and starts, the others
//    fetch the rest of the lower triangle with all their loads in flight; the zeros of the upper-right
//    quadrant are stored by wavefronts 1-3 a chunk per step; the factor is stored without waiting
//    (phase 2 runs under the stores).
// What the instructions cost on this chip (scripts/dev/valu_probe.hip): any instruction of a wavefront
// issues every ~4.4 cycles, a dependent f64 FMA every 4.9 -- nothing to hide, only instructions to
// remove -- and an f64 MFMA holds the SIMD for 64.5 cycles during which the wavefront issues no other
// vector instruction.
#pragma once
#include "gh_common.h"
#include <type_traits>

#ifndef GH_POTF2_BODY_H_
#define GH_POTF2_BODY_H_
// phase time stamps for scripts/dev/potf2_phases.hip; nothing in the library build
#ifndef GH_POTF2_STAMP
#define GH_POTF2_STAMP(k)
#endif
#ifndef GH_POTF2_STAMP2
#define GH_POTF2_STAMP2(k)
#endif

#define GH_POTF2_S_DOUBLES (128 * 129 / 2)      // packed lower triangle of the block
#define GH_POTF2_D_DOUBLES (8 * 136 + 64 + 136) // packed lower triangles of the eight 16x16 diagonal inverses, 64 spare slots, a dummy tile

namespace gh_potf2 {
typedef double v4d __attribute__((ext_vector_type(4)));

// offset of row i in the packed row-major lower triangle
__device__ __forceinline__ int rowbase(int i) { return (i * (i + 1)) >> 1; }
// the same without a 32-bit integer multiply (quarter rate) in the lane-dependent part: u < 16 is the
// lane's offset inside a 16-row tile, rbu = rowbase16(u), t the (wave-uniform) tile row:
// rowbase(16 t + u) = rowbase(u) + t (16 u + 8) + 128 t^2
__device__ __forceinline__ int rowbase16(int u) { return __mul24(u, u + 1) >> 1; }
__device__ __forceinline__ int rowbase_t(int t, int u, int rbu) { return rbu + __mul24(16 * u + 8, t) + 128 * t * t; }

// value of `v` in lane `src` (a compile-time constant after unrolling), delivered through SGPRs
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ v4d mma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

#define GH_SB() __builtin_amdgcn_sched_barrier(0)

// tile slot e = ui (ui + 1) / 2 - 1 + uj (ui = 1..6, uj <= ui) of the 27 tiles a step updates below its diagonal tile
__host__ __device__ constexpr int slot_ui(int e) { int ui = 1; while ((ui + 1) * (ui + 2) / 2 - 1 <= e) ++ui; return e < 0 ? 1 : ui; }
__host__ __device__ constexpr int slot_uj(int e) { return e < 0 ? 0 : e - (slot_ui(e) * (slot_ui(e) + 1) / 2 - 1); }

// The sixteen column steps of the register-resident 16x16 diagonal step (see potf2_body, phase 1 (a)).
// v[k]: row `lane & 15` of the block; w[k]: column `lane & 15` of its inverse (e_i on entry).
// No pivot test in here: a non-positive (or NaN) pivot makes 1/sqrt NaN and poisons every later
// column, so L(15,15) > 0 afterwards says that all sixteen were fine.
// MODE (scripts/dev/diag_probe.hip only): 2 = without the DPP updates, 4 = without the 1/sqrt chains.
#define GH_DPP_UPD(acc, src, mul, k) \
  asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul))
#define GH_UPDK(j, k) do { if ((k) > (j)) { GH_DPP_UPD(v[k], v[j], v[j], k); GH_DPP_UPD(w[k], v[j], w[j], k); } } while (0)
#define GH_COL(j) do { \
    double y_; \
    if (!(MODE & 4)) { \
      double d_; \
      asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #j " row_mask:0xf bank_mask:0xf" : "=v"(d_) : "v"(v[j])); \
      /* 1/sqrt(d): v_rsq_f64 seed y0 (relative error e ~ 2^-23 or better) and ONE third-order step     */ \
      /* y0 (1 + e + 3/2 e^2), e = (1 - d y0^2) / 2: error O(e^3) < 2^-60; with e2 = 2 e = 1 - d y0^2 the */ \
      /* bracket is 1 + e2 (1/2 + 3/8 e2): rsq + 4 dependent operations                                  */ \
      const double y0_ = __builtin_amdgcn_rsq(d_); \
      const double e2_ = fma(-(d_ * y0_), y0_, 1.0); \
      y_ = fma(y0_ * e2_, fma(0.375, e2_, 0.5), y0_); \
    } else { y_ = 0.99; } \
    v[j] *= y_; w[j] *= y_; \
    asm volatile("s_nop 1" ::: "memory");          /* VALU write -> DPP read of v[j]: two wait states */ \
    GH_POTF2_STAMP2(48 + 2 * (j)); \
    if (!(MODE & 2)) { \
    GH_UPDK(j, 1); GH_UPDK(j, 2); GH_UPDK(j, 3); GH_UPDK(j, 4); GH_UPDK(j, 5); GH_UPDK(j, 6); GH_UPDK(j, 7); GH_UPDK(j, 8); \
    GH_UPDK(j, 9); GH_UPDK(j, 10); GH_UPDK(j, 11); GH_UPDK(j, 12); GH_UPDK(j, 13); GH_UPDK(j, 14); GH_UPDK(j, 15); } \
    GH_POTF2_STAMP2(49 + 2 * (j)); \
  } while (0)
template <int MODE = 0>
__device__ __forceinline__ void diag16(double (&v)[16], double (&w)[16]) {
  GH_COL(0); GH_COL(1); GH_COL(2); GH_COL(3); GH_COL(4); GH_COL(5); GH_COL(6); GH_COL(7);
  GH_COL(8); GH_COL(9); GH_COL(10); GH_COL(11); GH_COL(12); GH_COL(13); GH_COL(14); GH_COL(15);
}
#undef GH_COL
#undef GH_UPDK
#undef GH_DPP_UPD

// s: GH_POTF2_S_DOUBLES, dscr: GH_POTF2_D_DOUBLES doubles, fail_at_p: one int -- all LDS.
// Returns false when the block is not positive definite (then *info is set) or an earlier one was not.
// STORE = false (the HODLR leaf kernel, potf2_kinv_kernel): nothing is written to A or dinv -- on return the packed lower
// triangle of L^-1 is in s, behind a workgroup barrier, and the caller goes on from there.
// SRC: where the block comes from.  The default reads A (row pitch lda); GhPotf2Kern (the HODLR leaf kernel) EVALUATES it --
// K(x_c, x_r) + yerr_r^2 on the diagonal through the a + b F(r^2) fast form, identity padding beyond `size` -- so that the leaves'
// covariance blocks are never written to memory at all.
struct GhPotf2Mem {
  static constexpr bool kLoadsInFlight = true;  // all 59 loads of a lane issued before the first is waited for
  const double* A; long lda;
  // element (row i, column c <= i); i wave-uniform in the bulk loads: the row address is SGPR arithmetic, the lane adds 32 bits
  __device__ __forceinline__ double operator()(int i, int c) const { return *(const double*)((const char*)(A + (long)i * lda) + (unsigned)c * 8u); }
};
struct GhPotf2Kern {
  static constexpr bool kLoadsInFlight = false;
  GhFast fast; const double* x; const double* yerr; int nd, size;       // x, yerr: the leaf's first point
  __device__ __forceinline__ double operator()(int i, int c) const {
    if (i >= size || c >= size) return i == c ? 1.0 : 0.0;
    double v = gh_fast_value(fast, x + (long)c * nd, x + (long)i * nd);   // ordered arguments (x_min, x_max): c <= i
    if (i == c) { const double e = yerr[i]; v += e * e; }
    return v;
  }
};
template <bool STORE = true, class SRC = GhPotf2Mem>
__device__ __forceinline__ bool potf2_body(double* A, long lda, double* dinv, long long* info, long long base,
                                           double* s, double* dscr, int* fail_at_p, const SRC& src) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (an SGPR: everything derived from it is scalar)
  const int fr = lane & 15, fq = lane >> 4;     // MFMA operand row / k sub-index of this lane
  const int rbfr = rowbase16(fr);
  int uq[4], rbuq[4];                           // fq + 4 r: row of MFMA result register r inside a tile
#pragma unroll
  for (int r = 0; r < 4; ++r) { uq[r] = fq + 4 * r; rbuq[r] = rowbase16(uq[r]); }
  // latency-bound chain of dependent steps that, under look-ahead, shares its CU with wavefronts
  // of the trailing SYRK issuing 64 MFMAs back to back: take the instruction arbiter's top priority
  __builtin_amdgcn_s_setprio(3);
  GH_POTF2_STAMP(0);
  const long long info_in = *info;              // (looked at after the loads have been issued)
  if (tid == 0) (*fail_at_p) = -1;
  // ================================================================ phase 1: Cholesky
  // (a) diagonal block jb by wavefront 0, register-resident.  Lane i (of every 16-lane row: the
  //     four rows work redundantly) holds row i of the block in v[0..15] AND column i of the block's
  //     inverse in w[0..15] (starts as e_i).  Column step j:  v[j] *= 1/sqrt(d_j), w[j] *= 1/sqrt(d_j),
  //     then for k > j:  v[k] -= L_kj v[j]  (Cholesky)  and  w[k] -= L_kj w[j]  (forward substitution),
  //     L_kj = lane k's v[j] delivered INSIDE the FMA by DPP row_newbcast (gfx90a+; the only DPP form
  //     64-bit operations take): one v_fmac_f64_dpp per update.  The first form of this step moved
  //     L_kj through SGPRs: two v_readlane + the SGPR-read hazard + the FMA = 22 cycles per update
  //     against 5-6 (scripts/dev/valu_probe.hip), 3.1 us per diagonal step of which this is the rest.
  //     Everything here is issue-bound (a dependent f64 FMA issues every 4.9 cycles, an independent
  //     one every 4.4), so the order of independent work does not matter.
  int bad = -1;                                 // (wavefront 0) first non-positive pivot of the block
  auto diag_factor = [&](int jb) {
    const int c0 = 16 * jb;
    const int i = lane & 15;
    const int rbi = rowbase_t(jb, i, rowbase16(i)) + c0;
    double v[16], w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double a = s[rbi + (k < i ? k : i)];
      v[k] = (k <= i) ? a : 0.0;
      w[k] = (k == i) ? 1.0 : 0.0;
    }
    diag16(v, w);
    // Results to LDS from the first 16-lane row by UNCONDITIONAL stores (32 predicated ones, each with
    // its exec-mask branch and a mask reloaded from a spilled SGPR, were 2460 cycles of this step):
    //   L row i, k = 15 .. 0 to s[row i][min(k, i)]: what lane i does not own (k > i, garbage) lands on
    //     its own diagonal slot BEFORE the diagonal itself is stored (one wavefront's LDS stores keep
    //     their order);
    //   D^-1 column i, r = 0 .. 15 to (max(r, i), i): the rows above the diagonal (exact zeros) land
    //     on slot (i, i) before w[i] does.
    if (lane < 16) {
#pragma unroll
      for (int k = 15; k >= 0; --k) s[rbi + (k < i ? k : i)] = v[k];
      const int rbi16 = rowbase16(i), db = 136 * jb + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) dscr[db + (rowbase(r) > rbi16 ? rowbase(r) : rbi16)] = w[r];
    }
    // all sixteen pivots positive?  (see diag16)  If not -- rare -- the first bad one is the first
    // diagonal entry of the stored block that is not a positive number.
    if (((__builtin_amdgcn_ballot_w64(v[15] > 0.0) >> 15) & 1ull) == 0ull && bad < 0) {
      const double dg = s[rbi + i];
      const unsigned long long m = __builtin_amdgcn_ballot_w64(!(dg > 0.0)) & 0xffffull;
      bad = c0 + (m ? (int)__builtin_ctzll(m) : 15);
    }
  };
  // D^-1 of step jb as the A operand of X^T = D^-1 A^T:  lane <- D^-1(fr, 4 kk + fq)
  auto dinv_operand = [&](int jb, double (&dv)[4]) {
    const int b = 136 * jb + rbfr;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kk + fq;
      const double x = dscr[b + (k < fr ? k : fr)];
      dv[kk] = (k <= fr) ? x : 0.0;
    }
  };
  // X^T of tile row t for step jb (X = A[t][jb] D^-T, the final L[t][jb]): lane holds
  // X(fr, fq + 4 q), q = 0..3 -- the operand layout of BOTH sides of C -= X X^T with k = fq + 4 q
  auto xt_tile = [&](int t, int jb, const double (&dv)[4]) -> v4d {
    const int b = rowbase_t(t, fr, rbfr) + 16 * jb + fq;
    double bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bv[kk] = s[b + 4 * kk];
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mma(dv[kk], bv[kk], acc);
    return acc;
  };
  auto store_x = [&](int t, int jb, v4d x) {
    const int b = rowbase_t(t, fr, rbfr) + 16 * jb + fq;
#pragma unroll
    for (int q = 0; q < 4; ++q) s[b + 4 * q] = x[q];
  };
  // C(ti, tj) -= X_ti X_tj^T, tj <= ti
  auto upd_tile = [&](int ti, int tj, v4d xi, v4d xj) {
    const int cc = 16 * tj + fr;
    v4d acc;
    int idx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * ti + fq + 4 * r;
      idx[r] = rowbase_t(ti, uq[r], rbuq[r]) + cc;   // (cc > rr on a diagonal tile: a valid address of the next row, read and dropped)
      acc[r] = s[idx[r]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = mma(-xi[q], xj[q], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * ti + fq + 4 * r;
      if (ti != tj || cc <= rr) s[idx[r]] = acc[r];
    }
  };

  // Zeros above the diagonal.  The store phases below write whole 64-column half rows (value or zero), so
  // what is left is the quadrant rows 0..63 x columns 64..127 of the factor tile and of L^-1: eight
  // chunks of eight rows, one per step, by wavefronts 1-3 (fire-and-forget 16-byte stores where the
  // addresses allow; all 130 KB of the first version at once kept them at the store queue for 4 us).
  const bool wide_ok = ((((unsigned long long)A | (unsigned long long)dinv) & 15ull) == 0) && ((lda & 1) == 0);
  auto zero_chunk = [&](int W, int c) {
    if (!STORE) return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 z2 = {0.0, 0.0};
    const int pr = lane & 31, sub = lane >> 5;  // column pair, row of the pair of rows
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it == 1 && W != 1) continue;          // rows 6, 7 of the chunk: wavefront 1 again
      const int i = 8 * c + (it == 0 ? 2 * (W - 1) : 6) + sub;
      double* const pa = A + (long)i * lda + 64 + 2 * pr;
      double* const pd = dinv + i * 128 + 64 + 2 * pr;
      if (wide_ok) { *(d2*)pa = z2; *(d2*)pd = z2; }
      else { pa[0] = 0.0; pa[1] = 0.0; pd[0] = 0.0; pd[1] = 0.0; }
    }
  };
  double* const dump = dscr + 8 * 136 + lane;   // 64 spare slots: where a lane's store that must not happen goes
  // byte addressing relative to s for the worker passes: element (16 t + u, c) of the packed triangle is at
  // 8 rowbase16(u) + (128 u + 64) t + 8 (128 t^2 + c)  -- lane constants, one v_mad_u32_u24, a scalar
  char* const sb = (char*)s;
  const int dummy_off = (int)((char*)(dscr + 8 * 136 + 64) - sb);       // a 16x16 packed tile nobody reads
  const int c1fr = 128 * fr + 64, rbfq = 8 * (rbfr + fq);               // row fr, column fq (+ 4 kk)
  int c1q[4], rbq[4];                                                   // row fq + 4 r, column fr
#pragma unroll
  for (int r = 0; r < 4; ++r) { c1q[r] = 128 * uq[r] + 64; rbq[r] = 8 * (rbuq[r] + fr); }
  // One pass of a worker wavefront W = 1, 2, 3 (a compile-time constant: three straight-line copies, no
  // branch inside, so that the scheduler can run the LDS traffic of one tile under the MFMAs of another
  // -- with a guard per tile the stores of a tile waited for its four dependent MFMAs before the loads
  // of the next were issued: 3.7 us for a pass whose 64 MFMAs take 1.7).  Tiles that do not exist in
  // this pass (rows past the end) are computed on re-read data and stored to the spare slots.
  v4d x[7];
  auto worker_pass = [&](auto Wc, int jb) {
    constexpr int W = decltype(Wc)::value;
    auto block = [&](auto NXc, auto NTc) {
      constexpr int NX = decltype(NXc)::value, NT = decltype(NTc)::value;   // X tiles, tile slots computed in this pass
      // On this chip the f64 MFMA runs at the rate of the vector f64 FMA and a wavefront issues nothing
      // else to the vector unit while one executes (64.5 cycles each; with ~640 other vector instructions
      // in the pass, hand-pipelining the LDS traffic between the MFMAs alone changed nothing: 137 cycles
      // per MFMA either way).  So the cost of a pass is its MFMAs PLUS every other vector instruction,
      // and the first thing to do is to have few of those: addresses as one v_mad_u32_u24 + one add from per-lane constants, tiles
      // that do not exist in this pass (rows past the end) redirected by a SCALAR select to a 16x16
      // dummy tile instead of per-store v_cndmask, X negated once per tile row instead of per MFMA.
      // (1) X^T of the tile rows below jb; the X tiles of the previous step go to their place as
      //     L[.][jb-1] first (x[u] is still tile row jb+u of step jb-1)
      double dv[4];
      dinv_operand(jb, dv);
      if (jb > 0) {
#pragma unroll
        for (int u = 0; u < (NX == 7 ? 7 : NX + 1); ++u) {      // (the previous pass had one tile row more)
          const int t = jb + u;
          const bool mine = t < 8 && t % 3 == W - 1;                    // (scalar)
          const int o = mine ? __mul24(c1fr, t) + 8 * (128 * t * t + 16 * (jb - 1)) : dummy_off;
#pragma unroll
          for (int q = 0; q < 4; ++q) *(double*)(sb + o + rbfq + 32 * q) = x[u][q];
        }
      }
      double bv[NX][4];                         // all operand loads first, then the MFMAs back to back
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int t = jb + 1 + u < 8 ? jb + 1 + u : 7;
        const int o = __mul24(c1fr, t) + rbfq + 8 * (128 * t * t + 16 * jb);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bv[u][kk] = *(const double*)(sb + o + 32 * kk);
      }
      GH_SB();
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mma(dv[kk], bv[u][kk], acc);
        x[u] = acc;
      }
      v4d nx[7];
#pragma unroll
      for (int u = 1; u < 7; ++u) nx[u] = -x[u];
      // (2) this wavefront's nine tile slots e = 3 n + W - 1 of the 27 below the diagonal tile
      //     (tile (jb+1+ui, jb+1+uj), e = ui (ui + 1) / 2 - 1 + uj, ui = 1..6, uj <= ui)
      //     Software-pipelined by hand, pinned by sched_barrier: the LDS reads of tile n+1 are issued before
      //     the MFMAs of tile n (a read issued after them is waited for in full: ~150 cycles per tile),
      //     the stores of tile n come after the MFMAs of tile n+1 have been issued (no s_nop 15 for the result).
      v4d acc[NT];
      int o[NT][4];
#pragma unroll
      for (int n = -1; n < NT + 1; ++n) {
        if (n + 1 < NT) {                        // addresses and loads of tile n+1
          const int e = 3 * (n + 1) + W - 1, ui = slot_ui(e), uj = slot_uj(e);
          const bool exists = jb + 1 + ui < 8;
          const int ti = jb + 1 + ui, tj = jb + 1 + uj;
          const int so = exists ? 8 * (128 * ti * ti + 16 * tj) : dummy_off;   // (scalar)
          const int tm = exists ? ti : 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[n + 1][r] = __mul24(c1q[r], tm) + rbq[r] + so;
            acc[n + 1][r] = *(const double*)(sb + o[n + 1][r]);
          }
        }
        GH_SB();
        if (n >= 0 && n < NT) {                 // MFMAs of tile n
          const int e = 3 * n + W - 1, ui = slot_ui(e), uj = slot_uj(e);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[n] = mma(nx[ui][q], x[uj][q], acc[n]);
        }
        GH_SB();
        if (n >= 1) {                           // stores of tile n-1
          const int e = 3 * (n - 1) + W - 1, ui = slot_ui(e), uj = slot_uj(e);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (ui == uj) *(double*)((fr <= uq[r]) ? sb + o[n - 1][r] : (char*)dump) = acc[n - 1][r];   // diagonal tile: lower part only
            else *(double*)(sb + o[n - 1][r]) = acc[n - 1][r];
          }
        }
        GH_SB();
      }
    };
    // (the seven X tiles and nine slots of the first two passes; five and five from then on: rows jb+6.. are past the end)
    if (jb >= 0) {
      if (jb < 2) block(std::integral_constant<int, 7>(), std::integral_constant<int, 9>());
      else block(std::integral_constant<int, 5>(), std::integral_constant<int, 5>());
    }
    zero_chunk(W, jb + 1);
    return true;
  };
  // One barrier per 16-column step.  Pass jb = -1 is the first diagonal step alone; for jb >= 0, between
  // two barriers:
  //   wavefront 0:    X of tile row jb+1, diagonal tile jb+1, diagonal step jb+1 (-> D^-1 of jb+1)
  //   wavefronts 1-3: store the X tiles of the PREVIOUS step (the final L[.][jb-1]: the tiles they
  //                   replace were still being read by the others then), X^T of all tile rows below jb
  //                   (recomputed by each wavefront: no exchange), their third of the tile updates.
  // (The step loop is NOT unrolled -- unrolled, the masks and addresses of eight diagonal steps and 77
  // tile updates are hoisted and live across the whole kernel, 334 VGPRs, and the code no longer fits
  // the instruction cache.  x[u] is tile row jb+1+u.)
  // ---------------------------------------------------------------- block -> packed LDS image
  // Before (and, for wavefronts 1-3, beside) the first diagonal step -- NOT inside the step loop, whose
  // invariant-code motion would compute the 59 load addresses ahead of it and keep them, in scratch.
  // Wavefront 0: the first diagonal tile (lane i: row i) into its place in s; it then starts pass -1.
  // Wavefronts 1-3: the rest of the lower triangle in 64-column half rows: (row i, columns 0..63) for
  // i = 16..127, (row i, columns 64..127) for i = 64..127 -- 176 half rows, 59 per wavefront, ALL loads
  // in flight before the first is waited for.  The column index is clamped to the diagonal (the
  // surplus lanes re-read a cache line that is fetched anyway).
  if (wave == 0) {
    const int i = lane & 15;
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = src(i, k < i ? k : i);
    if (info_in != 0) return false;             // uniform over the workgroup: an earlier block already failed
#pragma unroll
    for (int k = 15; k >= 0; --k) s[rowbase16(i) + (k < i ? k : i)] = t[k];
  } else if (!SRC::kLoadsInFlight) {
    // (an evaluated block: nothing to have in flight -- element by element straight into the image; 59 evaluated values held in
    //  registers across the phase were 640 bytes of scratch per lane)
    if (info_in != 0) return false;
#pragma unroll 1
    for (int m = 0; m < 59; ++m) {
      const int q = (wave - 1) + 3 * m;
      const int i = q < 112 ? 16 + q : (q < 176 ? q - 48 : 127), j = (q < 112 ? 0 : 64) + lane;
      const double v = src(i, j < i ? j : i);
      *((q < 176 && j <= i) ? &s[rowbase(i) + j] : dump) = v;
    }
  } else {
    double v[59];
#pragma unroll
    for (int m = 0; m < 59; ++m) {
      const int q = (wave - 1) + 3 * m;         // (scalar)
      const int i = q < 112 ? 16 + q : (q < 176 ? q - 48 : 127), j = (q < 112 ? 0 : 64) + lane;
      v[m] = src(i, j < i ? j : i);
    }
    if (info_in != 0) return false;
#pragma unroll
    for (int m = 0; m < 59; ++m) {
      const int q = (wave - 1) + 3 * m;
      const int i = q < 112 ? 16 + q : (q < 176 ? q - 48 : 127), j = (q < 112 ? 0 : 64) + lane;
      *((q < 176 && j <= i) ? &s[rowbase(i) + j] : dump) = v[m];
    }
  }
#pragma unroll 1
  for (int jb = -1; jb < 7; ++jb) {
    GH_POTF2_STAMP(10 + 4 * (jb + 1));
    if (wave == 0) {
      if (jb >= 0) {
        double dv[4];
        dinv_operand(jb, dv);
        const v4d x1 = xt_tile(jb + 1, jb, dv);
        upd_tile(jb + 1, jb + 1, x1, x1);
      }
      GH_POTF2_STAMP(11 + 4 * (jb + 1));
      diag_factor(jb + 1);
      GH_POTF2_STAMP(12 + 4 * (jb + 1));
    } else if (wave == 1) {
      if (!worker_pass(std::integral_constant<int, 1>(), jb)) return false;
    } else if (wave == 2) {
      if (!worker_pass(std::integral_constant<int, 2>(), jb)) return false;
    } else {
      if (!worker_pass(std::integral_constant<int, 3>(), jb)) return false;
    }
    __syncthreads();
  }
  if (wave == 2) store_x(7, 6, x[0]);           // 1 + 7 % 3; step 6 left tile row 7 in x[0]
  if (tid == 0) (*fail_at_p) = bad;
  __syncthreads();
  GH_POTF2_STAMP(2);
  if ((*fail_at_p) >= 0) {
    if (tid == 0) *info = base + (*fail_at_p) + 1;
    return false;
  }
  // factor -> HBM: the half rows that hold part of the lower triangle, value or zero (no per-store
  // predicate); nothing waits for these stores
  if (STORE) {
    // (a pointer the optimiser cannot relate to the one of the load phase: otherwise the 32 load
    // addresses are kept for these stores across the whole step loop, in scratch)
    double* Ast = A;
    asm volatile("" : "+s"(Ast));
    const int j = tid & 127, ih = wave >> 1;      // (ih scalar: the row addresses are SGPR arithmetic)
#pragma unroll
    for (int q0 = 0; q0 < 64; q0 += 16) {
      if (q0 < 32 && (wave & 1)) continue;
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ih + 2 * (q0 + q);
        v[q] = s[(ih + q0 + q) * (2 * (q0 + q) + 1) + (j < i ? j : i)];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ih + 2 * (q0 + q);
        Ast[(long)i * lda + j] = (j <= i) ? v[q] : 0.0;
      }
    }
  }
  __syncthreads();                                // every thread has the factor in registers: s is free
  GH_POTF2_STAMP(3);

  // ================================================================ phase 2: L^-1, in place in LDS
  // (a) the eight 16x16 diagonal inverses exist already: into the diagonal tiles of s
#pragma unroll
  for (int e = tid; e < 8 * 256; e += 256) {
    const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
    if (c <= r) s[rowbase_t(b, r, rowbase16(r)) + 16 * b + c] = dscr[136 * b + rowbase16(r) + c];
  }
  __syncthreads();
  GH_POTF2_STAMP(4);
  // (b) doubling: blocks of size sz = 16, 32, 64.  Pair p: P0 = 2 p sz,
  //     A^-1 = s[P0 : P0+sz, P0 : P0+sz], B^-1 = s[P0+sz : P0+2sz, P0+sz : P0+2sz] (both lower
  //     triangular, already inverted), C = L[P0+sz : P0+2sz, P0 : P0+sz] (still the factor)
  //     ->  C is overwritten by  X = -B^-1 (C A^-1).  Job = (pair, 16-column tile column tj): always
  //     four jobs, one per wavefront.  T(:, tj) = C A^-1(:, tj) lands in MFMA result registers
  //     T(fq + 4 r + 16 kt, fr), which IS the B operand of the second product with k = fq + 4 r.
  //     One barrier between the last read of C and its overwriting, one after.
  // (one instantiation per level: with the level a loop variable, the trip counts of the inner loops become
  //  constants only after the outer loop has been unrolled, too late for the register promotion of the
  //  operand buffers -- they stay in scratch memory)
  auto level = [&](auto LVc) {
    constexpr int lv = decltype(LVc)::value;
    constexpr int sz = 16 << lv, tps = 1 << lv;   // tiles per side of a block
    const int p = wave / tps, tj = wave % tps;
    const int P0 = 2 * p * sz, Q0 = P0 + sz;
    v4d T[4], X[4];
    // B operands of the first product: A^-1(16 kt + u, 16 tj + fr), u = 4 kk + fq, for EVERY kt (zero
    // above the diagonal -- tiles kt < tj whole; no wave-dependent branch: the level is one basic block
    // and the loads of one product run under the MFMAs of another)
    double bop[4][4];
#pragma unroll
    for (int kt = 0; kt < tps; ++kt) {
      const bool lower = kt > tj, ondiag = kt == tj;              // (scalar)
      const int tt = (P0 >> 4) + (lower ? kt : tj);               // a stored tile row whatever kt is
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int u = 4 * kk + fq;
        const int uu = lower ? u : (u > fr ? u : fr), cc = lower ? fr : (u > fr ? fr : u);
        const double b = s[rowbase_t(tt, uu, rowbase16(uu)) + P0 + 16 * tj + cc];
        bop[kt][kk] = (lower || (ondiag && u >= fr)) ? b : 0.0;
      }
    }
    // The operands of tile row ti+1 are loaded BEFORE the MFMAs of tile row ti are issued (sched_barrier
    // pins that): a load issued after them is waited for in full, once per tile.
    // (flat double-buffers indexed by constants after unrolling: a sub-array passed by reference to a
    //  lambda is not scalarised and ends up in scratch memory)
    double av[32];
    auto load_c = [&](int ti, int slot) {                        // C(16 ti + fr, 16 kt + 4 kk + fq)
      const int ra = rowbase_t((Q0 >> 4) + ti, fr, rbfr) + P0 + fq;
#pragma unroll
      for (int kt = 0; kt < tps; ++kt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) av[16 * slot + 4 * kt + kk] = s[ra + 16 * kt + 4 * kk];
    };
    load_c(0, 0);
#pragma unroll
    for (int ti = 0; ti < tps; ++ti) {
      if (ti + 1 < tps) load_c(ti + 1, (ti + 1) & 1);
      GH_SB();
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < tps; ++kt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mma(av[16 * (ti & 1) + 4 * kt + kk], bop[kt][kk], acc);
      T[ti] = -acc;                                              // (the sign of X = -B^-1 T)
      GH_SB();
    }
    // second product: A operand B^-1(16 ti + fr, 16 kt + fq + 4 r), zero above the diagonal
    double bi[32];
    auto load_b = [&](int ti, int slot) {
      const int rbq = rowbase_t((Q0 >> 4) + ti, fr, rbfr) + Q0;
#pragma unroll
      for (int kt = 0; kt < tps; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt < ti) {
            bi[16 * slot + 4 * kt + r] = s[rbq + 16 * kt + uq[r]];
          } else if (kt == ti) {
            const double b = s[rbq + 16 * kt + (uq[r] < fr ? uq[r] : fr)];
            bi[16 * slot + 4 * kt + r] = (uq[r] <= fr) ? b : 0.0;
          }
        }
    };
    load_b(0, 0);
#pragma unroll
    for (int ti = 0; ti < tps; ++ti) {
      if (ti + 1 < tps) load_b(ti + 1, (ti + 1) & 1);
      GH_SB();
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt <= ti; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mma(bi[16 * (ti & 1) + 4 * kt + r], T[kt][r], acc);
      X[ti] = acc;
      GH_SB();
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < tps; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s[rowbase_t((Q0 >> 4) + ti, uq[r], rbuq[r]) + P0 + 16 * tj + fr] = X[ti][r];
    }
    __syncthreads();
  };
  level(std::integral_constant<int, 0>());
  level(std::integral_constant<int, 1>());
  level(std::integral_constant<int, 2>());
  GH_POTF2_STAMP(5);
  // (c) L^-1 to HBM, likewise
  if (STORE) {
    const int j = tid & 127, ih = wave >> 1;      // (ih scalar: the row addresses are SGPR arithmetic)
#pragma unroll
    for (int q0 = 0; q0 < 64; q0 += 16) {
      if (q0 < 32 && (wave & 1)) continue;
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ih + 2 * (q0 + q);
        v[q] = s[(ih + q0 + q) * (2 * (q0 + q) + 1) + (j < i ? j : i)];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ih + 2 * (q0 + q);
        dinv[i * 128 + j] = (j <= i) ? v[q] : 0.0;
      }
    }
  }
  GH_POTF2_STAMP(6);
  return true;
}
// (the dense solver's form: the block is read from A)
template <bool STORE = true>
__device__ __forceinline__ bool potf2_body(double* A, long lda, double* dinv, long long* info, long long base,
                                           double* s, double* dscr, int* fail_at_p) {
  const GhPotf2Mem src{A, lda};
  return potf2_body<STORE, GhPotf2Mem>(A, lda, dinv, info, base, s, dscr, fail_at_p, src);
}
#undef GH_SB
}  // namespace gh_potf2
#endif
