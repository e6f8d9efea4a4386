// RETIRED ARM (source snapshot, not built): the first MFMA form of the 128x128 Cholesky + inverse kernel, 82 us per block
// against 33 us of gh_potf2_body.h (GEORGE_AMD_POTF2=v1 until round 3); phase table in DESIGN.md section 4.
// gh_potf2_body.h -- the 128x128 Cholesky + inverse of gh_potf2.hip as a device function, so that the
// fused panel kernel (gh_gemm.hip, panel_server_kernel) can run it from a persistent workgroup.
// See gh_potf2.hip for the description of the algorithm.
#pragma once
#include "gh_common.h"

#ifndef GH_POTF2_BODY_V1_H_
#define GH_POTF2_BODY_V1_H_
namespace gh_potf2_v1 {
typedef double v4d __attribute__((ext_vector_type(4)));

#define T 128
#define IP 17                                   // pitch of the 16x16 diagonal-inverse scratch
#define PK(i, j) ((((i) * ((i) + 1)) >> 1) + (j))   // packed lower-triangular index, j <= i

// value of `v` in lane `src` (a compile-time constant after unrolling), delivered through SGPRs
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------- MFMA tile helper
// one 16x16 tile:  acc += A(16 x 4*nkk) * B(4*nkk x 16), operands fetched by functors A(i, k) and
// B(k, j);  lane map: A operand lane l <- A(l & 15, 4kk + (l >> 4)),
//                     B operand lane l <- B(4kk + (l >> 4), l & 15),
//                     acc[r] <-> C((l >> 4) + 4r, l & 15).
// All operand fetches of a tile (at most NK k-steps) are issued BEFORE the dependent MFMA chain:
// the phase-2 operands come from HBM/L2 (`dinv`), and one exposed load latency per k-step is what
// made the first MFMA version 180 us.  Loads are unconditional (the k index is clamped, surplus
// MFMAs are skipped by a wave-uniform test): a per-element load predicate makes hipcc wait per
// element.
template <int NK, typename FA, typename FB>
__device__ __forceinline__ v4d tile_mma(v4d acc, int kk0, int kk1, FA fa, FB fb, int lane) {
  const int fr = lane & 15, fk = lane >> 4;
  double a[NK], b[NK];
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    const int kk = (kk0 + q < kk1) ? kk0 + q : kk1 - 1;
    a[q] = fa(fr, 4 * kk + fk);
    b[q] = fb(4 * kk + fk, fr);
  }
#pragma unroll
  for (int q = 0; q < NK; ++q)
    if (kk0 + q < kk1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}


// phase time stamps for scripts/potf2_phases.hip; nothing in the library build
#ifndef GH_POTF2_STAMP
#define GH_POTF2_STAMP(k)
#endif

#define GH_POTF2V1_S_DOUBLES (128 * 129 / 2)
#define GH_POTF2V1_INV_DOUBLES (8 * 16 * 17)
#define GH_POTF2V1_INV_DOUBLES_NARROW (64 * 16)
// s: T(T+1)/2 doubles, inv16: 8*16*IP doubles, rdiag: T doubles, fail_at_p: one int -- all LDS.
// Returns false when the block is not positive definite (then *info is set) or an earlier one was not.
// CWMAX: widest column chunk of the doubling products (32: scratch 64 x 32 doubles; 16: 64 x 16, two more
// barrier pairs per block, but 75 instead of 83 KB of LDS -- TWO workgroups per CU for batched launches).
template <int CWMAX = 32>
__device__ __forceinline__ bool potf2_body(double* A, long lda, double* dinv, long long* info, long long base,
                                           double* s, double* inv16, double* rdiag, int* fail_at_p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // latency-bound chain of dependent steps that, under look-ahead, shares its CU with wavefronts
  // of the trailing SYRK issuing 64 MFMAs back to back: take the instruction arbiter's top priority
  __builtin_amdgcn_s_setprio(3);
  GH_POTF2_STAMP(0);
  if (*info != 0) return false;                 // uniform: an earlier block already failed
  if (tid == 0) (*fail_at_p) = -1;
  // block -> packed LDS image; 16 unconditional loads in flight per thread (a load under the
  // `j <= i` predicate is waited for one at a time: 22 % of the kernel in the first version)
  {
    const int j = tid & 127, ih = tid >> 7;
#pragma unroll
    for (int q0 = 0; q0 < 64; q0 += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = A[(long)(ih + 2 * (q0 + q)) * lda + j];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ih + 2 * (q0 + q);
        if (j <= i) s[PK(i, j)] = v[q];
      }
    }
  }
  __syncthreads();
  GH_POTF2_STAMP(1);

  // ================================================================ phase 1: Cholesky
  // (a) diagonal block jb: ONE wavefront, register-resident: lane i (mod 16) holds row i of the
  //     block in 16 VGPR pairs, a column's pivot and multipliers travel by v_readlane (SGPR
  //     broadcast), the j/k loops are fully unrolled so every register index is static.  (The
  //     first MFMA version did this through volatile LDS round trips: ~1100 cycles per column.)
  auto diag_factor = [&](int jb) {
    const int c0 = 16 * jb;
    const int i = lane & 15;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (k <= i) ? s[PK(c0 + i, c0 + k)] : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double d = bcast_lane(a[j], j);
      if (!(d > 0.0)) {                         // also catches NaN (uniform: d is a broadcast)
        if (lane == 0 && (*fail_at_p) < 0) (*fail_at_p) = c0 + j;
        d = 1.0;
      }
      // sqrt(d) and 1/sqrt(d) from ONE v_rsq_f64 seed (~2^-23) + two Newton steps and a final
      // correction each: half the dependent chain of sqrt() followed by a division
      double y = __builtin_amdgcn_rsq(d);
      const double hd = 0.5 * d;
      y = fma(y, fma(-hd * y, y, 0.5), y);
      y = fma(y, fma(-hd * y, y, 0.5), y);
      double ajj = d * y;
      ajj = fma(0.5 * y, fma(-ajj, ajj, d), ajj);
      const double inv = fma(fma(-ajj, y, 1.0), y, y);
      if (lane == 0) rdiag[c0 + j] = inv;
      a[j] = (i == j) ? ajj : a[j] * inv;
#pragma unroll
      for (int k = j + 1; k < 16; ++k) {
        const double lkj = bcast_lane(a[j], k);
        a[k] -= a[j] * lkj;                     // meaningful for i >= k; other lanes' values are never stored
      }
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k <= i) s[PK(c0 + i, c0 + k)] = a[k];
    }
  };
  // (c) one 16x16 tile of the trailing update for step jb: C(ti,tj) -= P_ti P_tj^T, K = 16
  auto update_tile = [&](int jb, int ti, int tj) {                // ti >= tj > jb (absolute tile indices)
    const int c0 = 16 * jb, R0 = 16 * ti, C0 = 16 * tj;
    const int cc = C0 + (lane & 15);
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = R0 + (lane >> 4) + 4 * r;
      acc[r] = (cc <= rr) ? s[PK(rr, cc)] : 0.0;
    }
    acc = tile_mma<4>(acc, 0, 4,
                      [&](int i, int k) { return -s[PK(R0 + i, c0 + k)]; },
                      [&](int k, int j) { return s[PK(C0 + j, c0 + k)]; }, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = R0 + (lane >> 4) + 4 * r;
      if (cc <= rr) s[PK(rr, cc)] = acc[r];
    }
  };
  // Software pipeline over the 16-column steps: the trailing update of step jb first brings tile
  // column jb+1 up to date (all wavefronts), then wavefront 0 factors diagonal block jb+1 WHILE
  // wavefronts 1-3 finish the rest of the update -- the one-wavefront diagonal step (a fifth of
  // this kernel) no longer idles the other three.
  if (wave == 0) diag_factor(0);
  __syncthreads();
  for (int jb = 0; jb < 7; ++jb) {
    const int c0 = 16 * jb;
    GH_POTF2_STAMP(10 + 4 * jb);
    // (b) panel rows below the diagonal block: solve x D^T = a, one row per thread
    {
      const int r = c0 + 16 + tid;
      if (r < T) {
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = s[PK(r, c0 + k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          double v0 = x[k], v1 = 0.0;             // two partial sums: half the dependent FMA chain
#pragma unroll
          for (int m = 0; m + 1 < k; m += 2) {
            v0 -= x[m] * s[PK(c0 + k, c0 + m)];
            v1 -= x[m + 1] * s[PK(c0 + k, c0 + m + 1)];
          }
          if (k & 1) v0 -= x[k - 1] * s[PK(c0 + k, c0 + k - 1)];
          x[k] = (v0 + v1) * rdiag[c0 + k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s[PK(r, c0 + k)] = x[k];
      }
    }
    __syncthreads();
    GH_POTF2_STAMP(11 + 4 * jb);
    // (c1) tile column jb+1
    for (int ti = jb + 1 + wave; ti < 8; ti += 4) update_tile(jb, ti, jb + 1);
    __syncthreads();
    GH_POTF2_STAMP(12 + 4 * jb);
    // (a) of step jb+1  ||  (c2) the tiles right of column jb+1
    if (wave == 0) {
      diag_factor(jb + 1);
      GH_POTF2_STAMP(13 + 4 * jb);
    } else {
      const int m = 6 - jb;                       // tile rows/cols right of column jb+1
      const int ntiles = m * (m + 1) / 2;
      for (int e = wave - 1; e < ntiles; e += 3) {
        int ti = 0, acc_t = 0;
        while (acc_t + ti + 1 <= e) { acc_t += ti + 1; ++ti; }
        update_tile(jb, jb + 2 + ti, jb + 2 + (e - acc_t));
      }
    }
    __syncthreads();
  }
  GH_POTF2_STAMP(2);
  if ((*fail_at_p) >= 0) {                             // (all threads see it: barrier above)
    if (tid == 0) *info = base + (*fail_at_p) + 1;
    return false;
  }
  // factor back to HBM, strict upper triangle of the tile zeroed
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    A[(long)i * lda + j] = (j <= i) ? s[PK(i, j)] : 0.0;
  }

  // ================================================================ phase 2: L^-1, in place in LDS
  // (the factor is already in HBM; `s` is free to become L^-1, `inv16` is the scratch for C A^-1)
  // (a) the eight 16x16 diagonal inverses; wavefront w takes blocks 2w and 2w+1.
  //     Registers again: lane r (mod 16) holds ROW r of the block (a[]) and COLUMN r of its
  //     inverse (x[]); x_i = -(sum_{k<i} L_ik x_k) / L_ii with L_ik broadcast from lane i.
  __syncthreads();                                // (the write-back above still reads s)
  GH_POTF2_STAMP(3);
  for (int bb = 0; bb < 2; ++bb) {
    const int bI = 2 * wave + bb, d0 = 16 * bI;
    const int c = lane & 15;
    double a[16], x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (k <= c) ? s[PK(d0 + c, d0 + k)] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double rd = rdiag[d0 + i];
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) acc += bcast_lane(a[k], i) * x[k];      // x[k] = 0 for k < c
      x[i] = (i < c) ? 0.0 : ((i == c) ? rd : -acc * rd);
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i >= c) s[PK(d0 + i, d0 + c)] = x[i];
    }
  }
  __syncthreads();
  GH_POTF2_STAMP(4);
  // (b) doubling: blocks of size sz = 16, 32, 64.  Pair p: P0 = 2 p sz,
  //     A^-1 = s[P0 : P0+sz, P0 : P0+sz], B^-1 = s[P0+sz : P0+2sz, P0+sz : P0+2sz] (both lower
  //     triangular, already inverted), C = L[P0+sz : P0+2sz, P0 : P0+sz] (still the factor)
  //     ->  C is overwritten by  X = -B^-1 (C A^-1).  Columns go in chunks of <= 32 (the scratch
  //     holds 64 x 32 doubles), left to right: chunk c of T = C A^-1 needs the columns >= c of C
  //     only (A^-1 is lower triangular), so overwriting the chunks already done is safe.
  //     Operands of the first version came from HBM (`dinv`): 20 % of the kernel.
  auto tri = [&](int r, int c) { return s[PK(r > c ? r : c, r > c ? c : r)]; };   // (valid address for any r, c)
  double* scr = inv16;
  for (int sz = 16; sz <= 64; sz *= 2) {
    const int tps = sz / 16;                      // tiles per side of a block
    const int cw = sz < CWMAX ? sz : CWMAX, tpc = cw / 16;   // chunk width, tile columns per chunk
    const int npair = 64 / sz;
    for (int c0 = 0; c0 < sz; c0 += cw) {
      const int njobs = npair * tps * tpc;
      // T[:, chunk] = C A^-1[:, chunk]   (k >= column: A^-1 lower triangular)
      for (int e = wave; e < njobs; e += 4) {
        const int p = e / (tps * tpc), rem = e % (tps * tpc), ti = rem / tpc, tj = rem % tpc;
        const int P0 = 2 * p * sz, col = c0 + 16 * tj;          // column offset inside the block
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = tile_mma<16>(acc, col / 4, sz / 4,
                           [&](int i, int k) { return s[PK(P0 + sz + 16 * ti + i, P0 + k)]; },
                           [&](int k, int j) { const double v = tri(P0 + k, P0 + col + j); return k >= col + j ? v : 0.0; }, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          scr[(p * sz + 16 * ti + (lane >> 4) + 4 * r) * cw + 16 * tj + (lane & 15)] = acc[r];
      }
      __syncthreads();
      // X[:, chunk] = -B^-1 T[:, chunk]   (k <= row: B^-1 lower triangular)
      for (int e = wave; e < njobs; e += 4) {
        const int p = e / (tps * tpc), rem = e % (tps * tpc), ti = rem / tpc, tj = rem % tpc;
        const int P0 = 2 * p * sz;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = tile_mma<16>(acc, 0, 4 * (ti + 1),
                           [&](int i, int k) { const double v = tri(P0 + sz + 16 * ti + i, P0 + sz + k); return k <= 16 * ti + i ? -v : 0.0; },
                           [&](int k, int j) { return scr[(p * sz + k) * cw + 16 * tj + j]; }, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s[PK(P0 + sz + 16 * ti + (lane >> 4) + 4 * r, P0 + c0 + 16 * tj + (lane & 15))] = acc[r];
      }
      __syncthreads();
    }
  }
  GH_POTF2_STAMP(5);
  // (c) L^-1 to HBM, zeros above the diagonal
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    dinv[idx] = (j <= i) ? s[PK(i, j)] : 0.0;
  }
  GH_POTF2_STAMP(6);
  return true;
}
#undef PK
#undef T
#undef IP
}  // namespace gh_potf2_v1
#endif
