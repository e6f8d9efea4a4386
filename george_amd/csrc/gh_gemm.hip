// gh_gemm.hip -- fp64 GEMM / SYRK family on the CDNA4 matrix pipe.
//
// One kernel family serves every O(N^3) step of the solver (trailing SYRK
// update, panel updates, TRSM-by-inverse multiplies, multi-RHS solves, K^-1):
//     C[m][n] = beta * C[m][n] + alpha * sum_k A(m,k) * B(n,k)
// with each operand either "k-major" (row-major, k contiguous) or "m-major"
// (k strided).
//
// Tiling (gfx950): 128x128 C tile per 256-thread workgroup = 2x2 wavefronts,
// each wavefront a 64x64 sub-tile = 4x4 v_mfma_f64_16x16x4_f64 accumulators
// (64 f64 = 128 VGPRs/lane).  K is consumed in slabs of 16 through a
// double-buffered LDS image, one barrier per slab, 2 workgroups per CU.  Per 16
// MFMAs (1024 matrix-pipe cycles) a wavefront needs 8 fragment reads, so the
// kernel is MFMA-issue bound, not LDS bound.
//
// The operand path is gemm_f64_mfma_dma: global -> LDS by global_load_lds_dwordx4, XOR-swizzled image
// (the register-staged predecessor and the 4x4x4-4b instruction form are retired: scripts/dev/arms/);
// gemm_f64_valu is a plain-VALU kernel with the same semantics that cross-checks the MFMA lane maps
// on the device (GEORGE_AMD_NO_MFMA=1 / gh_debug_set_mfma(0)).
#include <stdlib.h>
#include <algorithm>
#include "gh_common.h"
#include "../../include/george_amd_debug.h"

#include "gh_gemm_tile.h"

#ifndef GH_GEMM_SP_DEFAULT
#define GH_GEMM_SP_DEFAULT 1
#endif

struct GemmDev {
  double* C; long ldc;
  const double* A; long lda;
  const double* B; long ldb;
  long K;
  double alpha, beta;
  int tiles_m, tiles_n;
  int lower, klo_max, khi_col, khi_row;
  long nblk;
  int preload;      // beta == +-alpha != 0: accumulators start from (beta/alpha) * C, write-back is store-only
  int prio;         // launch cannot fill the chip (panel-chain GEMMs): run at top wavefront priority
  int grouped;      // tile order inside an XCD's chunk: row groups walked column-major (tile_of) instead of row-major
  // staircase C: stair_n row groups of stair_h tile rows, the WIDEST FIRST in the tile order (slot s = group stair_n - 1 - s);
  // stair_pre[s] = tiles before slot s (stair_pre[stair_n] = nblk), so slot s is stair_h x ((pre[s+1] - pre[s]) / stair_h) tiles
  int stair_n, stair_h;
  unsigned stair_pre[GH_GEMM_STAIR_MAX + 1];
};

// XCD-aware remap (bijective for any nblk): workgroup b runs on XCD b % 8; give each XCD
// a contiguous range of logical tiles so neighbours share operand panels in one L2.
__device__ __forceinline__ long xcd_remap(long bid, long nblk) {
  const long q = nblk / 8, r = nblk % 8;
  const long xcd = bid % 8, idx = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Workgroup -> C tile.  Inside each XCD's chunk the order is "row groups, column-major inside a group": GH_TILE_GROUP tile rows
// at a time, walked column by column, so the ~64 tiles one XCD has in flight form an 8 x 8 block that shares 8 + 8 operand
// slabs in its L2 instead of the 1 + 64 of a row-major walk (whose column slabs stream from HBM every time: FETCH_SIZE was
// 4.5x the algorithmic read bytes, profiles/r04/traffic_N65536.json).  Bijective for any shape -- no padded grid -- so the
// skinny panel GEMMs keep their launch size (the round-1 attempt padded to whole super-tiles and lost 8 % there).
#ifndef GH_TILE_GROUP
#define GH_TILE_GROUP 8
#endif
__device__ __forceinline__ void grouped_rect(long l, int rows, int cols, int& tm, int& tn) {
  constexpr int G = GH_TILE_GROUP;
  const int g = (int)(l / ((long)G * cols));
  const int gg = min(G, rows - g * G);
  const long r = l - (long)g * G * cols;
  tn = (int)(r / gg);
  tm = g * G + (int)(r % gg);
}
__device__ __forceinline__ void grouped_tri(long l, int T, int& tm, int& tn) {
  // rows g*G .. g*G+G-1 hold G*g*G + G(G+1)/2 tiles; before group g: G*G*g(g-1)/2 + g*G(G+1)/2
  constexpr int G = GH_TILE_GROUP;
  constexpr long H = (long)G * (G + 1) / 2;
  long g = (long)((sqrt(((double)H - 0.5 * G * G) * ((double)H - 0.5 * G * G) + 2.0 * G * G * (double)l) - ((double)H - 0.5 * G * G)) /
                  ((double)G * G));
  auto start = [&](long q) { return (long)G * G * q * (q - 1) / 2 + q * H; };
  while (start(g) > l) --g;
  while (start(g + 1) <= l) ++g;
  const int r0 = (int)g * G, gg = min(G, T - r0);
  long r = l - start(g);
  const long rect = (long)gg * (r0 + 1);        // columns 0 .. r0 are full height
  if (r < rect) {
    tn = (int)(r / gg);
    tm = r0 + (int)(r % gg);
    return;
  }
  r -= rect;
  int j = 1;                                    // column r0 + j holds rows r0 + j .. r0 + gg - 1
  while (r >= gg - j) { r -= gg - j; ++j; }
  tn = r0 + j;
  tm = r0 + j + (int)r;
}
__device__ __forceinline__ bool tile_of(const GemmDev& g, int& tm, int& tn) {
  // (triangular operands: the K extent shrinks along the tile order, so contiguous per-XCD chunks
  //  would hand one XCD all the long tiles -- deal those round-robin, in plain row-major order)
  const bool rowmajor = !g.grouped;
  if (g.stair_n) {                                        // staircase: find the slot, walk it column by column (its height is one "row group")
    const long ls = xcd_remap(blockIdx.x, g.nblk);
    int s = 0;
    while (s + 1 < g.stair_n && ls >= (long)g.stair_pre[s + 1]) ++s;
    const int r = (int)(ls - (long)g.stair_pre[s]);
    tn = r / g.stair_h;
    tm = (g.stair_n - 1 - s) * g.stair_h + r % g.stair_h;
    return true;
  }
  const long l = (g.klo_max | g.khi_col | g.khi_row) ? (long)blockIdx.x : xcd_remap(blockIdx.x, g.nblk);
  if (g.lower) {
    // lower TRAPEZOID: tiles (tm, tn) with tn <= tm and tn < tiles_n (a square C, tiles_n == tiles_m, is the triangle): the first
    // tiles_n tile rows hold 1, 2, .. tiles_n tiles, every row below them tiles_n
    const long tri = (long)g.tiles_n * (g.tiles_n + 1) / 2;
    if (l < tri) {
      if (!rowmajor) {
        grouped_tri(l, g.tiles_n, tm, tn);
      } else {
        long t = (long)((sqrt(8.0 * (double)l + 1.0) - 1.0) * 0.5);
        while (t * (t + 1) / 2 > l) --t;
        while ((t + 1) * (t + 2) / 2 <= l) ++t;
        tm = (int)t;
        tn = (int)(l - t * (t + 1) / 2);
      }
    } else if (!rowmajor) {
      grouped_rect(l - tri, g.tiles_m - g.tiles_n, g.tiles_n, tm, tn);
      tm += g.tiles_n;
    } else {
      tm = g.tiles_n + (int)((l - tri) / g.tiles_n);
      tn = (int)((l - tri) % g.tiles_n);
    }
  } else if (!rowmajor) {
    grouped_rect(l, g.tiles_m, g.tiles_n, tm, tn);
  } else {
    tm = (int)(l / g.tiles_n);
    tn = (int)(l % g.tiles_n);
  }
  return true;
}

template <bool KM>
__device__ __forceinline__ void stage_load(const double* base, long ld, long r0, long k0, int tid, double2 (&v)[4]) {
  if (KM) {        // element (row, k) at base[row*ld + k]
    const int row = tid >> 3, kc = (tid & 7) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = *reinterpret_cast<const double2*>(base + (r0 + row + 32 * i) * ld + k0 + kc);
  } else {         // element (row, k) at base[k*ld + row]
    const int k = tid >> 6, m = (tid & 63) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = *reinterpret_cast<const double2*>(base + (k0 + k + 4 * i) * ld + r0 + m);
  }
}
template <bool KM>
__device__ __forceinline__ void stage_store(double* s, int tid, const double2 (&v)[4]) {
  if (KM) {
    const int row = tid >> 3, kc = (tid & 7) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<double2*>(s + (row + 32 * i) * LS + kc) = v[i];
  } else {
    const int k = tid >> 6, m = (tid & 63) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[m * LS + k + 4 * i] = v[i].x;
      s[(m + 1) * LS + k + 4 * i] = v[i].y;
    }
  }
}

__device__ __forceinline__ void k_range(const GemmDev& g, long row0, long col0, long& kbeg, long& kend) {
  kbeg = 0; kend = g.K;
  if (g.klo_max) kbeg = row0 > col0 ? row0 : col0;
  if (g.khi_col && col0 + BN < kend) kend = col0 + BN;
  if (g.khi_row && row0 + BM < kend) kend = row0 + BM;
}

// C tile write-back of one wavefront's 64x64 sub-tile at (r0, c0).  f64 MFMA C/D map:
// col = lane & 15, row = (lane >> 4) + 4 * reg.  The beta test is hoisted and the 16 C loads of a
// 16-row group are issued together: written as `beta == 0 ? v : beta * c + v` per element the
// compiler emits 64 serial load -> vmcnt(0) -> store round trips per lane.
//
// When beta == +-alpha (every accumulate call of the solver is C -= A B^T) the read moves to the
// FRONT instead: the accumulators start from (beta/alpha) * C -- exact, the ratio is +-1 -- while
// the first operand slab is still in flight, and the write-back is alpha * acc, store-only
// (measured on the SYRK shape at K = 1024: the trailing read costs 8-10 %).
__device__ __forceinline__ void gemm_init_acc(const GemmDev& g, v4d (&acc)[4][4], long r0, long c0, int fr, int fk) {
  if (g.preload) {
    const double rho = (g.beta == g.alpha) ? 1.0 : -1.0;
    const double* cbase = g.C + (r0 + fk) * g.ldc + c0 + fr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j][r] = rho * cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
}

__device__ __forceinline__ void gemm_epilogue(const GemmDev& g, const v4d (&acc)[4][4], long r0, long c0,
                                              int fr, int fk) {
  const double alpha = g.alpha, beta = g.preload ? 0.0 : g.beta;
  double* cbase = g.C + (r0 + fk) * g.ldc + c0 + fr;
  if (beta == 0.0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* crow = cbase + (long)(i * 16 + 4 * r) * g.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) crow[j * 16] = alpha * acc[i][j][r];
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double c[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[r][j] = cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = fma(beta, c[r][j], alpha * acc[i][j][r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA kernel (the default for every layout): slabs go global -> LDS directly
// (global_load_lds_dwordx4, 1 KiB per wavefront instruction), so there are no staging VGPRs and
// no ds_write pass, and the loads of slab t+1 have the 64-MFMA block of slab t (~4000 cycles) to
// land before the vmcnt(0) in front of the barrier.  The DMA writes LDS lane-linearly
// (base + lane*16), so the LDS image cannot be padded; it is XOR-swizzled instead, through the
// SOURCE address, so that a fragment read (16 rows x 2 k per 32 lanes) touches 32 distinct 8-byte
// slots of the 256-byte bank row, like the padded layout of the register-staged kernel:
//   k-major operand (row r, k contiguous): LDS image [row][16 k], one instruction = 8 rows;
//       16-byte piece p of row r holds k-pair p ^ ((r >> 1) & 7);
//   m-major operand (k strided, rows contiguous): LDS image [k][128 rows], one instruction = one
//       k (1 KiB of rows); piece p of k-row k holds row-pair p ^ ((k & 1) << 3), i.e. odd k swap
//       the two 128-byte halves of every 256 bytes -- for the reader that is "16-row group i^1".
// LOWER (== g.lower) only names the symbol: rocprof then tells the triangular-grid launches (the
// trailing SYRK updates, the roofline kernel of bench.py) from the rectangular panel GEMMs.
template <bool A_KM, bool B_KM, bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma_dma(GemmDev g) {
  __shared__ __attribute__((aligned(1024))) double sA[2][BM * BK];
  __shared__ __attribute__((aligned(1024))) double sB[2][BN * BK];
  int tm, tn;
  if (!tile_of(g, tm, tn)) return;
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const long row0 = (long)tm * BM, col0 = (long)tn * BN;
  long kbeg, kend;
  k_range(g, row0, col0, kbeg, kend);
  // (the wavefront index as a SCALAR: the LDS destination of every DMA goes through M0, and from a vector register that is a
  //  v_readfirstlane per instruction and slab)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;

  v4d acc[4][4];
  const long nk = (kend - kbeg) / BK;
  DmaOperand<A_KM> oa;
  DmaOperand<B_KM> ob;
  oa.init(g.A, g.lda, row0, kbeg, wave, lane, wm);
  ob.init(g.B, g.ldb, col0, kbeg, wave, lane, wn);
  const int dst = wave * 4 * 128;                  // this wavefront's first 1 KiB (= 128 doubles) piece

  if (nk > 0) { GH_DMA_ISSUE(oa, sA[0]) GH_DMA_ISSUE(ob, sB[0]) }
  gemm_init_acc(g, acc, row0 + wm * 64, col0 + wn * 64, fr, fk);      // C loads ride on the first slab's latency
  __syncthreads();                                  // (hipcc puts the vmcnt(0) of the DMA in front of the barrier)
  for (long kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    double a[4][4], b[4][4];
    if constexpr (A_KM && B_KM) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double2 v = oa.frag2(sA[cur], q, i); a[2 * q][i] = v.x; a[2 * q + 1][i] = v.y; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double2 v = ob.frag2(sB[cur], q, j); b[2 * q][j] = v.x; b[2 * q + 1][j] = v.y; }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kk][i] = oa.frag(sA[cur], kk, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[kk][j] = ob.frag(sB[cur], kk, j);
      }
    }
    // fragment reads first, THEN the DMA of the next slab into the other buffer (last read one
    // barrier ago): a DMA issued ahead of the reads would make the compiler wait for it
    // (vmcnt(0)) in front of every later ds_read.  The first k-group of MFMAs goes ahead of the
    // DMA issue: hipcc drains lgkmcnt to 0 after a global_load_lds, so MFMAs placed before it
    // start on counted waits as soon as their own fragments have landed.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      if (cur) { GH_DMA_ISSUE(oa, sA[0]) GH_DMA_ISSUE(ob, sB[0]) }
      else     { GH_DMA_ISSUE(oa, sA[1]) GH_DMA_ISSUE(ob, sB[1]) }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);              // (else 63 of the 64 MFMAs sink below the barrier)
    __syncthreads();
  }
  gemm_epilogue(g, acc, row0 + wm * 64, col0 + wn * 64, fr, fk);
}

// ---------------------------------------------------------------------------------------------
// The same kernel for k-major x k-major operands with the slab loop SOFTWARE-PIPELINED by half a slab (round 5).
// In gemm_f64_mfma_dma a wavefront leaves the end-of-slab barrier, issues its 16 ds_read_b128 and can start the first matrix
// instruction of the slab only when the first of them are back: ~300 cycles of a 4128-cycle slab in which this wavefront feeds
// nothing to the matrix pipe ("the 16 ds_read_b128 in front of each slab's first MFMA").  Here the barrier sits
// in the MIDDLE of a slab's 64 matrix instructions:
//     top of slab t     : read the second half of slab t's fragments (k-steps 2, 3) -- they land under k-steps 0, 1,
//                         whose fragments were read half a slab ago;
//     k-steps 0, 1      : 32 matrix instructions;
//     barrier           : every wavefront has read all of slab t (buffer t & 1 is free) and its DMA pieces of slab t + 1 are in;
//     DMA of slab t + 2 into buffer t & 1; read the FIRST half of slab t + 1's fragments (k-steps 0, 1) into the registers
//                         k-steps 0, 1 of slab t have just released -- they land under k-steps 2, 3 of slab t;
//     k-steps 2, 3      : 32 matrix instructions.
// No matrix instruction waits for an LDS read issued less than 32 matrix instructions (~2000 cycles) earlier, no extra registers
// (the halves alternate), the same two LDS buffers, a DMA has a full slab to land as before.  Every accumulator still adds k-steps
// 0, 1, 2, 3 of slab 0, 1, ... in this order: the bits are those of gemm_f64_mfma_dma.
// The fragment reads are inline asm: hipcc orders every LDS read it can see behind a vmcnt(0) once an LDS-DMA has been issued
// (it cannot tell the buffers apart), which would put the DMA's whole latency in front of the reads that follow it.  The waits
// are therefore written by hand: LDS reads return in order, `s_waitcnt lgkmcnt(8)` = "all but the 8 youngest are back"; the
// wait statements name the registers they guard as in/out operands so the compiler keeps every consumer behind them.
// (v2d, GH_SP_READ8 / GH_SP_WAIT / GH_SP_MFMA: gh_gemm_tile.h -- gh_tile128_nt runs the same loop)
template <bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_f64_mfma_dma_sp(GemmDev g) {
  // ONE array: the asm reads take the buffers' byte offsets as immediates (A0 | A1 | B0 | B1, 16 KiB each)
  __shared__ __attribute__((aligned(1024))) double sm[4 * BM * BK];
  double* const sA0 = sm; double* const sA1 = sm + BM * BK;
  double* const sB0 = sm + 2 * BM * BK; double* const sB1 = sm + 3 * BM * BK;
  int tm, tn;
  if (!tile_of(g, tm, tn)) return;
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const long row0 = (long)tm * BM, col0 = (long)tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  v4d acc[4][4];
  const long nk = g.K / BK;                            // even, >= 2 (the launcher's condition)
  DmaOperand<true> oa, ob;
  oa.init(g.A, g.lda, row0, 0, wave, lane, wm);
  ob.init(g.B, g.ldb, col0, 0, wave, lane, wn);
  const int dst = wave * 4 * 128;
  // LDS byte addresses of this lane's fragment pieces in buffer 0: half q of the row (16 i + fr) is piece off2[q] of its image
  const unsigned lbase = (unsigned)(unsigned long)(gh_lds_void*)sm;
  const unsigned pa0 = lbase + (unsigned)(oa.f0 + oa.off2[0]) * 8u, pa1 = lbase + (unsigned)(oa.f0 + oa.off2[1]) * 8u;
  const unsigned pb0 = lbase + 2u * BM * BK * 8u + (unsigned)(ob.f0 + ob.off2[0]) * 8u;
  const unsigned pb1 = lbase + 2u * BM * BK * 8u + (unsigned)(ob.f0 + ob.off2[1]) * 8u;
  constexpr int B1 = BM * BK * 8;                      // byte offset of buffer 1 of either operand
  v2d a01[4], b01[4], a23[4], b23[4];                  // [16-row group]: .x / .y = the even / odd k-step of the half

  GH_DMA_ISSUE(oa, sA0) GH_DMA_ISSUE(ob, sB0)
  gemm_init_acc(g, acc, row0 + wm * 64, col0 + wn * 64, fr, fk);
  __syncthreads();
  GH_DMA_ISSUE(oa, sA1) GH_DMA_ISSUE(ob, sB1)
  GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, 0);
  for (long kt = 0; kt < nk; kt += 2) {
    const bool more = kt + 2 < nk;
    // ---- slab kt (buffer 0)
    GH_SP_READ8(a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3], pa1, pb1, 0);
    GH_SP_WAIT(8, a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3]);
    GH_SP_MFMA(a01, b01, 0) GH_SP_MFMA(a01, b01, 1)
    __builtin_amdgcn_sched_barrier(0);
    GH_SP_WAIT(0, a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3]);
    __syncthreads();                                   // (vmcnt(0): my pieces of slab kt + 1; everybody is done with buffer 0)
    if (more) { GH_DMA_ISSUE(oa, sA0) GH_DMA_ISSUE(ob, sB0) }
    GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, B1);
    __builtin_amdgcn_sched_barrier(0);
    GH_SP_MFMA(a23, b23, 0) GH_SP_MFMA(a23, b23, 1)
    __builtin_amdgcn_sched_barrier(0);
    // ---- slab kt + 1 (buffer 1)
    GH_SP_READ8(a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3], pa1, pb1, B1);
    GH_SP_WAIT(8, a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3]);
    GH_SP_MFMA(a01, b01, 0) GH_SP_MFMA(a01, b01, 1)
    __builtin_amdgcn_sched_barrier(0);
    GH_SP_WAIT(0, a23[0], a23[1], a23[2], a23[3], b23[0], b23[1], b23[2], b23[3]);
    if (more) {
      __syncthreads();
      if (kt + 3 < nk) { GH_DMA_ISSUE(oa, sA1) GH_DMA_ISSUE(ob, sB1) }
      GH_SP_READ8(a01[0], a01[1], a01[2], a01[3], b01[0], b01[1], b01[2], b01[3], pa0, pb0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    GH_SP_MFMA(a23, b23, 0) GH_SP_MFMA(a23, b23, 1)
    __builtin_amdgcn_sched_barrier(0);
  }
  gemm_epilogue(g, acc, row0 + wm * 64, col0 + wn * 64, fr, fk);
}

// ---------------------------------------------------------------------------------------------
// 64x64-tile variant for launches that cannot fill the chip anyway (the GEMMs inside the panel
// chain: <= 128 tiles of 128x128, K = 128..1024).  A 128x128x128 tile is 512 MFMAs per wavefront
// = 15 us on its one CU no matter how idle the other 255 are; four times as many workgroups of a
// quarter of the work bring such a launch from ~23 us to ~10.  k-major operands only (all panel
// GEMMs are); same DMA + swizzle scheme, 2x2 wavefronts of 32 x 16*NBJ.  NBJ = 2: 64x64 tiles;
// NBJ = 4: 64x128 tiles, for the in-place calls  P <- P L_jj^-T  (C == A, N = 128): a workgroup
// must own whole rows there, or it would overwrite columns a neighbour is still reading.
template <int NBJ>
__global__ __launch_bounds__(256) void gemm_f64_mfma_dma64(GemmDev g) {
  constexpr int BNT = 32 * NBJ;
  __shared__ __attribute__((aligned(1024))) double sA[2][64 * BK];
  __shared__ __attribute__((aligned(1024))) double sB[2][BNT * BK];
  const long l = blockIdx.x;
  int tm, tn;
  if (g.lower) {        // the four 64x64 quarters of every lower 128x128 tile (the documented granularity)
    const long l4 = l >> 2;
    long t = (long)((sqrt(8.0 * (double)l4 + 1.0) - 1.0) * 0.5);
    while (t * (t + 1) / 2 > l4) --t;
    while ((t + 1) * (t + 2) / 2 <= l4) ++t;
    tm = 2 * (int)t + (int)((l >> 1) & 1);
    tn = 2 * (int)(l4 - t * (t + 1) / 2) + (int)(l & 1);
  } else {
    tm = (int)(l / g.tiles_n); tn = (int)(l % g.tiles_n);
  }
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const long row0 = (long)tm * 64, col0 = (long)tn * BNT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const long nk = g.K / BK;
  // DMA: instruction i (of 2) of this wavefront moves rows wave*16 + 8i + (lane>>3), piece lane&7
  const double* ga[2];
  const double* gb[NBJ];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 16 + i * 8 + (lane >> 3);
    ga[i] = g.A + (row0 + r) * g.lda + (((lane & 7) ^ ((r >> 1) & 7)) * 2);
  }
#pragma unroll
  for (int i = 0; i < NBJ; ++i) {
    const int r = wave * 8 * NBJ + i * 8 + (lane >> 3);
    gb[i] = g.B + (col0 + r) * g.ldb + (((lane & 7) ^ ((r >> 1) & 7)) * 2);
  }
  const int sw = (fr >> 1) & 7;
  int offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = GH_KM_OFFK(kk, fk, sw);
  const int rowA = (wm * 32 + fr) * BK, rowB = (wn * 16 * NBJ + fr) * BK;
  const int dstA = wave * 2 * 128, dstB = wave * NBJ * 128;
#define GH_DMA64_ISSUE(buf)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
    __builtin_amdgcn_global_load_lds((gh_glb_void*)ga[i], (gh_lds_void*)(sA[buf] + dstA + i * 128), 16, 0, 0); \
    ga[i] += BK;                                                                                       \
  }                                                                                                    \
  _Pragma("unroll") for (int i = 0; i < NBJ; ++i) {                                                    \
    __builtin_amdgcn_global_load_lds((gh_glb_void*)gb[i], (gh_lds_void*)(sB[buf] + dstB + i * 128), 16, 0, 0); \
    gb[i] += BK;                                                                                       \
  }
  v4d acc[2][NBJ];
  const double alpha = g.alpha;
  double* cbase = g.C + (row0 + wm * 32 + fk) * g.ldc + col0 + wn * 16 * NBJ + fr;
  if (nk > 0) { GH_DMA64_ISSUE(0) }
  if (g.preload) {
    const double rho = (g.beta == g.alpha) ? 1.0 : -1.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) acc[i][j][r] = rho * cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();
  for (long kt = 0; kt < nk; ++kt) {
    const int cur = (int)(kt & 1);
    const double* pa = sA[cur] + rowA;
    const double* pb = sB[cur] + rowB;
    double a[4][2], b[4][NBJ];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[kk][i] = pa[i * 16 * BK + offk[kk]];
#pragma unroll
      for (int j = 0; j < NBJ; ++j) b[kk][j] = pb[j * 16 * BK + offk[kk]];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      if (cur) { GH_DMA64_ISSUE(0) } else { GH_DMA64_ISSUE(1) }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
#undef GH_DMA64_ISSUE
  const double beta = g.preload ? 0.0 : g.beta;
  if (beta == 0.0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = alpha * acc[i][j][r];
  } else {
    double c[2][4][NBJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) c[i][r][j] = cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
          cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16] = fma(beta, c[i][r][j], alpha * acc[i][j][r]);
  }
}

// ---------------------------------------------------------------------------------------------
// K = 128, whole K resident: the two GEMMs of every link of the panel chain (rows <- rows L_jj^-T,
// in place, and the update of the panel's other columns) have K = 128 and a grid far below the chip.
// In gemm_f64_mfma_dma64 they took 16-18 us for 3.4-6.9 us of MFMA: eight K steps, each of them
// waiting for a DMA that was issued one step earlier (~1.5 us of L2 latency per step).  Here all eight
// 16-deep K chunks of both operands are requested AT ONCE into eight LDS buffers (64..144 KB of the
// CU's 160 KB): one memory latency for the whole tile instead of eight.  (The code waits for chunk c
// by count, s_waitcnt vmcnt((7 - c) * IPC); as compiled, hipcc's LDS-DMA tracking puts a vmcnt(0) in
// front of the first barrier, so the MFMAs start when the last chunk has landed.  Lifting that was tried
// (raw s_barrier, fragment reads and the C preload as inline asm with their own waits: MFMAs of chunk 0
// under the DMAs of chunks 1-7): bit-identical results, no gain -- 3.10 / 7.77 / 31.4 ms at N = 4096 /
// 8192 / 16384 against 3.0 / 7.6 / 31.1.  C requested behind the DMAs and
// added in the epilogue instead of starting the accumulators from it: measured, no gain, and the
// sums are then no longer bit-identical to the other GEMM kernels'.)
// Tile (16 MI WM) x (16 NJ WN), WM x WN = 4 wavefronts of (16 MI) x (16 NJ); k-major operands, same
// DMA + swizzle scheme as above.  <2,2,2,2>: 64 x 64;  <1,4,1,2>: 16 x 128 for the in-place call
// (a workgroup owns whole rows, see above).
template <int WM, int WN, int MI, int NJ>
__global__ __launch_bounds__(256) void gemm_f64_mfma_k128(GemmDev g) {
  constexpr int TR = 16 * MI * WM, TC = 16 * NJ * WN;
  constexpr int IA = (TR / 8 + 3) / 4, IB = (TC / 8 + 3) / 4, IPC = IA + IB;   // DMA instructions per chunk per wavefront
  __shared__ __attribute__((aligned(1024))) double sA[8][TR * BK];
  __shared__ __attribute__((aligned(1024))) double sB[8][TC * BK];
  const int tm = (int)(blockIdx.x / g.tiles_n), tn = (int)(blockIdx.x % g.tiles_n);
  if (g.lower && (long)tn * TC > (long)tm * TR + TR - 1) return;         // (uniform) tile strictly above the diagonal
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const long row0 = (long)tm * TR, col0 = (long)tn * TC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fk = lane >> 4;
  double* cbase = g.C + (row0 + wm * 16 * MI + fk) * g.ldc + col0 + wn * 16 * NJ + fr;
  v4d acc[MI][NJ];
  if (g.preload) {                              // (before the DMAs: older in the vmcnt order)
    const double rho = (g.beta == g.alpha) ? 1.0 : -1.0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j][r] = rho * cbase[(long)(i * 16 + 4 * r) * g.ldc + j * 16];
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  // every wavefront issues IA + IB instructions per chunk (a tile with fewer 8-row groups than
  // wavefronts has some moved twice: the counts must agree for the vmcnt arithmetic)
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
#pragma unroll
    for (int q = 0; q < IA; ++q) {
      const int idx = (wave + 4 * q) % (TR / 8), r = idx * 8 + (lane >> 3);
      const double* ga = g.A + (row0 + r) * g.lda + kc * BK + (((lane & 7) ^ ((r >> 1) & 7)) * 2);
      __builtin_amdgcn_global_load_lds((gh_glb_void*)ga, (gh_lds_void*)(sA[kc] + idx * 128), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < IB; ++q) {
      const int idx = (wave + 4 * q) % (TC / 8), r = idx * 8 + (lane >> 3);
      const double* gb = g.B + (col0 + r) * g.ldb + kc * BK + (((lane & 7) ^ ((r >> 1) & 7)) * 2);
      __builtin_amdgcn_global_load_lds((gh_glb_void*)gb, (gh_lds_void*)(sB[kc] + idx * 128), 16, 0, 0);
    }
  }
  const int sw = (fr >> 1) & 7;
  int offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = GH_KM_OFFK(kk, fk, sw);
  const int rowA = (wm * 16 * MI + fr) * BK, rowB = (wn * 16 * NJ + fr) * BK;
#define GH_K128_CHUNK(kc)                                                                              \
  do {                                                                                                 \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((7 - (kc)) * IPC) : "memory");                            \
    __syncthreads();                                                                                   \
    double a[4][MI], b[4][NJ];                                                                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                 \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) a[kk][i] = sA[kc][rowA + i * 16 * BK + offk[kk]]; \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) b[kk][j] = sB[kc][rowB + j * 16 * BK + offk[kk]]; \
    }                                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                   \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                 \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);    \
  } while (0)
  GH_K128_CHUNK(0); GH_K128_CHUNK(1); GH_K128_CHUNK(2); GH_K128_CHUNK(3);
  GH_K128_CHUNK(4); GH_K128_CHUNK(5); GH_K128_CHUNK(6); GH_K128_CHUNK(7);
#undef GH_K128_CHUNK
  const double alpha = g.alpha, beta = g.preload ? 0.0 : g.beta;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        double* c = cbase + (long)(i * 16 + 4 * r) * g.ldc + j * 16;
        *c = (beta == 0.0) ? alpha * acc[i][j][r] : fma(beta, *c, alpha * acc[i][j][r]);
      }
}

// Plain-VALU kernel with identical semantics: validation arm for the MFMA lane maps
// (GEORGE_AMD_NO_MFMA=1) -- each thread owns an 8x8 micro-tile of the 128x128 C tile.
template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f64_valu(GemmDev g) {
  __shared__ double sA[BM * LS];
  __shared__ double sB[BN * LS];
  int tm, tn;
  if (!tile_of(g, tm, tn)) return;       // (uniform per workgroup, before any barrier)
  const long row0 = (long)tm * BM, col0 = (long)tn * BN;
  long kbeg, kend;
  k_range(g, row0, col0, kbeg, kend);
  const int tid = threadIdx.x;
  const int tr = (tid >> 4) * 8, tc = (tid & 15) * 8;
  double acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
  double2 ra[4], rb[4];
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    stage_load<A_KM>(g.A, g.lda, row0, k0, tid, ra);
    stage_load<B_KM>(g.B, g.ldb, col0, k0, tid, rb);
    __syncthreads();
    stage_store<A_KM>(sA, tid, ra);
    stage_store<B_KM>(sB, tid, rb);
    __syncthreads();
    for (int k = 0; k < BK; ++k) {
      double a[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = sA[(tr + i) * LS + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = sB[(tc + j) * LS + k];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
  }
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      double* c = g.C + (row0 + tr + i) * g.ldc + col0 + tc + j;
      const double v = g.alpha * acc[i][j];
      *c = (g.beta == 0.0) ? v : g.beta * (*c) + v;
    }
}

// 0 = plain VALU (validation arm: cross-checks the MFMA lane maps on the device), 1 = the LDS-DMA
// v_mfma_f64_16x16x4 kernels (default).  GEORGE_AMD_NO_MFMA=1 / gh_debug_set_mfma(0).
static int g_mfma = -1;
static int mfma_mode() {
  if (g_mfma < 0) g_mfma = getenv("GEORGE_AMD_NO_MFMA") ? 0 : 1;
  return g_mfma;
}
bool gh_use_mfma() { return mfma_mode() != 0; }
extern "C" int gh_debug_set_mfma(int mode) {
  const int prev = mfma_mode();
  g_mfma = mode == 0 ? 0 : 1;
  return prev;
}

// the half-slab software-pipelined form of the k-major x k-major kernel (gemm_f64_mfma_dma_sp): 1 = on for every launch it
// can take (the default: 1.3 - 2.6 % faster on every shape of the factorisation, 72.4 TFLOP/s on the SYRK shape and 75.5 at
// K = 4096, profiles/r05/gemm_sp_ab.md), 0 = off.  gh_debug_set_gemm_sp (the bit-compare test and A/B scripts); same bits either way.
static int g_gemm_sp = -1;
static int gemm_sp_mode() {
  if (g_gemm_sp < 0) g_gemm_sp = GH_GEMM_SP_DEFAULT;
  return g_gemm_sp;
}
// tile order of the launches without k clipping: -1 by size (above), 0 row-major, 1 the 8-row groups walked column-major
static int g_gemm_grouped = -1;
extern "C" int gh_debug_set_gemm_grouped(int mode) {
  const int prev = g_gemm_grouped;
  g_gemm_grouped = mode < 0 ? -1 : (mode != 0);
  return prev;
}
extern "C" int gh_debug_set_gemm_sp(int mode) {
  const int prev = gemm_sp_mode();
  g_gemm_sp = mode < 0 ? GH_GEMM_SP_DEFAULT : (mode != 0);
  return prev;
}

int gh_launch_gemm(const GhGemm& h, hipStream_t st) {
  if (h.M <= 0 || h.N <= 0) return GH_OK;
  if (h.M % BM || h.N % BN || h.K % BK) { gh_set_error("gemm: sizes must be multiples of the tile (%ld %ld %ld)", (long)h.M, (long)h.N, (long)h.K); return GH_ERR_BAD_ARG; }
  if (h.lower && h.M < h.N) { gh_set_error("gemm: lower needs M >= N (a lower trapezoid: the leading N columns of a lower-triangular C)"); return GH_ERR_BAD_ARG; }
  GemmDev g;
  g.C = h.C; g.ldc = h.ldc; g.A = h.A; g.lda = h.lda; g.B = h.B; g.ldb = h.ldb; g.K = h.K;
  g.alpha = h.alpha; g.beta = h.beta;
  g.tiles_m = (int)(h.M / BM); g.tiles_n = (int)(h.N / BN);
  g.lower = h.lower; g.klo_max = h.klo_max; g.khi_col = h.khi_col; g.khi_row = h.khi_row;
  g.nblk = h.lower ? (long)g.tiles_n * (g.tiles_n + 1) / 2 + (long)(g.tiles_m - g.tiles_n) * g.tiles_n : (long)g.tiles_m * g.tiles_n;
  g.stair_n = 0; g.stair_h = 0;
  if (h.stair_n > 0) {
    // (validated by gh_dev_gemm_nt_stair: k-major operands, no k clipping, groups of whole tiles, at most GH_GEMM_STAIR_MAX of them)
    g.stair_n = h.stair_n; g.stair_h = (int)(h.stair_rows / BM);
    g.tiles_m = g.stair_n * g.stair_h; g.tiles_n = (int)(h.stair_cols[h.stair_n - 1] / BN);
    long at = 0;
    for (int s = 0; s < h.stair_n; ++s) { g.stair_pre[s] = (unsigned)at; at += (long)g.stair_h * (h.stair_cols[h.stair_n - 1 - s] / BN); }
    g.stair_pre[h.stair_n] = (unsigned)at;
    g.nblk = at;
  }
  g.prio = g.nblk <= 512 ? 1 : 0;
  // (grouped tile order only once the column operand outgrows what the 256 MB MALL keeps between tile rows -- 128 MiB, N > 16384
  //  at K = 1024: below that a row-major walk already finds its slabs on chip and is 1 % faster, profiles/r04/gemm_tile_order_ab.md)
  g.grouped = (!(h.klo_max | h.khi_col | h.khi_row) && (long)h.N * h.K * 8 > (128L << 20)) ? 1 : 0;
  if (g_gemm_grouped >= 0 && !(h.klo_max | h.khi_col | h.khi_row)) g.grouped = g_gemm_grouped;      // (A/B: gh_debug_set_gemm_grouped)
  g.preload = (h.beta != 0.0 && (h.beta == h.alpha || h.beta == -h.alpha)) ? 1 : 0;
  if (g.nblk > 0x7fffffffL) { gh_set_error("gemm: grid too large"); return GH_ERR_BAD_ARG; }
  const dim3 grid((unsigned)g.nblk), block(256);
  const int mode = mfma_mode();
  // (operands the LDS-DMA path cannot take -- an odd leading dimension or a base that is not 16-byte aligned; none of
  //  the solver's own calls -- go through the plain-VALU kernel)
#define GH_GEMM_LAUNCH(AK, BKM) hipLaunchKernelGGL((gemm_f64_valu<AK, BKM>), grid, block, 0, st, g)
  // (... or a row pitch so large that the lane's byte offset inside a slab -- up to 127 rows x ld x 8 bytes, the 32-bit
  //  voffset of buffer_load ... lds -- would pass 2^31: ld >= 2 M doubles, sixteen times the largest matrix that fits in HBM)
  constexpr int64_t GH_DMA_MAX_LD = ((1LL << 31) - 4096) / (128 * 8);
  const bool dma = mode == 1 && h.lda % 2 == 0 && h.ldb % 2 == 0 && h.lda <= GH_DMA_MAX_LD && h.ldb <= GH_DMA_MAX_LD &&
                   ((uintptr_t)h.A % 16) == 0 && ((uintptr_t)h.B % 16) == 0;
#define GH_DMA_LAUNCH(AK, BKM)                                                                            \
  do {                                                                                                   \
    if (h.lower) hipLaunchKernelGGL((gemm_f64_mfma_dma<AK, BKM, true>), grid, block, 0, st, g);          \
    else         hipLaunchKernelGGL((gemm_f64_mfma_dma<AK, BKM, false>), grid, block, 0, st, g);         \
  } while (0)
  const bool inplace = (const double*)h.C == h.A || (const double*)h.C == h.B;
  if (dma && !h.stair_n && h.a_km && h.b_km && h.K == 128 && g.nblk <= 128 && !h.klo_max && !h.khi_col && !h.khi_row && !h.small_lds &&
      (!inplace || ((const double*)h.C == h.A && (const double*)h.C != h.B && h.N == 128 && !h.lower))) {
    GemmDev q = g;
    if (inplace) {                      // whole rows per workgroup: 16 x 128 tiles
      q.tiles_m = (int)(h.M / 16); q.tiles_n = 1;
      q.nblk = q.tiles_m;
      hipLaunchKernelGGL((gemm_f64_mfma_k128<1, 4, 1, 2>), dim3((unsigned)q.nblk), block, 0, st, q);
    } else {                            // 64 x 64 or 32 x 32 tiles (of a lower-triangular C: those that touch the triangle)
      // (32 x 32 and 64 x 32 tiles were measured too: 64 x 64 wins)
      q.tiles_m = (int)(h.M / 64); q.tiles_n = (int)(h.N / 64);
      q.nblk = (long)q.tiles_m * q.tiles_n;
      hipLaunchKernelGGL((gemm_f64_mfma_k128<2, 2, 2, 2>), dim3((unsigned)q.nblk), block, 0, st, q);
    }
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  if (dma && !h.stair_n && h.a_km && h.b_km && g.nblk <= 128 && !h.klo_max && !h.khi_col && !h.khi_row && (!h.lower || h.M == h.N) &&
      (!inplace || (h.N == 128 && !h.lower))) {
    // sub-chip launch: 64-row tiles, 2-4x the workgroups (see gemm_f64_mfma_dma64)
    GemmDev q = g;
    if (inplace) {                      // whole rows per workgroup: 64 x 128 tiles
      q.tiles_m = (int)(h.M / 64); q.tiles_n = 1;
      q.nblk = q.tiles_m;
      hipLaunchKernelGGL(gemm_f64_mfma_dma64<4>, dim3((unsigned)q.nblk), block, 0, st, q);
    } else {
      q.tiles_m = (int)(h.M / 64); q.tiles_n = (int)(h.N / 64);
      q.nblk = h.lower ? 4 * g.nblk : (long)q.tiles_m * q.tiles_n;
      hipLaunchKernelGGL(gemm_f64_mfma_dma64<2>, dim3((unsigned)q.nblk), block, 0, st, q);
    }
  }
  else if (dma && h.a_km && h.b_km && gemm_sp_mode() && !h.klo_max && !h.khi_col && !h.khi_row && h.K % (2 * BK) == 0) {
    if (h.lower) hipLaunchKernelGGL(gemm_f64_mfma_dma_sp<true>, grid, block, 0, st, g);
    else         hipLaunchKernelGGL(gemm_f64_mfma_dma_sp<false>, grid, block, 0, st, g);
  }
  else if (dma && h.a_km && h.b_km)   GH_DMA_LAUNCH(true, true);
  else if (dma && h.a_km && !h.b_km)  GH_DMA_LAUNCH(true, false);
  else if (dma && !h.a_km && !h.b_km) GH_DMA_LAUNCH(false, false);
  else if (dma)                       GH_DMA_LAUNCH(false, true);
#undef GH_DMA_LAUNCH
  else if (h.a_km && h.b_km) GH_GEMM_LAUNCH(true, true);
  else if (h.a_km && !h.b_km) GH_GEMM_LAUNCH(true, false);
  else if (!h.a_km && !h.b_km) GH_GEMM_LAUNCH(false, false);
  else GH_GEMM_LAUNCH(false, true);
#undef GH_GEMM_LAUNCH
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// One launch for a staircase-shaped C: row group g (group_rows rows of c and a) is updated over columns [0, group_cols[g]).
// The per-rank trailing update of the multi-GPU solvers (whole tile rows per rank: each row reaches as far as its own diagonal
// tile) was one launch per tile row -- 8 launches of ~1.3 chip-fulls each per step at N = 65536 on 8 ranks; as ONE grid of
// ~11 chip-fulls the tails of seven launches disappear.  Widest group first; inside a group column by column (the 8 x 8 tile
// blocks of tile_of's grouped order).  Same kernel, same per-tile arithmetic: bit-identical to the per-group launches.
extern "C" int gh_dev_gemm_nt_stair(double* c, int64_t ldc, const double* a, int64_t lda, const double* b, int64_t ldb,
                                    int64_t group_rows, int32_t n_groups, const int64_t* group_cols, int64_t k, void* stream) {
  if (n_groups <= 0) return GH_OK;
  if (!group_cols || group_rows <= 0 || group_rows % BM) { gh_set_error("gemm_nt_stair: group_rows must be a positive multiple of 128"); return GH_ERR_BAD_ARG; }
  long tiles = 0;
  for (int g = 0; g < n_groups; ++g) {
    if (group_cols[g] <= 0 || group_cols[g] % BN || (g > 0 && group_cols[g] < group_cols[g - 1])) {
      gh_set_error("gemm_nt_stair: group_cols must be positive multiples of 128, non-decreasing"); return GH_ERR_BAD_ARG; }
    tiles += (group_rows / BM) * (group_cols[g] / BN);
  }
  const bool aligned = lda % 2 == 0 && ldb % 2 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
  if (n_groups > GH_GEMM_STAIR_MAX || tiles <= 128 || tiles > 0x7fffffffL || !aligned || mfma_mode() != 1) {
    // (too many groups for the kernel's table, a sub-chip launch that the 64-row-tile kernels serve better, or operands the
    //  LDS-DMA kernel cannot take: one launch per group)
    for (int g = n_groups - 1; g >= 0; --g)
      GH_CHECK(gh_dev_gemm_nt(c + (int64_t)g * group_rows * ldc, ldc, a + (int64_t)g * group_rows * lda, lda, b, ldb, group_rows, group_cols[g], k, 0, stream));
    return GH_OK;
  }
  GhGemm h{};
  h.C = c; h.ldc = ldc; h.A = a; h.lda = lda; h.B = b; h.ldb = ldb;
  h.M = (int64_t)n_groups * group_rows; h.N = group_cols[n_groups - 1]; h.K = k;
  h.alpha = -1.0; h.beta = 1.0; h.a_km = true; h.b_km = true;
  h.stair_n = n_groups; h.stair_rows = group_rows; h.stair_cols = group_cols;
  return gh_launch_gemm(h, (hipStream_t)stream);
}

extern "C" int gh_dev_gemm_nt(double* c, int64_t ldc, const double* a, int64_t lda,
                              const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k,
                              int32_t lower, void* stream) {
  GhGemm g{};
  g.C = c; g.ldc = ldc; g.A = a; g.lda = lda; g.B = b; g.ldb = ldb;
  g.M = m; g.N = n; g.K = k; g.alpha = -1.0; g.beta = 1.0;
  g.a_km = true; g.b_km = true; g.lower = lower != 0;
  return gh_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int gh_dev_gemm(double* c, int64_t ldc, const double* a, int64_t lda,
                           const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k,
                           double alpha, double beta, int32_t flags, void* stream) {
  GhGemm g{};
  g.C = c; g.ldc = ldc; g.A = a; g.lda = lda; g.B = b; g.ldb = ldb;
  g.M = m; g.N = n; g.K = k; g.alpha = alpha; g.beta = beta;
  g.a_km = !(flags & GH_GEMM_A_MMAJOR); g.b_km = !(flags & GH_GEMM_B_NMAJOR);
  g.lower = (flags & GH_GEMM_LOWER) != 0; g.klo_max = (flags & GH_GEMM_KLO_MAX) != 0;
  g.khi_col = (flags & GH_GEMM_KHI_COL) != 0; g.khi_row = (flags & GH_GEMM_KHI_ROW) != 0;
  return gh_launch_gemm(g, (hipStream_t)stream);
}

// ------------------------------------------------------------ micro-benchmarks
// fp64 MFMA issue-rate ceiling: 4 independent accumulators per wavefront, 8 waves/CU.
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double* out, int iters) {
  v4d acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  const v4d s = acc0 + acc1 + acc2 + acc3;
  if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];   // keep the chain live
}
extern "C" int gh_microbench_mfma_f64(double* tflops_out) {
  if (gh_device_count() <= 0) { gh_set_error("no HIP device"); return GH_ERR_HIP; }
  double* d = nullptr;
  GH_HIP(hipMalloc((void**)&d, 64));
  const int iters = 20000, blocks = 256 * 2;
  hipEvent_t e0, e1;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, d, 100);
  GH_HIP(hipDeviceSynchronize());
  GH_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, d, iters);
  GH_HIP(hipEventRecord(e1, 0));
  GH_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GH_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 2048.0;
  *tflops_out = flops / (ms * 1e-3) * 1e-12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
  return GH_OK;
}
// ---- instruction-level suite: pins the fp64 ceilings the roofline is priced against ----
// MODE 0: v_mfma_f64_16x16x4_f64, 4 independent accumulators; MODE 1: v_fma_f64, 8 independent
// chains per lane; MODE 2: v_mfma_f64_4x4x4_4b_f64.  stats[0]=shader cycles (s_memtime),
// stats[1]=100 MHz wall ticks, both for workgroup 0 / wavefront 0.
template <int MODE>
__global__ __launch_bounds__(256) void fp64_rate_kernel(double* out, long long* stats, int iters) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
  const double a2 = a + 0.5, b2 = b - 0.5;
  double f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, f4 = a + 2, f5 = b + 2, f6 = a + 3, f7 = b + 3;
  double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {     // 8 independent accumulators, 2x2 distinct operand registers (as in a GEMM tile)
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, m1, 0, 0, 0);
      m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, m2, 0, 0, 0);
      m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, m3, 0, 0, 0);
      m4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m4, 0, 0, 0);
      m5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, m5, 0, 0, 0);
      m6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, m6, 0, 0, 0);
      m7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, m7, 0, 0, 0);
    } else if (MODE == 1) {
      f0 = fma(f0, a, b); f1 = fma(f1, a, b); f2 = fma(f2, a, b); f3 = fma(f3, a, b);
      f4 = fma(f4, a, b); f5 = fma(f5, a, b); f6 = fma(f6, a, b); f7 = fma(f7, a, b);
    } else {
      q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, q3, 0, 0, 0);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = c1 - c0; stats[1] = w1 - w0; }
  const v4d s = m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7;
  const double t = s[0] + s[1] + s[2] + s[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + q0 + q1 + q2 + q3;
  if (t == 12345.678) out[0] = t;
}
template <int MODE>
static int run_rate(int blocks, int iters, double flop_per_wave_iter, double* tf, double* cyc_per_instr, double* ghz,
                    double instr_per_iter) {
  double* d = nullptr; long long* st = nullptr;
  GH_HIP(hipMalloc((void**)&d, 64));
  GH_HIP(hipMalloc((void**)&st, 64));
  hipEvent_t e0, e1;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(fp64_rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 200);
  GH_HIP(hipDeviceSynchronize());
  GH_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(fp64_rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, iters);
  GH_HIP(hipEventRecord(e1, 0));
  GH_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GH_HIP(hipEventElapsedTime(&ms, e0, e1));
  long long h[2] = {0, 0};
  GH_HIP(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
  *tf = (double)blocks * 4.0 * iters * flop_per_wave_iter / (ms * 1e-3) * 1e-12;
  *cyc_per_instr = (double)h[0] / ((double)iters * instr_per_iter);
  *ghz = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;    // cycles per ns (wall tick = 10 ns)
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); (void)hipFree(st);
  return GH_OK;
}
extern "C" int gh_microbench_suite(double* out, int n) {
  if (gh_device_count() <= 0) { gh_set_error("no HIP device"); return GH_ERR_HIP; }
  if (!out || n < 16) { gh_set_error("need room for 16 doubles"); return GH_ERR_BAD_ARG; }
  for (int i = 0; i < n; ++i) out[i] = 0.0;
  double tf, cyc, ghz;
  const int iters = 20000;
  // f64 MFMA 16x16x4 at 1, 2, 4 wavefronts per SIMD (256 / 512 / 1024 workgroups of 4 waves on 256 CUs)
  GH_CHECK(run_rate<0>(256, iters, 8 * 2048.0, &tf, &cyc, &ghz, 8)); out[0] = tf; out[1] = cyc; out[2] = ghz;
  GH_CHECK(run_rate<0>(512, iters, 8 * 2048.0, &tf, &cyc, &ghz, 8)); out[3] = tf; out[4] = cyc; out[5] = ghz;
  GH_CHECK(run_rate<0>(1024, iters, 8 * 2048.0, &tf, &cyc, &ghz, 8)); out[6] = tf;
  // v_fma_f64 at 4 and 8 wavefronts per SIMD
  GH_CHECK(run_rate<1>(1024, iters * 4, 8 * 128.0, &tf, &cyc, &ghz, 8)); out[7] = tf; out[8] = cyc; out[9] = ghz;
  GH_CHECK(run_rate<1>(2048, iters * 4, 8 * 128.0, &tf, &cyc, &ghz, 8)); out[10] = tf;
  // f64 MFMA 4x4x4 (4 blocks): 4*4*4*4*2 = 512 flop per instruction
  GH_CHECK(run_rate<2>(512, iters, 4 * 512.0, &tf, &cyc, &ghz, 4)); out[11] = tf; out[12] = cyc;
  return GH_OK;
}

// ---- the fp64 matrix-pipe CEILING (SURVEY 8d's second roofline denominator) ------------------------------------------
// A bare v_mfma_f64_16x16x4_f64 issue loop that is limited by the pipe and nothing else.  The suite above reads 34-47 TFLOP/s
// -- BELOW what the GEMM kernel sustains -- and is NOT a ceiling: written with the builtin, its loop is compiled into eight
// matrix instructions wrapped in 128 v_accvgpr_write / v_accvgpr_read moves per iteration (hipcc parks the accumulators in
// VGPRs across iterations and shuttles them through AGPRs), so it measures the vector ALU's move rate (visible with
// `hipcc -S --cuda-device-only`; profiles/r04/README.md).  Here the loop body is inline assembly with the accumulators pinned
// to VGPRs: eight independent matrix instructions and a scalar loop counter, nothing else.  Also: 64x more workgroups than
// slots, each short, so that the dispatcher's refill evens the load as it does for a GEMM grid, and the resident wavefronts
// per SIMD pinned by a dynamic LDS request (96 KiB -> one 4-wavefront workgroup per CU = 1 wavefront per SIMD, 64 KiB -> 2,
// 40 KiB -> 4; 160 KiB per CU).  One instruction occupies a SIMD's pipe for 64 cycles (scripts/dev/valu_probe.hip), so the
// ceiling is 256 CU x 4 SIMD x 2048 flop / 64 cycles x the clock the chip holds under that load -- which is what this measures.
extern __shared__ double mfma_ceiling_lds[];
__global__ __launch_bounds__(256) void mfma_f64_ceiling_kernel(double* out, int iters) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9, a2 = a + 0.5, b2 = b - 0.5;
  v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
  // Inline assembly, accumulators pinned to VGPRs: written with the builtin, hipcc keeps the eight accumulators in VGPRs ACROSS
  // iterations and copies all 64 registers into AGPRs and back around the eight MFMAs of every iteration (128 v_accvgpr moves
  // per 8 matrix instructions) -- THAT is what round 3's loops measured (34-47 TFLOP/s), and this kernel's first form too
  // (35 / 48 / 51); the GEMM kernel, whose accumulators live in one place, sustains 70.
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %8, %9, %0\n\t"
                 "v_mfma_f64_16x16x4_f64 %1, %8, %11, %1\n\t"
                 "v_mfma_f64_16x16x4_f64 %2, %10, %9, %2\n\t"
                 "v_mfma_f64_16x16x4_f64 %3, %10, %11, %3\n\t"
                 "v_mfma_f64_16x16x4_f64 %4, %8, %9, %4\n\t"
                 "v_mfma_f64_16x16x4_f64 %5, %8, %11, %5\n\t"
                 "v_mfma_f64_16x16x4_f64 %6, %10, %9, %6\n\t"
                 "v_mfma_f64_16x16x4_f64 %7, %10, %11, %7"
                 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7)
                 : "v"(a), "v"(b), "v"(a2), "v"(b2));
  }
  const v4d s = m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7;
  if (s[0] + s[1] + s[2] + s[3] == 12345.678) { out[0] = s[0]; mfma_ceiling_lds[threadIdx.x] = s[1]; }   // keep the chain (and the LDS) live
}
// out[0..2] = TFLOP/s at 1 / 2 / 4 resident wavefronts per SIMD, out[3] = the best of them, out[4..6] = milliseconds (n >= 8)
extern "C" int gh_microbench_mfma_f64_ceiling(double* out, int n) {
  if (gh_device_count() <= 0) { gh_set_error("no HIP device"); return GH_ERR_HIP; }
  if (!out || n < 8) { gh_set_error("need room for 8 doubles"); return GH_ERR_BAD_ARG; }
  double* d = nullptr;
  GH_HIP(hipMalloc((void**)&d, 64));
  GH_HIP(hipFuncSetAttribute((const void*)mfma_f64_ceiling_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipEvent_t e0, e1;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  GH_HIP(hipGetDeviceProperties(&prop, 0));
  int dev = 0;
  (void)hipGetDevice(&dev);
  GH_HIP(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int lds[3] = {96 * 1024, 64 * 1024, 40 * 1024}, per_cu[3] = {1, 2, 4};
  out[3] = 0.0;
  for (int v = 0; v < 3; ++v) {
    const int iters = 600, blocks = ncu * per_cu[v] * 64;
    hipLaunchKernelGGL(mfma_f64_ceiling_kernel, dim3(ncu * per_cu[v]), dim3(256), lds[v], 0, d, 50);
    GH_HIP(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      GH_HIP(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(mfma_f64_ceiling_kernel, dim3(blocks), dim3(256), lds[v], 0, d, iters);
      GH_HIP(hipEventRecord(e1, 0));
      GH_HIP(hipEventSynchronize(e1));
      float ms = 0;
      GH_HIP(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    GH_HIP(hipGetLastError());
    out[v] = (double)blocks * 4.0 * iters * 8.0 * 2048.0 / (best * 1e-3) * 1e-12;
    out[4 + v] = best;
    out[3] = std::max(out[3], out[v]);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
  return GH_OK;
}

__global__ void copy16_kernel(const double2* __restrict__ in, double2* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// four 16-byte loads in flight per lane before the first store, non-temporal stores (the copy is
// streamed once: no point in keeping it in the L2s)
__global__ __launch_bounds__(256) void copy16x4_kernel(const double2* __restrict__ in, double2* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long base = (long)blockIdx.x * 256 * 4 + threadIdx.x; base < n; base += stride) {
    double2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (base + 256 * q < n) ? in[base + 256 * q] : double2{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (base + 256 * q < n) {
        __builtin_nontemporal_store(v[q].x, &out[base + 256 * q].x);
        __builtin_nontemporal_store(v[q].y, &out[base + 256 * q].y);
      }
  }
}
// Best of a few launch shapes of a 2 GiB -> 2 GiB device copy, read + written bytes per second.
// (The plain grid-stride copy16_kernel with 2048 workgroups gives 4.6-4.9 TB/s; it stays the kernel
// the FETCH_SIZE / WRITE_SIZE calibration passes of scripts/profile_r02.sh look for.)
extern "C" int gh_microbench_hbm_copy(double* gbps_out) {
  if (gh_device_count() <= 0) { gh_set_error("no HIP device"); return GH_ERR_HIP; }
  const long n = 1L << 27;   // 2 GiB in + 2 GiB out
  double2 *a = nullptr, *b = nullptr;
  GH_HIP(hipMalloc((void**)&a, n * sizeof(double2)));
  GH_HIP(hipMalloc((void**)&b, n * sizeof(double2)));
  GH_HIP(hipMemset(a, 0, n * sizeof(double2)));
  hipEvent_t e0, e1;
  GH_HIP(hipEventCreate(&e0)); GH_HIP(hipEventCreate(&e1));
  double best = 0.0;
  for (int variant = 0; variant < 6; ++variant) {
    const int grids[6] = {256 * 8, 256 * 8, 256 * 16, 256 * 32, 256 * 64, 256 * 128};
    auto launch = [&]() {
      if (variant == 0) hipLaunchKernelGGL(copy16_kernel, dim3(grids[0]), dim3(256), 0, 0, a, b, n);
      else hipLaunchKernelGGL(copy16x4_kernel, dim3(grids[variant]), dim3(256), 0, 0, a, b, n);
    };
    launch();
    GH_HIP(hipDeviceSynchronize());
    GH_HIP(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) launch();
    GH_HIP(hipEventRecord(e1, 0));
    GH_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GH_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double rate = 5.0 * 2.0 * n * sizeof(double2) / (ms * 1e-3) * 1e-9;
    if (rate > best) best = rate;
  }
  *gbps_out = best;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
  return GH_OK;
}
