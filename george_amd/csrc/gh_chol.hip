// gh_chol.hip -- dense blocked Cholesky solver on one MI355X.
//
// Replaces BasicSolver (reference src/george/solvers/basic.py:51-121) and the
// SciPy/LAPACK dpotrf/dpotrs behind it.  The covariance matrix is built on the
// device (gh_kmat.hip), lives in HBM as one row-major Np x Np array (Np = N
// rounded up to 128, identity-padded) of which only the LOWER triangle is ever
// touched (K = L L^T; the reference's upper factor U is L^T), and is factorised
// in place:
//
//   for each outer panel of NB columns:
//     for each 128-column step of the panel:
//        potf2_inv   : 128x128 diagonal block -> L_jj and L_jj^-1 (one workgroup, all in LDS)
//        trsm (gemm) : rows below <- rows * L_jj^-T            (MFMA, in place)
//        update(gemm): remaining panel columns -= ...          (MFMA)
//     trailing SYRK   : A22 -= L21 L21^T, K = NB, lower tiles  (MFMA; >95 % of the flops)
//
// Solves are blocked substitutions that multiply by the stored 128x128 diagonal
// inverses: TRSV kernels for one right-hand side, the same MFMA GEMM for many.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include "gh_common.h"
#include "gh_spin.h"
#include "../../include/george_amd_debug.h"

#define T 128                 // tile edge
#define LP 129                // LDS row pitch of the potf2 tile (odd -> conflict-free columns)

// ============================================================= potf2 + inverse
// One workgroup factorises a 128x128 block held entirely in LDS (129 KiB) and
// inverts the triangular factor; `dinv` receives L^-1 (row-major 128x128, zeros
// above the diagonal).  info: 0 = ok so far; set to base+j+1 at the first
// non-positive pivot (LAPACK dpotrf `info`), after which every later call is a no-op.
__global__ __launch_bounds__(256) void potf2_inv_kernel(double* A, long lda, double* dinv,
                                                        long long* info, long long base) {
  __shared__ double s[T * LP];
  __shared__ int fail_at;
  const int tid = threadIdx.x;
  if (*info != 0) return;               // uniform: an earlier block already failed
  if (tid == 0) fail_at = -1;
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    s[i * LP + j] = A[(long)i * lda + j];
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;       // 16x16 cyclic ownership of the trailing block
  for (int j = 0; j < T; ++j) {
    const double d = s[j * LP + j];
    if (!(d > 0.0)) {                            // also catches NaN
      if (tid == 0) fail_at = j;
      break;                                     // uniform: every thread read the same d
    }
    const double ajj = sqrt(d);
    __syncthreads();                             // everyone has read d before it is overwritten
    if (tid < T) {
      if (tid > j) s[tid * LP + j] /= ajj;
      else if (tid == j) s[j * LP + j] = ajj;
    }
    __syncthreads();
    // rank-1 update of the trailing lower triangle
    for (int i = j + 1 + ((ty - (j + 1)) & 15); i < T; i += 16) {
      const double lij = s[i * LP + j];
      for (int k = j + 1 + ((tx - (j + 1)) & 15); k <= i; k += 16)
        s[i * LP + k] -= lij * s[k * LP + j];
    }
    __syncthreads();
  }
  __syncthreads();
  if (fail_at >= 0) {
    if (tid == 0) *info = base + fail_at + 1;
    return;
  }
  // write the factor back, zeroing the strict upper triangle of the tile
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    A[(long)i * lda + j] = (j <= i) ? s[i * LP + j] : 0.0;
  }
  __syncthreads();
  // L^-1, column by column: lane pair (c, h) owns column c, h splits the dot product by
  // k parity; x_i (i > c) is kept in the unused upper triangle at s[c][i].  No workgroup
  // barrier: a column only depends on itself, and the two lanes sit in one wavefront
  // (LDS operations of a wavefront execute in program order; volatile stops the compiler
  // from caching or reordering them).
  {
    volatile double* vs = s;
    const int c = tid >> 1, h = tid & 1;
    const double xc = 1.0 / vs[c * LP + c];
    for (int i = c + 1; i < T; ++i) {
      double acc = (h == 0) ? vs[i * LP + c] * xc : 0.0;
      for (int k = c + 1 + h; k < i; k += 2) acc += vs[i * LP + k] * vs[c * LP + k];
      acc += __shfl_xor(acc, 1, 64);
      const double xi = -acc / vs[i * LP + i];
      if (h == 0) vs[c * LP + i] = xi;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx >> 7, j = idx & 127;       // dinv[i][j] = x_i of column j
    double v;
    if (j < i) v = s[j * LP + i];
    else if (j == i) v = 1.0 / s[i * LP + i];
    else v = 0.0;
    dinv[idx] = v;
  }
}

// ================================================================= reductions
__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double block_sum_256(double v, double* sh /* >= 4 */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// out[0] (+)= 2 * sum_i log(A[i][i])   (basic.py:69); one workgroup, fixed order
__global__ __launch_bounds__(256) void logdet_kernel(const double* A, long lda, long n, double* out, int accumulate) {
  __shared__ double sh[4];
  double v = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) v += log(A[i * lda + i]);
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + 2.0 * v;
}
// out[0] = sum_i a[i] * b[i]
// (fail != nullptr: the chained solve's time-out flag travels with the result, out[2] = flag: one copy back instead of two)
__global__ __launch_bounds__(256) void dot_kernel(const double* a, const double* b, long n, double* out, const int* fail) {
  __shared__ double sh[4];
  double v = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) v += a[i] * b[i];
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) { out[0] = v; if (fail) out[2] = (double)*fail; }
}

// Two-stage versions for long vectors: `part[g]` = the g-th contiguous slice, then one workgroup adds
// the slices in index order (fixed order: bitwise reproducible).  One workgroup walking 65536
// diagonal entries, each in its own cache line, took 195 us; the dot product 97 us.
__global__ __launch_bounds__(256) void logdet_part_kernel(const double* A, long lda, long n, double* part) {
  __shared__ double sh[4];
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) v += log(A[i * lda + i]);
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void dot_part_kernel(const double* a, const double* b, long n, double* part) {
  __shared__ double sh[4];
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) v += a[i] * b[i];
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}
// out[0] (+)= scale * sum_{g < m} part[g], m <= 64, added in index order by one lane
__global__ void reduce_final_kernel(const double* part, int m, double scale, double* out, int accumulate, const int* fail) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double v = 0.0;
  for (int g = 0; g < m; ++g) v += part[g];
  out[0] = (accumulate ? out[0] : 0.0) + scale * v;
  if (fail) out[2] = (double)*fail;
}
#define RED_SLICES 64
static int launch_logdet(const double* A, long lda, long n, double* out, double* part, hipStream_t st) {
  const int g = (int)std::min<long>(RED_SLICES, (n + 2047) / 2048);
  if (g <= 1) {
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, st, A, lda, n, out, 0);
  } else {
    hipLaunchKernelGGL(logdet_part_kernel, dim3(g), dim3(256), 0, st, A, lda, n, part);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, part, g, 2.0, out, 0, (const int*)nullptr);
  }
  GH_HIP(hipGetLastError());
  return GH_OK;
}
static int launch_dot(const double* a, const double* b, long n, double* out, double* part, hipStream_t st, const int* fail = nullptr) {
  const int g = (int)std::min<long>(RED_SLICES, (n + 4095) / 4096);
  if (g <= 1) {
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, a, b, n, out, fail);
  } else {
    hipLaunchKernelGGL(dot_part_kernel, dim3(g), dim3(256), 0, st, a, b, n, part);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, part, g, 1.0, out, 0, fail);
  }
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// ======================================================== single-RHS solves
// Forward step j of L z = y (right-looking).  Every workgroup recomputes
// z_j = L_jj^-1 w_j from the current working vector w (128x128 mat-vec from L2), workgroup 0
// publishes it into z, workgroups b >= 1 update their 128 rows: w[i] -= L[i, jblock] . z_j.
__global__ __launch_bounds__(256) void trsv_fwd_step(const double* L, long ld, const double* dinv_j,
                                                     long j0, double* w, double* z) {
  __shared__ double zj[T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double2 wv = *reinterpret_cast<const double2*>(w + j0 + 2 * lane);
  // 8 rows per trip: all eight 1-KiB row loads are in flight before the first reduction
  // (one load per trip left this kernel latency-bound: 52 us per step at N = 16384)
  for (int r0 = wave * 8; r0 < T; r0 += 32) {
    double2 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const double2*>(dinv_j + (r0 + q) * T + 2 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double v = wave_sum(a[q].x * wv.x + a[q].y * wv.y);
      if (lane == 0) zj[r0 + q] = v;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < T) z[j0 + tid] = zj[tid];
    return;
  }
  const long row0 = j0 + (long)blockIdx.x * T;
  const double zx = zj[2 * lane], zy = zj[2 * lane + 1];
  for (int r0 = wave * 8; r0 < T; r0 += 32) {
    double2 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const double2*>(L + (row0 + r0 + q) * ld + j0 + 2 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double v = wave_sum(a[q].x * zx + a[q].y * zy);
      if (lane == 0) w[row0 + r0 + q] -= v;
    }
  }
}
// The whole forward sweep L z = y as ONE launch: workgroup b owns block row b.  It walks the blocks
// L[b, 0..b-1] left to right, folding each z_j into per-lane partial sums as soon as workgroup j
// has published it, then solves its own diagonal block with the stored inverse and publishes z_b.
// The step-per-launch version above costs a launch gap plus two dependent 128x128 mat-vecs per block
// row (25-27 us, 14.3 ms at N = 65536 against 2.2 ms of HBM time for the triangle); here a link of the
// chain is  z_j seen -> 128 FMAs per lane -> reduce -> one mat-vec -> store,  and the L blocks of a
// row stream in ahead of the wait (the next block is loaded into registers before it).  512 threads:
// wavefront w takes rows 16w..16w+15, a lane two columns.  Deadlock freedom: workgroup b only waits
// for workgroups j < b, and a 1-D grid is dispatched in blockIdx order, so whatever it waits for is
// resident or finished (the grid need not fit the chip).  A wait that outlasts ~2 s raises *fail
// instead of hanging.
//
// z ITSELF IS THE MESSAGE (round 3; the flag-per-block-row predecessor, 10.7 us per link, is
// scripts/dev/arms/trsv_chain_flags.hip.inc).  z is pre-filled with a sentinel (all bits set: a NaN no
// arithmetic produces), workgroup j publishes its 128 values as agent-scope atomic stores (write-through
// past its XCD's L2; no fences: a release/acquire pair costs an L2 write-back on one side and an
// invalidate on the other at every link -- chain neighbours sit on different XCDs -- 16-29 us measured),
// and a consumer's first wavefront polls those 128 values directly -- lane l its two -- until none is
// the sentinel, then hands them to the other wavefronts through LDS.  Against the flag version a link
// loses one L2 round trip (flag seen -> THEN z fetched), the producer's s_waitcnt + barrier + flag
// store, the sixteen one-lane stores of a wavefront (now one 128-byte store from lanes 0-15) and 5/6 of
// its cross-lane traffic: the 16 row sums of a wavefront are formed by a transposing butterfly (8 + 4 +
// 2 + 1 exchanges inside a row of 16 lanes, then 2 across rows: 17 instead of 96), which leaves row q's
// total in lane q; y is fetched before the loop (it was a dependent load on the critical path).
// Measured: 3.4 us per link; the solve part of compute()+log_likelihood() 0.68 -> 0.22 ms at N = 8192,
// apply_inverse(y) 9.0 -> 6.6 ms at N = 65536 (two sweeps over 17 GB: 5.2 TB/s, 0.65 of HBM; was 0.39).
#define CHAIN_THREADS 512
__device__ __forceinline__ double2 ld_coherent2(const double* p) {
  double2 v;
  v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
#define CHAIN_SENTINEL 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ bool chain_ready(double v) { return (unsigned long long)__double_as_longlong(v) != CHAIN_SENTINEL; }
// v[0..15] per lane -> returns, in lane l, the sum over all 64 lanes of v[l & 15]
__device__ __forceinline__ double transpose_sum16(double (&v)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool up = (lane & 8) != 0;
    const double keep = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8];
    v[i] = keep + __shfl_xor(send, 8, 64);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool up = (lane & 4) != 0;
    const double keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
    v[i] = keep + __shfl_xor(send, 4, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool up = (lane & 2) != 0;
    const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
    v[i] = keep + __shfl_xor(send, 2, 64);
  }
  {
    const bool up = (lane & 1) != 0;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + __shfl_xor(send, 1, 64);
  }
  double t = v[0];
  t += __shfl_xor(t, 16, 64);
  t += __shfl_xor(t, 32, 64);
  return t;
}
__global__ __launch_bounds__(CHAIN_THREADS) void trsv_fwd_chain_direct(const double* L, long ld, const double* dinv,
                                                                       const double* y, double* z, int* fail) {
  __shared__ double zs[2][T];
  __shared__ double ws[T];
  __shared__ int gave_up;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = (long)b * T + wave * 16;
  double acc[16];
  double2 dv[16], blk[16];
  if (tid == 0) gave_up = 0;
  const double yv = y[row0 + (lane & 15)];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    acc[q] = 0.0;
    dv[q] = *reinterpret_cast<const double2*>(dinv + (long)b * T * T + (wave * 16 + q) * T + 2 * lane);
  }
  if (b > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) blk[q] = *reinterpret_cast<const double2*>(L + (row0 + q) * ld + 2 * lane);
  }
  __syncthreads();
  for (int j = 0; j < b; ++j) {
    if (wave == 0) {
      // The further from the front of the chain, the more patiently: a waiting workgroup first probes ONE value with one
      // lane (one request; every waiting workgroup hammering all 128 was the L2 queue as the critical path), and only the
      // next two in line poll the whole block at once.
      const int dist = b - j;
      GhSpin spin(fail);                                  // (gh_spin.h: the 2-s give-up and the abort word)
      bool ok = true;
      if (dist > 2) {
        while (!chain_ready(__hip_atomic_load(z + (long)j * T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          for (int q = dist > 64 ? 16 : dist >> 2; q > 0; --q) __builtin_amdgcn_s_sleep(8);   // 0 .. 8k cycles
          if (!spin.keep_waiting(63u)) { ok = false; break; }
        }
      }
      double2 zj = ld_coherent2(z + (long)j * T + 2 * lane);
      while (ok && !__all(chain_ready(zj.x) && chain_ready(zj.y))) {
        if (!spin.keep_waiting(1023u)) { ok = false; break; }
        zj = ld_coherent2(z + (long)j * T + 2 * lane);
      }
      if (!ok && lane == 0) gave_up = 1;
      *reinterpret_cast<double2*>(&zs[j & 1][2 * lane]) = zj;
    }
    __syncthreads();
    if (gave_up) break;
    const double2 zj = *reinterpret_cast<const double2*>(&zs[j & 1][2 * lane]);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += blk[q].x * zj.x + blk[q].y * zj.y;
    if (j + 1 < b) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        blk[q] = *reinterpret_cast<const double2*>(L + (row0 + q) * ld + (long)(j + 1) * T + 2 * lane);
    }
  }
  // (after a time-out the values published are garbage but NOT the sentinel: the workgroups behind come through, the
  //  host sees *fail)
  const double tot = transpose_sum16(acc, lane);
  if (lane < 16) ws[wave * 16 + lane] = yv - tot;
  __syncthreads();
  const double wx = ws[2 * lane], wy = ws[2 * lane + 1];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = dv[q].x * wx + dv[q].y * wy;
  double v = transpose_sum16(acc, lane);
  if (!chain_ready(v)) v = __longlong_as_double(0x7FF8000000000000LL);      // (a NaN with every bit set must not look unpublished)
  if (lane < 16) __hip_atomic_store(z + row0 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Backward step j of L^T x = z.  x_j = L_jj^-T w_j; columns c < j0: w[c] -= sum_r L[j0+r][c] x_j[r].
// Workgroup j0/128 (the last one) publishes x_j; workgroup b < j0/128 updates columns [128b, 128b+128).
__global__ __launch_bounds__(256) void trsv_bwd_step(const double* L, long ld, const double* dinv_j,
                                                     long j0, double* w, double* x) {
  __shared__ double xj[T];
  __shared__ double part[2][T];
  const int tid = threadIdx.x, c = tid & 127, h = tid >> 7;
  double acc = 0.0;
#pragma unroll 16
  for (int r = h * 64; r < h * 64 + 64; ++r) acc += dinv_j[r * T + c] * w[j0 + r];
  part[h][c] = acc;
  __syncthreads();
  if (tid < T) xj[tid] = part[0][tid] + part[1][tid];
  __syncthreads();
  const long nb = j0 / T;
  if ((long)blockIdx.x == nb) {
    if (tid < T) x[j0 + tid] = xj[tid];
    return;
  }
  const long col0 = (long)blockIdx.x * T;
  acc = 0.0;
#pragma unroll 16
  for (int r = h * 64; r < h * 64 + 64; ++r) acc += L[(j0 + r) * ld + col0 + c] * xj[r];
  __syncthreads();
  part[h][c] = acc;
  __syncthreads();
  if (tid < T) w[col0 + tid] -= part[0][tid] + part[1][tid];
}

// The backward sweep L^T x = z as one chained launch, the mirror image of trsv_fwd_chain_direct: the
// chain runs from the LAST block to the first, so workgroup w owns block column b = nt-1-w (its
// predecessors in the chain then have smaller workgroup indices and are dispatched first).
//   x_b = L_bb^-T ( z_b - sum_{j>b} L_jb^T x_j )
// Wavefront v takes rows 16v..16v+15 of every block L_jb (row segments of 1 KiB, a lane two
// columns), accumulates its lane's two columns of L_jb^T x_j over all j, and the eight wavefront
// partials meet in LDS once, before the diagonal solve (same scheme again with L_bb^-1).
// x itself is the message, as in trsv_fwd_chain_direct: x pre-filled with the sentinel, the first wavefront of a
// consumer polls the 128 values of x_j and hands them on through LDS (they used to be sixteen broadcast loads from
// L2 per wavefront, after the flag had been seen).
__global__ __launch_bounds__(CHAIN_THREADS) void trsv_bwd_chain_direct(const double* L, long ld, const double* dinv, int nt,
                                                                       const double* zin, double* x, int* fail) {
  __shared__ double red[8][T];
  __shared__ double wv[T];
  __shared__ double xs[2][T];
  __shared__ int gave_up;
  const int w = blockIdx.x, b = nt - 1 - w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long col0 = (long)b * T + 2 * lane;
  double2 acc = make_double2(0.0, 0.0);
  double2 dv[16], blk[16];
  if (tid == 0) gave_up = 0;
  const double zv = tid < T ? zin[(long)b * T + tid] : 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q)
    dv[q] = *reinterpret_cast<const double2*>(dinv + (long)b * T * T + (wave * 16 + q) * T + 2 * lane);
  if (b + 1 < nt) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      blk[q] = *reinterpret_cast<const double2*>(L + ((long)(nt - 1) * T + wave * 16 + q) * ld + col0);
  }
  __syncthreads();
  for (int j = nt - 1; j > b; --j) {
    if (wave == 0) {
      const int dist = j - b;
      GhSpin spin(fail);
      bool ok = true;
      if (dist > 2) {
        while (!chain_ready(__hip_atomic_load(x + (long)j * T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          for (int q = dist > 64 ? 16 : dist >> 2; q > 0; --q) __builtin_amdgcn_s_sleep(8);
          if (!spin.keep_waiting(63u)) { ok = false; break; }
        }
      }
      double2 xj = ld_coherent2(x + (long)j * T + 2 * lane);
      while (ok && !__all(chain_ready(xj.x) && chain_ready(xj.y))) {
        if (!spin.keep_waiting(1023u)) { ok = false; break; }
        xj = ld_coherent2(x + (long)j * T + 2 * lane);
      }
      if (!ok && lane == 0) gave_up = 1;
      *reinterpret_cast<double2*>(&xs[j & 1][2 * lane]) = xj;
    }
    __syncthreads();
    if (gave_up) break;
    const double* xw = &xs[j & 1][wave * 16];              // x_j[16 wave + q]: LDS broadcast reads
#pragma unroll
    for (int q = 0; q < 16; ++q) { const double xr = xw[q]; acc.x += blk[q].x * xr; acc.y += blk[q].y * xr; }
    if (j - 1 > b) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        blk[q] = *reinterpret_cast<const double2*>(L + ((long)(j - 1) * T + wave * 16 + q) * ld + col0);
    }
  }
  red[wave][2 * lane] = acc.x;
  red[wave][2 * lane + 1] = acc.y;
  __syncthreads();
  if (tid < T) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += red[q][tid];
    wv[tid] = zv - v;
  }
  __syncthreads();
  acc = make_double2(0.0, 0.0);                         // x_b[c] = sum_r dinv_b[r][c] w[r]
#pragma unroll
  for (int q = 0; q < 16; ++q) { const double wr = wv[wave * 16 + q]; acc.x += dv[q].x * wr; acc.y += dv[q].y * wr; }
  __syncthreads();
  red[wave][2 * lane] = acc.x;
  red[wave][2 * lane + 1] = acc.y;
  __syncthreads();
  if (tid < T) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += red[q][tid];
    if (!chain_ready(v)) v = __longlong_as_double(0x7FF8000000000000LL);
    __hip_atomic_store(x + (long)b * T + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ========================================================= predict reductions
// partial[s][c] = sum over the s-th row chunk of V[r][c] * z[r]  and of V[r][c]^2
__global__ __launch_bounds__(256) void colreduce_kernel(const double* V, long ldv, long nrows, long rows_per,
                                                        const double* z, double* pmu, double* pvar, long ncols_p) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols_p) return;
  const long r0 = (long)blockIdx.y * rows_per;
  const long r1 = r0 + rows_per < nrows ? r0 + rows_per : nrows;
  double am = 0.0, av = 0.0;
  for (long r = r0; r < r1; ++r) {
    const double v = V[r * ldv + c];
    am += v * z[r];
    av += v * v;
  }
  pmu[(long)blockIdx.y * ncols_p + c] = am;
  pvar[(long)blockIdx.y * ncols_p + c] = av;
}
__global__ void colfinal_kernel(const double* pmu, const double* pvar, long nchunks, long ncols_p, long m,
                                double* mu, double* var /* in: k(xs,xs) diag; may be NULL */) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  double am = 0.0, av = 0.0;
  for (long s = 0; s < nchunks; ++s) { am += pmu[s * ncols_p + c]; av += pvar[s * ncols_p + c]; }
  mu[c] = am;
  if (var) var[c] -= av;
}
__global__ void fill_kernel(double* p, long n, double v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void eye_kernel(double* p, long n, long ld) {   // p must be pre-zeroed
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i * ld + i] = 1.0;
}
// mirror the lower triangle of an n x n matrix into the upper one (out may be != in)
__global__ void symmetrize_kernel(const double* in, long ldi, double* out, long ldo, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const long i = idx / n, j = idx % n;
  out[i * ldo + j] = (j <= i) ? in[i * ldi + j] : in[j * ldi + i];
}
__global__ void copy2d_kernel(const double* in, long ldi, double* out, long ldo, long rows, long cols) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const long i = idx / cols, j = idx % cols;
  out[i * ldo + j] = in[i * ldi + j];
}

// ================================================================= the solver
struct EvPair { hipEvent_t a, b; };

struct gh_chol {
  gh_chol_opts opts;
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;             // high-priority panel stream (look-ahead)
  hipStream_t st3 = nullptr;             // second panel stream: rows-below TRSM beside the potf2 chain
  bool shared_streams = false;           // st, st2, st3, st4, st_mask belong to the process (gh_shared_streams): not destroyed here
  hipStream_t st4 = nullptr;             // third panel stream: in-panel rows >= j+2 (everything off the potf2 chain)
  hipEvent_t ev_diag[32] = {};             // one per 128-column step of a panel (panels of up to 4096 columns)
  hipEvent_t ev_aux = nullptr, ev_aux2 = nullptr;
  std::vector<hipEvent_t> ev_p, ev_w, ev_nf;   // deep look-ahead: panel j factored / W(j) done / U(j, j+2) done
  hipStream_t st_mask = nullptr;         // main-stream stand-in that leaves CUs to the panel chain (small N)
  hipStream_t tail = nullptr;            // where the last factor() ended: the stream on which its results are complete in stream order


  int mask_reserved = -1;                // CUs st_mask leaves out (-1: not created yet, 0: creation failed)
  hipEvent_t ev_xfer = nullptr;
  hipEvent_t ev_sync[3] = {nullptr, nullptr, nullptr};
  int64_t n = 0, np = 0;
  int ndim = 0;
  bool computed = false;
  int64_t info = 0;
  double logdet = 0.0;
  GhBuf A, dinv, x, yerr, v0, v1, v2, scal, rhs, work, work2, scratch, chain;
  long long* d_info = nullptr;           // = (long long*)(scal + 2): the failure word lives beside the scalars (set in compute_enqueue)
  bool build_on_chain = false;           // this compute(): inputs + kernel-matrix build were enqueued on the chain stream (st2)
  gh_chol_profile prof;
  std::vector<EvPair> ev_pool;
  size_t ev_used = 0;
  std::vector<size_t> ev_trailing, ev_panel, ev_update;   // ev_update: EVERY trailing-update launch (wide SYRKs and block-column GEMMs)
  std::vector<double> ev_update_flops;                    // algorithmic flops of each ev_update launch
  std::vector<double> upd_intervals;                      // last compute(): (start ms, end ms, flops) per trailing-update launch
  // returns an index into ev_pool (the vector may grow, so never keep pointers), or -1
  long next_ev() {
    if (ev_used == ev_pool.size()) {
      EvPair p;
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
      ev_pool.push_back(p);
    }
    return (long)ev_used++;
  }
  ~gh_chol() {
    for (auto& p : ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& e : ev_sync) if (e) (void)hipEventDestroy(e);
    if (ev_xfer) (void)hipEventDestroy(ev_xfer);
    if (ev_aux) (void)hipEventDestroy(ev_aux);
    if (ev_aux2) (void)hipEventDestroy(ev_aux2);
    for (auto* v : {&ev_p, &ev_w, &ev_nf}) for (auto e : *v) (void)hipEventDestroy(e);
    for (auto& e : ev_diag) if (e) (void)hipEventDestroy(e);
    if (st4 && !shared_streams) (void)hipStreamDestroy(st4);
    if (st3 && !shared_streams) (void)hipStreamDestroy(st3);
    if (st_mask && !shared_streams) (void)hipStreamDestroy(st_mask);


    if (st2 && !shared_streams) (void)hipStreamDestroy(st2);
    if (st && !shared_streams) (void)hipStreamDestroy(st);
  }
};

// ---- the process-wide streams (see gh_common.h)
#include <map>
#include <mutex>
namespace {
struct SharedStreams {
  hipStream_t q[4] = {nullptr, nullptr, nullptr, nullptr};
  bool made = false;
  bool main_crowded = false;       // the main stream shares a dispatcher with the chain, rows-below or near stream (measured when they are made)
  std::map<int, hipStream_t> masked;
};
std::mutex g_ss_mu;
std::map<int, SharedStreams> g_ss;
}
// (Where the runtime puts these streams matters more than which streams there are: DESIGN.md, "One set of streams per
//  process" -- gh_prime_device() takes care of the common case, gh_debug_stream_dispatch() shows the placement.)
namespace {
__global__ void place_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
// ms until a one-workgroup kernel on `small` has completed when it is launched right after a ~1 ms grid (2^17 workgroups,
// far more than the chip holds) on `big`: ~0.05 when the two queues dispatch independently, the grid's duration when not
double dispatch_wait_ms(hipStream_t big, hipStream_t small) {
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(place_spin_kernel, dim3(1 << 17), dim3(64), 0, big, 5000LL);
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(place_spin_kernel, dim3(1), dim3(64), 0, small, 0LL);
  (void)hipStreamSynchronize(small);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  (void)hipDeviceSynchronize();
  return ms;
}
}  // namespace
// do kernels of `a` and `b` (streams of the current device) dispatch side by side?  (gh_mgpu.hip: two communicators at once)
bool gh_streams_dispatch_independently(hipStream_t a, hipStream_t b) {
  if (!a || !b || a == b) return false;
  hipLaunchKernelGGL(place_spin_kernel, dim3(1), dim3(64), 0, a, 0LL);          // (queues are made at first use)
  hipLaunchKernelGGL(place_spin_kernel, dim3(1), dim3(64), 0, b, 0LL);
  (void)hipDeviceSynchronize();
  const bool ok = dispatch_wait_ms(a, b) <= 0.3 && dispatch_wait_ms(b, a) <= 0.3;
  (void)hipGetLastError();
  return ok;
}
bool gh_shared_main_crowded(int device) {
  std::lock_guard<std::mutex> lk(g_ss_mu);
  auto it = g_ss.find(device);
  return it != g_ss.end() && it->second.main_crowded;
}
bool gh_shared_streams(int device, hipStream_t q[4]) {
  std::lock_guard<std::mutex> lk(g_ss_mu);
  SharedStreams& ss = g_ss[device];
  if (!ss.made) {
    ss.made = true;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    gh_prime_device(device);
    if (hipStreamCreate(&ss.q[0]) != hipSuccess) { ss.q[0] = nullptr; (void)hipGetLastError(); }
    int lo = 0, hi = 0;                    // numerically lowest value = highest priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // (the rows-below and near streams at normal or low priority instead of the chain's: no difference, 6.98 / 7.07 / 7.02 ms at N = 8192)
    for (int i = 1; i < 4 && ss.q[0]; ++i)
      if (hipStreamCreateWithPriority(&ss.q[i], hipStreamNonBlocking, hi) != hipSuccess) { ss.q[i] = nullptr; (void)hipGetLastError(); break; }
    // Does the main stream share a dispatcher with one of the panel streams?  (Which queues end up together depends on how
    // many the process made before: never in a process that made none, with the rows-below stream after five application
    // streams, with the chain stream after six.)  Decides where a look-ahead factorisation is joined: factor_lookahead_deep.
    if (ss.q[0] && ss.q[1] && ss.q[2] && ss.q[3]) {
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(place_spin_kernel, dim3(1), dim3(64), 0, ss.q[i], 0LL);   // (queues are made at first use)
      (void)hipDeviceSynchronize();
      for (int i = 1; i < 4 && !ss.main_crowded; ++i)
        ss.main_crowded = dispatch_wait_ms(ss.q[0], ss.q[i]) > 0.3 || dispatch_wait_ms(ss.q[i], ss.q[0]) > 0.3;
      (void)hipGetLastError();
    }
  }
  for (int i = 0; i < 4; ++i) q[i] = ss.q[i];
  return ss.q[0] != nullptr;
}
hipStream_t gh_shared_masked_stream(int device, int reserve_cus) {
  std::lock_guard<std::mutex> lk(g_ss_mu);
  SharedStreams& ss = g_ss[device];
  auto it = ss.masked.find(reserve_cus);
  if (it != ss.masked.end()) return it->second;
  hipStream_t st = nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 2 * reserve_cus) {
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    for (int c = reserve_cus; c < ncu; ++c) mask[c / 32] |= (1u << (c % 32));
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) { st = nullptr; (void)hipGetLastError(); }
  }
  ss.masked[reserve_cus] = st;             // (a failed creation is not retried)
  return st;
}

static int set_device(gh_chol* s) {
  if (gh_device_count() <= 0) { gh_set_error("no HIP device available: the george_amd solver needs an MI355X"); return GH_ERR_HIP; }
  GH_HIP(hipSetDevice(s->opts.device));
  return GH_OK;
}

extern "C" int gh_chol_create(const gh_chol_opts* opts, gh_chol** out) {
  if (!out) { gh_set_error("null output"); return GH_ERR_BAD_ARG; }
  gh_chol* s = new gh_chol();
  memset(&s->opts, 0, sizeof(s->opts));
  memset(&s->prof, 0, sizeof(s->prof));
  if (opts) s->opts = *opts;
  if (s->opts.nb < 0) s->opts.nb = 0;       // 0 = choose per problem size (panel_width())
  if (s->opts.nb % T) { delete s; gh_set_error("nb must be a multiple of 128"); return GH_ERR_BAD_ARG; }
  int rc = set_device(s);
  if (rc != GH_OK) { delete s; return rc; }
  hipStream_t shq[4] = {nullptr, nullptr, nullptr, nullptr};
  if (gh_shared_streams(s->opts.device, shq)) {
    s->shared_streams = true;
    s->st = shq[0];
  } else {
    gh_prime_device(s->opts.device);
    if (hipStreamCreate(&s->st) != hipSuccess) { delete s; gh_set_error("hipStreamCreate failed"); return GH_ERR_HIP; }
  }
  if (s->opts.lookahead) {
    int lo = 0, hi = 0;                    // numerically lowest value = highest priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (s->shared_streams) s->st2 = shq[1];
    else if (hipStreamCreateWithPriority(&s->st2, hipStreamNonBlocking, hi) != hipSuccess) { s->st2 = nullptr; (void)hipGetLastError(); }
    for (auto& e : s->ev_sync)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete s; gh_set_error("hipEventCreate failed"); return GH_ERR_HIP; }
    if (s->st2) {
      if (s->shared_streams) s->st3 = shq[2];
      bool ok = (s->shared_streams ? s->st3 != nullptr : hipStreamCreateWithPriority(&s->st3, hipStreamNonBlocking, hi) == hipSuccess) &&
                hipEventCreateWithFlags(&s->ev_aux, hipEventDisableTiming) == hipSuccess;
      for (auto& e : s->ev_diag) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); if (s->st3 && !s->shared_streams) (void)hipStreamDestroy(s->st3); s->st3 = nullptr; }
      if (s->st3) {
        if (s->shared_streams) s->st4 = shq[3];
        bool ok4 = (s->shared_streams ? s->st4 != nullptr : hipStreamCreateWithPriority(&s->st4, hipStreamNonBlocking, hi) == hipSuccess) &&
                   hipEventCreateWithFlags(&s->ev_aux2, hipEventDisableTiming) == hipSuccess;
        if (!ok4) { (void)hipGetLastError(); if (s->st4 && !s->shared_streams) (void)hipStreamDestroy(s->st4); s->st4 = nullptr; }
      }
    }
  }
  *out = s;
  return GH_OK;
}
// Which of the handle's streams really run side by side?  HIP maps streams onto a few hardware queues
// and two streams on one queue serialise.  out[i * 6 + j] (i < j) = milliseconds for a 300-us spin
// kernel on stream i and one on stream j launched together (0.3 = concurrent, 0.6 = one queue);
// streams: 0 the caller's null stream, 1 main, 2 chain, 3 rows-below, 4 near, 5 CU-masked trailing.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
extern "C" int gh_debug_stream_overlap(gh_chol* s, double* out, int n) {
  if (!s || !out || n < 36) { gh_set_error("bad argument"); return GH_ERR_BAD_ARG; }
  int rc = set_device(s);
  if (rc != GH_OK) return rc;
  hipStream_t q[6] = {nullptr, s->st, s->st2, s->st3, s->st4, s->st_mask};
  for (int i = 0; i < 36; ++i) out[i] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = i + 1; j < 6; ++j) {
      if ((i > 0 && !q[i]) || !q[j]) { out[i * 6 + j] = -1.0; continue; }
      GH_HIP(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, q[i], 30000LL);
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, q[j], 30000LL);
      GH_HIP(hipDeviceSynchronize());
      out[i * 6 + j] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  return GH_OK;
}
// The other way two streams can be in each other's way: a grid with far more workgroups than the chip holds keeps
// its queue's dispatcher busy for its whole duration.  out[i * 6 + j] (i != j) = milliseconds until a ONE-workgroup
// kernel launched on stream j right after such a grid on stream i (2^18 workgroups of 64 threads spinning ~50 us:
// ~2 ms) has completed: tens of microseconds when the two queues dispatch independently, the grid's duration when the
// small kernel has to wait for the big one's dispatch to end.  Streams as in gh_debug_stream_overlap.
extern "C" int gh_debug_stream_dispatch(gh_chol* s, double* out, int n) {
  if (!s || !out || n < 36) { gh_set_error("bad argument"); return GH_ERR_BAD_ARG; }
  int rc = set_device(s);
  if (rc != GH_OK) return rc;
  hipStream_t q[6] = {nullptr, s->st, s->st2, s->st3, s->st4, s->st_mask};
  for (int i = 0; i < 36; ++i) out[i] = 0.0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      if (i == j) continue;
      if ((i > 0 && !q[i]) || (j > 0 && !q[j])) { out[i * 6 + j] = -1.0; continue; }
      GH_HIP(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin_kernel, dim3(1 << 18), dim3(64), 0, q[i], 5000LL);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, q[j], 0LL);
      GH_HIP(hipStreamSynchronize(q[j]));
      out[i * 6 + j] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      GH_HIP(hipDeviceSynchronize());
    }
  return GH_OK;
}
extern "C" void gh_chol_destroy(gh_chol* s) {
  if (!s) return;
  (void)hipSetDevice(s->opts.device);
  // entry points that were handed DEVICE output pointers return with copies still queued: drain the
  // handle's streams before its buffers go back to the allocator / the block cache
  for (hipStream_t st : {s->st, s->st2, s->st3, s->st4, s->st_mask}) if (st) (void)hipStreamSynchronize(st);
  delete s;
}
extern "C" int64_t gh_chol_info(const gh_chol* s) { return s ? s->info : 0; }
extern "C" int64_t gh_chol_size(const gh_chol* s) { return s ? s->n : 0; }
extern "C" int64_t gh_chol_device_bytes(const gh_chol* s) {
  if (!s) return 0;
  size_t tot = 0;
  for (const GhBuf* b : {&s->A, &s->dinv, &s->x, &s->yerr, &s->v0, &s->v1, &s->v2, &s->scal, &s->rhs, &s->work, &s->work2,
                         &s->scratch, &s->chain}) tot += b->p ? b->bytes : 0;
  return (int64_t)tot;
}
extern "C" int gh_chol_get_update_intervals(const gh_chol* s, double* out, int32_t max_launches, int32_t* n_out) {
  if (!s || !n_out || (max_launches > 0 && !out)) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  const int32_t have = (int32_t)(s->upd_intervals.size() / 3);
  *n_out = have;
  for (int32_t i = 0; i < have && i < max_launches; ++i)
    for (int q = 0; q < 3; ++q) out[3 * i + q] = s->upd_intervals[3 * (size_t)i + q];
  return GH_OK;
}
extern "C" int gh_chol_get_profile(const gh_chol* s, gh_chol_profile* out) {
  if (!s || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  *out = s->prof;
  return GH_OK;
}

static inline double* blk(double* A, int64_t ld, int64_t r, int64_t c) { return A + r * ld + c; }

// C = A * B^T (alpha=1,beta=0) or C -= A * B^T helpers on k-major operands
// (set by factor(): the K = 128 GEMMs of the chain hold 128-144 KiB of LDS per workgroup -- a whole CU.  Below
//  Np = 24576 the trailing SYRK leaves 32 CUs out and they run there; above it every CU carries two SYRK
//  workgroups and a chain workgroup that needs a CU to itself waits for one to drain while the dispatcher
//  holds it empty: N = 65536 went from 1.407 to 1.433 s.  There they keep the 32-KiB K-loop kernel.)
static thread_local bool t_gemm_small_lds = false;
static int gemm_nt(hipStream_t st, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                   int64_t M, int64_t N, int64_t K, double alpha, double beta, bool lower) {
  GhGemm g{};
  g.small_lds = t_gemm_small_lds;
  g.C = C; g.ldc = ldc; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta; g.a_km = true; g.b_km = true; g.lower = lower;
  return gh_launch_gemm(g, st);
}

// gh_potf2.hip: MFMA-blocked 128x128 Cholesky + inverse (the default); GEORGE_AMD_POTF2=simple
// selects the first scalar version above for A/B validation
int gh_launch_potf2_mfma(double* A, int64_t lda, double* dinv, long long* info, long long base, hipStream_t st);
static bool use_simple_potf2() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("GEORGE_AMD_POTF2"); v = (e && e[0] == 's') ? 1 : 0; }
  return v == 1;
}

// in-place lower Cholesky of the n x n block at A (n multiple of 128) + diagonal-block inverses
static int potrf_block(hipStream_t st, double* A, int64_t ld, int64_t n, double* dinv, long long* d_info, long long base) {
  for (int64_t j0 = 0; j0 < n; j0 += T) {
    double* dj = dinv + (j0 / T) * T * T;
    if (use_simple_potf2()) {
      hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, st, blk(A, ld, j0, j0), (long)ld, dj, d_info, base + j0);
      GH_HIP(hipGetLastError());
    } else {
      GH_CHECK(gh_launch_potf2_mfma(blk(A, ld, j0, j0), ld, dj, d_info, base + j0, st));
    }
    const int64_t rem = n - (j0 + T);
    if (rem > 0) {
      double* P = blk(A, ld, j0 + T, j0);
      GH_CHECK(gemm_nt(st, P, ld, P, ld, dj, T, rem, T, T, 1.0, 0.0, false));          // P <- P L_jj^-T (in place)
      GH_CHECK(gemm_nt(st, blk(A, ld, j0 + T, j0 + T), ld, P, ld, P, ld, rem, rem, T, -1.0, 1.0, true));
    }
  }
  return GH_OK;
}
// A21 (m x n) <- A21 L11^-T, L11 n x n lower with diagonal-block inverses dinv
static int trsm_right(hipStream_t st, const double* L11, int64_t ld11, const double* dinv, double* A21, int64_t lda, int64_t m, int64_t n) {
  for (int64_t j0 = 0; j0 < n; j0 += T) {
    double* Xj = A21 + j0;
    if (j0 > 0)
      GH_CHECK(gemm_nt(st, Xj, lda, A21, lda, L11 + j0 * ld11, ld11, m, T, j0, -1.0, 1.0, false));
    GH_CHECK(gemm_nt(st, Xj, lda, Xj, lda, dinv + (j0 / T) * T * T, T, m, T, T, 1.0, 0.0, false));
  }
  return GH_OK;
}

// z = L^-1 w as one chained launch; flags[nt] = the time-out flag (cleared here; the words before it are no longer
// used: the flag-per-block-row kernels are retired, scripts/dev/arms/trsv_chain_flags.hip.inc).  w is read only and
// must not be z (z is pre-filled with the sentinel).
static int launch_trsv_fwd_chain(const double* L, long ld, const double* dinv, int64_t nt, const double* w, double* z,
                                 unsigned* flags, hipStream_t st) {
  GH_HIP(hipMemsetAsync(flags + nt, 0, sizeof(unsigned), st));
  GH_HIP(hipMemsetAsync(z, 0xFF, (size_t)nt * T * sizeof(double), st));
  hipLaunchKernelGGL(trsv_fwd_chain_direct, dim3((unsigned)nt), dim3(CHAIN_THREADS), 0, st, L, ld, dinv, w, z, (int*)(flags + nt));
  GH_HIP(hipGetLastError());
  return GH_OK;
}
static int launch_trsv_bwd_chain(const double* L, long ld, const double* dinv, int64_t nt, const double* w, double* x,
                                 unsigned* flags, hipStream_t st) {
  GH_HIP(hipMemsetAsync(flags + nt, 0, sizeof(unsigned), st));
  GH_HIP(hipMemsetAsync(x, 0xFF, (size_t)nt * T * sizeof(double), st));
  hipLaunchKernelGGL(trsv_bwd_chain_direct, dim3((unsigned)nt), dim3(CHAIN_THREADS), 0, st, L, ld, dinv, (int)nt, w, x, (int*)(flags + nt));
  GH_HIP(hipGetLastError());
  return GH_OK;
}
extern "C" int gh_dev_potrf_block(double* a, int64_t lda, int64_t n, double* dinv, int64_t* info_dev, int64_t base_index, void* stream) {
  if (n % T) { gh_set_error("potrf_block: n must be a multiple of 128"); return GH_ERR_BAD_ARG; }
  // (tile operations of the multi-GPU driver run beside trailing updates that own every CU: small-LDS GEMMs, see t_gemm_small_lds)
  struct Guard { bool prev; Guard() : prev(t_gemm_small_lds) { t_gemm_small_lds = true; } ~Guard() { t_gemm_small_lds = prev; } } guard;
  return potrf_block((hipStream_t)stream, a, lda, n, dinv, (long long*)info_dev, base_index);
}
extern "C" int gh_dev_trsm_right(const double* l11, int64_t ld11, const double* dinv, double* a21, int64_t lda, int64_t m, int64_t n, void* stream) {
  if (n % T || m % T) { gh_set_error("trsm_right: sizes must be multiples of 128"); return GH_ERR_BAD_ARG; }
  struct Guard { bool prev; Guard() : prev(t_gemm_small_lds) { t_gemm_small_lds = true; } ~Guard() { t_gemm_small_lds = prev; } } guard;
  return trsm_right((hipStream_t)stream, l11, ld11, dinv, a21, lda, m, n);
}
extern "C" int gh_dev_trsv_lower(const double* l, int64_t ldl, const double* dinv, int64_t n,
                                 const double* w, double* z, void* scratch, void* stream) {
  if (n % T || n <= 0 || !scratch) { gh_set_error("trsv_lower: n must be a positive multiple of 128"); return GH_ERR_BAD_ARG; }
  const int64_t nt = n / T;
  if (w == z) { gh_set_error("trsv_lower: w and z must not be the same array"); return GH_ERR_BAD_ARG; }
  return launch_trsv_fwd_chain(l, (long)ldl, dinv, nt, w, z, (unsigned*)scratch, (hipStream_t)stream);
}
extern "C" int gh_dev_trsv_lower_t(const double* l, int64_t ldl, const double* dinv, int64_t n,
                                   const double* w, double* x, void* scratch, void* stream) {
  if (n % T || n <= 0 || !scratch) { gh_set_error("trsv_lower_t: n must be a positive multiple of 128"); return GH_ERR_BAD_ARG; }
  const int64_t nt = n / T;
  if (w == x) { gh_set_error("trsv_lower_t: w and x must not be the same array"); return GH_ERR_BAD_ARG; }
  return launch_trsv_bwd_chain(l, (long)ldl, dinv, nt, w, x, (unsigned*)scratch, (hipStream_t)stream);
}
extern "C" int gh_dev_logdet_accum(const double* a, int64_t lda, int64_t n, double* out_dev, void* stream) {
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, (long)lda, (long)n, out_dev, 1);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// Outer panel width.  Wider panels raise the SYRK's arithmetic intensity and K-loop length and
// halve the number of look-ahead hand-overs, but put more work into the latency-bound panel chain.
// Measured (bench.py --nb, after the chain work of this round): 1024 wins from N = 16384 up
// (N = 16384: 36.5 vs 37.8 ms, N = 20480: 60.7 vs 63.3 ms, N = 65536: 1437 vs 1480 ms), 512 at
// N = 8192 (10.9 vs 11.2 ms), 1024 again at N <= 4096 (4.95 vs 5.03 ms at 4096, 1.16 vs 1.26 ms
// at 1024: the fewer hand-overs between the streams the better), 128 and 256 nowhere.
static int64_t panel_width(const gh_chol* s) {
  if (s->opts.nb > 0) return s->opts.nb;
  // (swept again after the second 128x128 kernel and the K = 128 GEMM, N = 2048 .. 20480, nb = 256 .. 2048:
  //  1024 wins or ties everywhere -- 512 used to win between 4096 and 12288, when a chain link cost twice as much)
  return 1024;
}
// Start columns of the outer panels, pc[0] = 0 < pc[1] < ... < pc[P] = Np.  With the width left to the solver (opts.nb == 0) the
// panels are 2048 columns wide while the trailing matrix behind them is larger than GH_WIDE_PANEL_MIN_TRAILING columns and 1024
// after (round 6; round 5 measured a uniform 2048 at -1.4 % for N = 65536 and +3.6 % for N = 32768: the K = 2048 update runs its
// tiles 2 % faster and halves the launches, but a 2048-column panel is sixteen chain links + a block-column update twice as
// deep, which only a trailing update of more than ~20 ms hides -- at 66 TFLOP/s that is a trailing matrix of ~25 000 columns).
#ifndef GH_WIDE_PANEL_MIN_TRAILING
#define GH_WIDE_PANEL_MIN_TRAILING 25600
#endif

static int g_build_on_chain = 1;
extern "C" int gh_debug_set_build_on_chain(int on) {
  const int prev = g_build_on_chain;
  g_build_on_chain = on ? 1 : 0;
  return prev;
}
static int g_adaptive_panels = 1;        // 0: off; 1: on (GH_WIDE_PANEL_MIN_TRAILING); > 1: on with this many trailing columns as the bound
extern "C" int gh_debug_set_adaptive_panels(int on) {
  const int prev = g_adaptive_panels;
  g_adaptive_panels = on < 0 ? 1 : on;
  return prev;
}
static std::vector<int64_t> panel_starts(const gh_chol* s) {
  std::vector<int64_t> pc;
  const int64_t np = s->np, nb = panel_width(s);
  const bool adaptive = s->opts.nb == 0 && g_adaptive_panels && !use_simple_potf2();
  for (int64_t k0 = 0; k0 < np;) {
    pc.push_back(k0);
    const int64_t bound = g_adaptive_panels > 1 ? g_adaptive_panels : GH_WIDE_PANEL_MIN_TRAILING;
    int64_t w = (adaptive && np - (k0 + 2 * nb) >= bound) ? 2 * nb : nb;
#ifdef GH_WIDE_PANEL_TIER2
    if (adaptive && np - (k0 + 4 * nb) >= GH_WIDE_PANEL_TIER2) w = 4 * nb;
#endif
    k0 += std::min<int64_t>(w, np - k0);
  }
  pc.push_back(np);
  return pc;
}

// One panel step: factor the nb x nb diagonal block at k0, TRSM the rows below it.
static int panel_step(gh_chol* s, hipStream_t st, int64_t k0, int64_t nb) {
  double* A = s->A.d();
  const int64_t np = s->np, ld = np;
  double* dinv = s->dinv.d() + (k0 / T) * T * T;
  const int64_t m = np - (k0 + nb);
  const bool on_panel_stream = (st == s->st2);
  // (Retired arms, all measured and slower, sources under scripts/dev/arms/: only row block j+1 on the chain and the
  //  other in-panel rows on a third stream; the chain on CUs of its own; only the potf2 launches on reserved CUs; the
  //  whole panel as two persistent flag-driven launches.  DESIGN.md section 4, "Where N < 24k stands".)
  if (!s->st3 || !on_panel_stream || m <= 0 || nb / T > 32 || use_simple_potf2()) {
    GH_CHECK(potrf_block(st, blk(A, ld, k0, k0), ld, nb, dinv, s->d_info, k0));
    if (m > 0) {
      GH_CHECK(trsm_right(st, blk(A, ld, k0, k0), ld, dinv, blk(A, ld, k0 + nb, k0), ld, m, nb));
    }
    return GH_OK;
  }
  // Look-ahead panels: the potf2 chain of the diagonal block stays on `st`; the TRSM of the rows
  // below runs on a second panel stream, column block j as soon as L_jj^-1 exists, so that only
  // the last block's TRSM is left when the chain ends (instead of all nb/128 of them).
  hipStream_t sa = s->st3;
  double* Ak = blk(A, ld, k0, k0);
  double* B = blk(A, ld, k0 + nb, k0);
  // (the rows-below stream's first operation waits for ev_diag[0], recorded on `st` behind everything this panel needs: no event of
  //  its own at the panel's start -- a record costs the recording stream ~6 us before its next kernel)
  for (int64_t j0 = 0; j0 < nb; j0 += T) {
    double* dj = dinv + (j0 / T) * T * T;
    GH_CHECK(gh_launch_potf2_mfma(blk(Ak, ld, j0, j0), ld, dj, s->d_info, k0 + j0, st));
    GH_HIP(hipEventRecord(s->ev_diag[j0 / T], st));
    GH_HIP(hipStreamWaitEvent(sa, s->ev_diag[j0 / T], 0));
    double* Xj = B + j0;
    if (j0 > 0) GH_CHECK(gemm_nt(sa, Xj, ld, B, ld, Ak + j0 * ld, ld, m, T, j0, -1.0, 1.0, false));
    GH_CHECK(gemm_nt(sa, Xj, ld, Xj, ld, dj, T, m, T, T, 1.0, 0.0, false));
    const int64_t rem = nb - (j0 + T);
    if (rem > 0) {
      double* P = blk(Ak, ld, j0 + T, j0);
      GH_CHECK(gemm_nt(st, P, ld, P, ld, dj, T, rem, T, T, 1.0, 0.0, false));
      GH_CHECK(gemm_nt(st, blk(Ak, ld, j0 + T, j0 + T), ld, P, ld, P, ld, rem, rem, T, -1.0, 1.0, true));
    }
  }
  GH_HIP(hipEventRecord(s->ev_aux, sa));
  GH_HIP(hipStreamWaitEvent(st, s->ev_aux, 0));
  return GH_OK;
}

// Right-looking factorisation with one-panel look-ahead on two HIP streams:
//   main stream  : trailing updates (the MFMA-bound >90 % of the work);
//   panel stream : (high priority) potf2/TRSM chain of the NEXT panel, which is latency-bound
//                  and would otherwise sit on the critical path between two trailing updates.
// Step k: the main stream first updates only block column k+1 (the next panel), signals the panel
// stream, then updates the rest of the trailing matrix while the panel stream factors panel k+1.
// The two touch disjoint regions: panel k+1 = columns [k1, k1+nb1), the remainder = rows and
// columns >= k1+nb1; both only READ panel k.
// Small matrices are bound by the panel chain (128 potf2 workgroups in a row at N = 16384), and
// that chain runs 4x slower when its workgroups share a CU with wavefronts of the trailing SYRK
// (in-kernel timers: potf2 114 us alone, 420-480 us beside SYRK -- LDS-queue contention, wavefront
// priority does not help).  For those sizes the trailing updates go to a stream whose CU mask
// leaves 32 CUs free for the panel stream.  The count is not arbitrary: the mask bits are dealt
// round-robin over the 8 XCDs and then over the 4 shader engines of each, and the workgroup
// dispatcher feeds shader engines evenly -- leaving out 8 CUs (one engine of every XCD one CU
// short) costs the SYRK 12 %, the same as leaving out 32 (every engine one short), so 32 it is:
// 12.5 % of the chip, about the panel's share of the flops at N = 16384 (9 %).  Larger matrices
// hide the chain behind the SYRK anyway and keep all 256 CUs.  The masked stream takes ~1 s to
// create (ROCm 7.2): once per process (gh_shared_masked_stream).  GEORGE_AMD_RESERVE_CUS=0 disables, =<n> forces n CUs.
static hipStream_t trailing_stream(gh_chol* s) {
  int want = s->np < 24576 ? 32 : 0;
  if (const char* e = getenv("GEORGE_AMD_RESERVE_CUS")) want = atoi(e);
  if (want <= 0 || want >= 128) return s->st;
  if (s->mask_reserved != want) {
    if (s->st_mask) { (void)hipStreamSynchronize(s->st_mask); if (!s->shared_streams) (void)hipStreamDestroy(s->st_mask); s->st_mask = nullptr; }
    s->mask_reserved = want;                                       // (a failed creation is not retried)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, s->opts.device) == hipSuccess && prop.multiProcessorCount > 2 * want) {
      const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
      std::vector<uint32_t> mask(words, 0u);
      for (int c = want; c < ncu; ++c) mask[c / 32] |= (1u << (c % 32));
      if (s->shared_streams) s->st_mask = gh_shared_masked_stream(s->opts.device, want);
      else if (hipExtStreamCreateWithCUMask(&s->st_mask, words, mask.data()) != hipSuccess) { s->st_mask = nullptr; (void)hipGetLastError(); }
    }
    if (s->st_mask && !s->ev_xfer && hipEventCreateWithFlags(&s->ev_xfer, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError(); if (!s->shared_streams) (void)hipStreamDestroy(s->st_mask); s->st_mask = nullptr;
    }
  }
  return s->st_mask ? s->st_mask : s->st;
}

// Look-ahead of depth d.  After panel j is factored, its trailing update is issued per block column
// for the d columns next to it and as ONE lower-triangular SYRK for the rest:
//   chain stream `sp` : panel(j) -> U(j, j+1) -> panel(j+1) -> ...            (the critical path)
//   near stream  `sn` : U(j, j+2), ..., U(j, j+d)                             (narrow GEMMs, high priority)
//   main stream  `sm` : W(j) = U(j, j+d+1 ...)                                (the wide SYRK)
// U(j, c): A[c0:, c0:c0+nb_c] -= L[c0:, j-panel] L[c0:c0+nb_c, j-panel]^T.  Every column receives its
// updates in panel order: U(j, c) follows U(j-1, c), which is the previous launch of the same
// stream except at the window's edges -- U(j, j+1) follows U(j-1, j+1) from the near stream (event
// ev_nf[j-1]), U(j, j+d) follows W(j-1) (event ev_w[j-1]) -- and everything of panel j follows
// ev_p[j].  With d = 1 the chain can start panel j+1 only when W(j-1) is over, so at sizes where the
// early SYRKs outlast a panel and the late ones do not (N ~ 8k-24k) the total is the SUM of the
// larger of the two per step; with d > 1 the chain runs up to d panels ahead during the SYRK-bound
// early steps and spends that lead in the chain-bound late ones.
static int factor_lookahead_deep(gh_chol* s, int depth) {
  // depth 1 (the default): the only work of the "near" stream is the rows below the diagonal block of
  // U(j, j+1), and the first thing that needs them is the rows-below TRSM of panel j+1 on st3 -- so it
  // goes to st3 itself: one hardware queue less.  (The process degrades by 20-40 % at N <= 16384 once
  // eight queues are in use -- null stream + this handle's + the application's; scripts/dev/queue_pattern.py.)
  hipStream_t sm = trailing_stream(s), sn = (depth == 1 && s->st3) ? s->st3 : s->st4;
  hipStream_t sp = s->st2;
  // Which stream a wide update W(j) runs on when `sm` is the CU-masked one (Np < 24576: 32 CUs kept free for the chain, worth
  // 12 % of the SYRK's rate): the reservation pays where the step is bound by the chain -- the late panels -- and costs where
  // it is bound by W(j) itself, the first panels, whose chain ends long before their update does.  W(j) longer than
  // GH_FULLCHIP_MS (estimated at 60 TFLOP/s) goes to the unmasked main stream; consecutive updates on different streams are
  // ordered through ev_w, and the chain's K = 128 GEMMs follow (t_gemm_small_lds below).  Measured (profiles/r04/schedule_ab.md):
  // N = 16384 29.08 -> 28.57 ms, 20480 51.9 -> 50.1, 24064 80.9 -> 76.8; a threshold of 1.2 ms loses 1.6 % at 12288 / 16384;
  // the same rule above Np = 24576 (late panels on the masked stream) is worth 1.3 % at 24576, nothing at 32768 and 65536 --
  // not the second it takes to make the masked stream.
#ifndef GH_FULLCHIP_MS
#define GH_FULLCHIP_MS 2.5
#endif
  const std::vector<int64_t> pc = panel_starts(s);
  auto wide_stream = [&](int j) -> hipStream_t {
    if (sm == s->st) return sm;
    const int cwj = j + depth + 1;
    if (j < 0 || cwj > (int)pc.size() - 2) return sm;
    const double m2 = (double)(s->np - pc[cwj]);
    const double ms = (m2 / T) * (m2 / T + 1.0) / 2.0 * 2.0 * T * T * (double)(pc[j + 1] - pc[j]) / 60e12 * 1e3;
    return ms > GH_FULLCHIP_MS ? s->st : sm;
  };
  hipStream_t sw_prev = nullptr;
  double* A = s->A.d();
  const int64_t np = s->np, ld = np;
  const int P = (int)pc.size() - 1;
  // (measured at N = 16384, depth 1: whole block column on the chain 34.8 ms; diagonal block on the chain + rows
  //  below on the near stream 35.6; the same with the chain on CUs of its own 48.5 -- retired, scripts/dev/arms/)
  const bool prof = s->opts.profile != 0;
  while ((int)s->ev_p.size() < P) {
    hipEvent_t e[3];
    for (auto& x : e) GH_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    s->ev_p.push_back(e[0]); s->ev_w.push_back(e[1]); s->ev_nf.push_back(e[2]);
  }
  auto c0 = [&](int c) { return pc[c]; };
  auto nbc = [&](int c) { return pc[c + 1] - pc[c]; };
  auto narrow = [&](hipStream_t st, int j, int c) -> int {         // U(j, c)
    const double* Pj = blk(A, ld, c0(c), c0(j));
    const long eu = prof ? s->next_ev() : -1;
    if (eu >= 0) { GH_HIP(hipEventRecord(s->ev_pool[eu].a, st)); s->ev_update.push_back((size_t)eu); }
    GH_CHECK(gemm_nt(st, blk(A, ld, c0(c), c0(c)), ld, Pj, ld, Pj, ld, np - c0(c), nbc(c), nbc(j), -1.0, 1.0, false));
    if (eu >= 0) GH_HIP(hipEventRecord(s->ev_pool[eu].b, st));
    // (algorithmic flops of the block column: its lower part only -- the strictly-upper tiles of the
    //  diagonal block are computed for convenience and never read)
    const double tr = (double)(np - c0(c)) / T, tc = (double)nbc(c) / T;
    const double fl = (tr * tc - tc * (tc - 1.0) / 2.0) * 2.0 * T * T * (double)nbc(j);
    s->prof.update_flops += fl;
    if (eu >= 0) s->ev_update_flops.push_back(fl);
    return GH_OK;
  };
  // everything queued so far (the build, on s->st) before any of the three streams starts
  if (!s->build_on_chain) {
    GH_HIP(hipEventRecord(s->ev_sync[0], s->st));
    GH_HIP(hipStreamWaitEvent(sp, s->ev_sync[0], 0));
    GH_HIP(hipStreamWaitEvent(sn, s->ev_sync[0], 0));
    if (sm != s->st) GH_HIP(hipStreamWaitEvent(sm, s->ev_sync[0], 0));
  }   // (else the build is the chain stream's own work: the rows-below stream follows ev_aux, the update streams ev_p[0])
  for (int j = 0; j < P; ++j) {
    // ---- chain: column j is complete once U(j-1, j) has run (issued at the end of the previous turn)
    {
      const long ep = prof ? s->next_ev() : -1;
      if (ep >= 0) { GH_HIP(hipEventRecord(s->ev_pool[ep].a, sp)); s->ev_panel.push_back((size_t)ep); }
      // (panel j runs beside W(j-1): where that update owns every CU, the chain's K = 128 GEMMs keep to 32 KiB of LDS -- t_gemm_small_lds)
      t_gemm_small_lds = wide_stream(j >= 1 ? j - 1 : 0) == s->st;
      GH_CHECK(panel_step(s, sp, c0(j), nbc(j)));
      if (ep >= 0) GH_HIP(hipEventRecord(s->ev_pool[ep].b, sp));
      GH_HIP(hipEventRecord(s->ev_p[j], sp));
    }
    if (j + 1 >= P) break;
    // ---- U(j, j+1): the whole block column on the chain stream (its diagonal block is what the potf2 chain of
    // panel j+1 needs, the rows below it what that panel's rows-below TRSM needs)
    const hipEvent_t prev = (j >= 1) ? (depth >= 2 ? s->ev_nf[j - 1] : s->ev_w[j - 1]) : nullptr;
    if (prev) GH_HIP(hipStreamWaitEvent(sp, prev, 0));
    GH_CHECK(narrow(sp, j, j + 1));
    // ---- U(j, j+2 .. j+d) on the near stream
    const int last_near = std::min(j + depth, P - 1);
    if (j + 2 <= last_near || (depth >= 2 && j + 2 <= P - 1)) GH_HIP(hipStreamWaitEvent(sn, s->ev_p[j], 0));
    for (int c = j + 2; c <= last_near; ++c) {
      if (c == j + depth && j >= 1) GH_HIP(hipStreamWaitEvent(sn, s->ev_w[j - 1], 0));      // column c was inside W(j-1)
      GH_CHECK(narrow(sn, j, c));
      if (c == j + 2) GH_HIP(hipEventRecord(s->ev_nf[j], sn));
    }
    if (depth >= 2 && j + 2 > last_near) GH_HIP(hipEventRecord(s->ev_nf[j], sn));        // (nothing to do: keep the event defined)
    // ---- W(j): the rest, one lower-triangular SYRK on the main stream
    const int cw = j + depth + 1;
    hipStream_t sw = wide_stream(j);
    if (sw_prev && sw_prev != sw && j >= 1) GH_HIP(hipStreamWaitEvent(sw, s->ev_w[j - 1], 0));      // W(j-1) ran on the other stream
    sw_prev = sw;
    GH_HIP(hipStreamWaitEvent(sw, s->ev_p[j], 0));
    if (cw <= P - 1) {
      const int64_t kw = c0(cw), m2 = np - kw;
      const long et = prof ? s->next_ev() : -1;
      if (et >= 0) {
        GH_HIP(hipEventRecord(s->ev_pool[et].a, sw)); s->ev_trailing.push_back((size_t)et); s->ev_update.push_back((size_t)et);
        s->ev_update_flops.push_back((double)(m2 / T) * (m2 / T + 1) / 2.0 * 2.0 * T * T * (double)nbc(j));
      }
      const double* P2 = blk(A, ld, kw, c0(j));
      GH_CHECK(gemm_nt(sw, blk(A, ld, kw, kw), ld, P2, ld, P2, ld, m2, m2, nbc(j), -1.0, 1.0, true));
      if (et >= 0) GH_HIP(hipEventRecord(s->ev_pool[et].b, sw));
      const double tiles = (double)(m2 / T) * (m2 / T + 1) / 2.0;
      s->prof.trailing_flops += tiles * 2.0 * T * T * (double)nbc(j);
      s->prof.update_flops += tiles * 2.0 * T * T * (double)nbc(j);
      s->prof.n_trailing += 1;
    }
    GH_HIP(hipEventRecord(s->ev_w[j], sw));
  }
  // JOIN -- on the CHAIN stream, not on the main stream.  The host is far ahead of the device here, and a
  // hipStreamWaitEvent on the main stream issued now would sit at the head of that queue as a barrier packet for the whole
  // factorisation.  That is not free: whichever stream shares a DISPATCHER with the main stream (gh_debug_stream_dispatch;
  // which one does depends on how many queues the process made before) is served between polls of that barrier -- with the
  // chain or the rows-below stream there, N = 8192 took 11.0-12.8 instead of 7.2 ms per step (an application with five or
  // six streams of its own: profiles/r03/stream_placement_states.txt).  At the END of the chain's queue the same barriers
  // hold nothing up: the queue reaches them after its own last panel.  The caller continues on s->tail (log-det, copies).
  // (Waiting on the host for the last panel and joining on the main stream then works too, but costs 0.4 ms per step with a
  //  blocking wait and slows the device work by ~0.6 % when the host polls the event.)
  // In the placement a process gets that made no queues before, the main stream shares its dispatcher with the CU-masked
  // stream only, and there the join on the main stream is the faster one (N = 8192: 7.0 vs 7.3 ms, same box): the chain
  // join is used when the main stream was FOUND to share a dispatcher with a panel stream when the set was made
  // (gh_shared_main_crowded), and always with streams of the handle's own.
  const bool join_on_chain = !s->shared_streams || gh_shared_main_crowded(s->opts.device);
  if (sm != s->st && join_on_chain) {
    GH_HIP(hipEventRecord(s->ev_sync[2], sn));
    GH_HIP(hipStreamWaitEvent(sp, s->ev_sync[2], 0));
    GH_HIP(hipEventRecord(s->ev_xfer, sm));
    GH_HIP(hipStreamWaitEvent(sp, s->ev_xfer, 0));
    if (P >= 2) GH_HIP(hipStreamWaitEvent(sp, s->ev_w[P - 2], 0));          // (whichever stream the last wide update ran on)
    s->tail = sp;
    return GH_OK;
  }
  // (the trailing updates ran on the main stream itself -- large matrices -- or the old arm: the main stream joins)
  GH_HIP(hipEventRecord(s->ev_sync[1], sp));
  GH_HIP(hipStreamWaitEvent(s->st, s->ev_sync[1], 0));
  GH_HIP(hipEventRecord(s->ev_sync[2], sn));
  GH_HIP(hipStreamWaitEvent(s->st, s->ev_sync[2], 0));
  if (sm != s->st) {
    GH_HIP(hipEventRecord(s->ev_xfer, sm));
    GH_HIP(hipStreamWaitEvent(s->st, s->ev_xfer, 0));
  }
  return GH_OK;
}

#ifndef GH_LOOKAHEAD_DEPTH
#define GH_LOOKAHEAD_DEPTH 1
#endif
static int lookahead_depth(const gh_chol* s) {
  // depth 1 in this formulation (block column j+1 updated on the chain stream itself, the wide SYRK
  // alone on the main stream) beats the older scheme (block column on the main stream, depth "0") by
  // 7-12 % from N = 4096 to 16384 and is level with it above; deeper windows lose (size sweep in
  // profiles/r02/lookahead_depth_sweep.md): the narrow GEMMs of the window compete with the potf2 chain.
  // (-DGH_LOOKAHEAD_DEPTH=<d> builds the deeper windows for an A/B; an environment switch until round 5)
  return GH_LOOKAHEAD_DEPTH;
}

static int factor(gh_chol* s) {
  struct Guard { bool prev; Guard(bool v) : prev(t_gemm_small_lds) { t_gemm_small_lds = v; } ~Guard() { t_gemm_small_lds = prev; } }
      guard(s->opts.lookahead && s->st2 && trailing_stream(s) == s->st);       // no CUs kept free of the SYRK
  // (a matrix of ONE panel has nothing to look ahead to: on the main stream it saves the two cross-stream hand-overs,
  //  ~35 us each -- a tenth of the step at N = 1024)
  if (s->opts.lookahead && s->st2 && s->st3 && s->st4 && s->np > panel_width(s)) return factor_lookahead_deep(s, lookahead_depth(s));      // (panel widths: panel_starts())
  hipStream_t st = s->st;
  double* A = s->A.d();
  const int64_t np = s->np, ld = np;
  const bool prof = s->opts.profile != 0;
  const std::vector<int64_t> pc = panel_starts(s);             // (the same panel widths as the look-ahead schedule)
  for (size_t pj = 0; pj + 1 < pc.size(); ++pj) {
    const int64_t k0 = pc[pj], nb = pc[pj + 1] - pc[pj];
    const long ep = prof ? s->next_ev() : -1;
    if (ep >= 0) { GH_HIP(hipEventRecord(s->ev_pool[ep].a, st)); s->ev_panel.push_back((size_t)ep); }
    GH_CHECK(potrf_block(st, blk(A, ld, k0, k0), ld, nb, s->dinv.d() + (k0 / T) * T * T, s->d_info, k0));
    const int64_t m = np - (k0 + nb);
    if (m > 0)
      GH_CHECK(trsm_right(st, blk(A, ld, k0, k0), ld, s->dinv.d() + (k0 / T) * T * T, blk(A, ld, k0 + nb, k0), ld, m, nb));
    if (ep >= 0) GH_HIP(hipEventRecord(s->ev_pool[ep].b, st));
    if (m > 0) {
      const long et = prof ? s->next_ev() : -1;
      if (et >= 0) { GH_HIP(hipEventRecord(s->ev_pool[et].a, st)); s->ev_trailing.push_back((size_t)et); }
      const double* P = blk(A, ld, k0 + nb, k0);
      GH_CHECK(gemm_nt(st, blk(A, ld, k0 + nb, k0 + nb), ld, P, ld, P, ld, m, m, nb, -1.0, 1.0, true));
      if (et >= 0) GH_HIP(hipEventRecord(s->ev_pool[et].b, st));
      const double tiles = (double)(m / T) * (m / T + 1) / 2.0;
      s->prof.trailing_flops += tiles * 2.0 * T * T * (double)nb;
      s->prof.n_trailing += 1;
    }
  }
  return GH_OK;
}

// x, yerr into the handle's copies and the failure word cleared, ONE launch (device-resident inputs: as two copies
// and a memset these were three runtime operations with 5-30 us between them -- a tenth of a step at N = 1024)
__global__ void prep_inputs_kernel(const double* xs, long nx, const double* es, long ne, double* xd, double* ed, long long* info) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (long e = i; e < nx; e += stride) xd[e] = xs[e];
  for (long e = i; e < ne; e += stride) ed[e] = es[e];
  if (i == 0) *info = 0;
}

// Everything of compute() up to and including the log-det launch, enqueued on s->st without a
// host synchronisation; compute_finish() reads the scalars back.
struct ComputeCtx { long e_all = -1, e_build = -1; };
static int compute_enqueue(gh_chol* s, gh_kernel* k, const double* x, int64_t n, int32_t ndim, const double* yerr, ComputeCtx& c) {
  if (!s || !k || !x || !yerr || n <= 0) { gh_set_error("bad argument to compute"); return GH_ERR_BAD_ARG; }
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  GH_CHECK(set_device(s));
  GH_CHECK(k->upload());
  s->computed = false;
  s->info = 0;
  const int64_t np = gh_round_up(n, T);
  s->n = n; s->np = np; s->ndim = ndim;
  GH_CHECK(s->A.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(s->dinv.ensure((size_t)(np / T) * T * T * sizeof(double)));
  GH_CHECK(s->x.ensure((size_t)n * ndim * sizeof(double)));
  GH_CHECK(s->yerr.ensure((size_t)n * sizeof(double)));
  GH_CHECK(s->scal.ensure(256 * sizeof(double)));      // [0] log-det, [1] quadratic form, [8..72) and [72..136) slice sums
  // With look-ahead the first thing that needs the matrix is the chain stream's first panel: inputs and kernel-matrix build go to
  // THAT stream (everything else waits for panel events that follow them in its order) instead of the main stream + a
  // cross-stream hand-over -- 19 us between the build and the first potf2 of every compute() (profiles/r06/: N = 2048 timeline).
  s->build_on_chain = g_build_on_chain && s->opts.lookahead && s->st2 && s->st3 && s->st4 && np > panel_width(s);
  hipStream_t st = s->build_on_chain ? s->st2 : s->st;
  memset(&s->prof, 0, sizeof(s->prof));
  s->ev_used = 0; s->ev_trailing.clear(); s->ev_panel.clear(); s->ev_update.clear(); s->ev_update_flops.clear();
  const bool prof = s->opts.profile != 0;
  c.e_all = prof ? s->next_ev() : -1;
  c.e_build = prof ? s->next_ev() : -1;
  if (c.e_all >= 0) GH_HIP(hipEventRecord(s->ev_pool[c.e_all].a, st));
  s->d_info = (long long*)(s->scal.d() + 2);            // beside log-det [0] and quadratic form [1]: ONE copy brings them back
  if (gh_is_device_ptr(x) && gh_is_device_ptr(yerr)) {
    const long tot = (long)n * ndim;
    hipLaunchKernelGGL(prep_inputs_kernel, dim3((unsigned)std::min<long>((tot + 255) / 256, 1024)), dim3(256), 0, st,
                       x, tot, yerr, (long)n, s->x.d(), s->yerr.d(), s->d_info);
    GH_HIP(hipGetLastError());
  } else {
    GH_CHECK(gh_to_device(s->x.d(), x, (size_t)n * ndim, st));
    GH_CHECK(gh_to_device(s->yerr.d(), yerr, (size_t)n, st));
    GH_HIP(hipMemsetAsync(s->d_info, 0, sizeof(long long), st));
  }
  if (c.e_build >= 0) GH_HIP(hipEventRecord(s->ev_pool[c.e_build].a, st));
  GH_CHECK(gh_launch_kmat(k, s->x.d(), n, s->x.d(), n, s->yerr.d(), s->A.d(), np, np, np, 0, 0, true, true, st));
  if (c.e_build >= 0) GH_HIP(hipEventRecord(s->ev_pool[c.e_build].b, st));
  s->tail = s->st;
  GH_CHECK(factor(s));                                  // (may move s->tail to the chain stream)
  GH_CHECK(launch_logdet(s->A.d(), np, np, s->scal.d(), s->scal.d() + 8, s->tail));
  if (c.e_all >= 0) GH_HIP(hipEventRecord(s->ev_pool[c.e_all].b, s->tail));
  return GH_OK;
}
// after the stream has been synchronised and logdet / info copied to the host
static int compute_finish(gh_chol* s, const ComputeCtx& c, double ld_host, long long info_host, double* logdet_out) {
  if (s->opts.profile && c.e_all >= 0 && c.e_build >= 0) {
    float ms = 0;
    GH_HIP(hipEventElapsedTime(&ms, s->ev_pool[c.e_all].a, s->ev_pool[c.e_all].b)); s->prof.ms_total = ms;
    GH_HIP(hipEventElapsedTime(&ms, s->ev_pool[c.e_build].a, s->ev_pool[c.e_build].b)); s->prof.ms_build = ms;
    for (size_t i : s->ev_trailing) { GH_HIP(hipEventElapsedTime(&ms, s->ev_pool[i].a, s->ev_pool[i].b)); s->prof.ms_trailing += ms; }
    // time during which ANY trailing-update launch (wide SYRK on the main stream, block-column GEMM on the
    // chain stream) was running: union of their event intervals, measured from the start of compute()
    if (!s->ev_update.empty()) {
      std::vector<std::pair<float, float>> iv;
      s->upd_intervals.clear();
      for (size_t q = 0; q < s->ev_update.size(); ++q) {
        const size_t i = s->ev_update[q];
        float a = 0, b = 0;
        GH_HIP(hipEventElapsedTime(&a, s->ev_pool[c.e_all].a, s->ev_pool[i].a));
        GH_HIP(hipEventElapsedTime(&b, s->ev_pool[c.e_all].a, s->ev_pool[i].b));
        iv.emplace_back(a, b);
        s->upd_intervals.push_back(a); s->upd_intervals.push_back(b);
        s->upd_intervals.push_back(q < s->ev_update_flops.size() ? s->ev_update_flops[q] : 0.0);
      }
      std::sort(iv.begin(), iv.end());
      double tot = 0.0;
      float lo = iv[0].first, hi = iv[0].second;
      for (size_t q = 1; q < iv.size(); ++q) {
        if (iv[q].first > hi) { tot += hi - lo; lo = iv[q].first; hi = iv[q].second; }
        else if (iv[q].second > hi) hi = iv[q].second;
      }
      tot += hi - lo;
      s->prof.ms_update_union = tot;
    }
    for (size_t i : s->ev_panel) { GH_HIP(hipEventElapsedTime(&ms, s->ev_pool[i].a, s->ev_pool[i].b)); s->prof.ms_panel += ms; }
  }
  if (info_host != 0) {
    s->info = info_host;
    gh_set_error("%lld-th leading minor of the array is not positive definite", info_host);
    return GH_ERR_NOT_PD;
  }
  s->logdet = ld_host;
  s->computed = true;
  if (logdet_out) *logdet_out = ld_host;
  return GH_OK;
}

extern "C" int gh_chol_compute(gh_chol* s, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                               const double* yerr, double* logdet_out) {
  ComputeCtx c;
  GH_CHECK(compute_enqueue(s, k, x, n, ndim, yerr, c));
  hipStream_t st = s->tail;                             // (the stream the factorisation ended on: compute_enqueue)
  double back[3] = {0.0, 0.0, 0.0};                     // [0] log-det, [1] (quadratic form), [2] the failure word's bits
  GH_HIP(hipMemcpyAsync(back, s->scal.d(), 3 * sizeof(double), hipMemcpyDeviceToHost, st));
  GH_HIP(hipStreamSynchronize(st));
  long long info_host = 0;
  memcpy(&info_host, &back[2], sizeof(long long));
  return compute_finish(s, c, back[0], info_host, logdet_out);
}

static int need_computed(gh_chol* s) {
  if (!s) { gh_set_error("null solver"); return GH_ERR_BAD_ARG; }
  if (!s->computed) { gh_set_error("you must call 'compute' first"); return GH_ERR_NOT_COMPUTED; }
  return set_device(s);
}

// load a length-n vector (host or device) into a zero-padded device vector of length np
static int load_vec(gh_chol* s, GhBuf& buf, const double* src) {
  GH_CHECK(buf.ensure((size_t)s->np * sizeof(double)));
  if (s->np > s->n) GH_HIP(hipMemsetAsync(buf.d() + s->n, 0, (size_t)(s->np - s->n) * sizeof(double), s->st));
  return gh_to_device(buf.d(), src, (size_t)s->n, s->st);
}
// z = L^-1 w  (w is destroyed)
// (defer: enqueue only; the caller reads the time-out flag back itself, chain_fail_flag())
static int trsv_forward(gh_chol* s, double* w, double* z, bool defer = false) {
  const int64_t nt = s->np / T;
  static const bool stepwise = getenv("GEORGE_AMD_TRSV_STEPS") != nullptr;       // A/B arm: one launch per block row
  if (!stepwise) {
    // forward and backward sweeps keep separate flag sets, [0, nt] and [nt + 1, 2 nt + 1]
    GH_CHECK(s->chain.ensure((size_t)(2 * nt + 2) * sizeof(unsigned)));
    unsigned* flags = (unsigned*)s->chain.p;
    GH_CHECK(launch_trsv_fwd_chain(s->A.d(), (long)s->np, s->dinv.d(), nt, w, z, flags, s->st));
    if (defer) return GH_OK;
    int failed = 0;
    GH_HIP(hipMemcpyAsync(&failed, flags + nt, sizeof(int), hipMemcpyDeviceToHost, s->st));
    GH_HIP(hipStreamSynchronize(s->st));
    if (failed) { gh_set_error("forward solve: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
    return GH_OK;
  }
  for (int64_t j = 0; j < nt; ++j) {
    hipLaunchKernelGGL(trsv_fwd_step, dim3((unsigned)(nt - j)), dim3(256), 0, s->st,
                       s->A.d(), (long)s->np, s->dinv.d() + j * T * T, (long)(j * T), w, z);
  }
  GH_HIP(hipGetLastError());
  return GH_OK;
}
// x = L^-T w  (w is destroyed)
static int trsv_backward(gh_chol* s, double* w, double* x, bool defer = false) {
  const int64_t nt = s->np / T;
  static const bool stepwise = getenv("GEORGE_AMD_TRSV_STEPS") != nullptr;
  if (!stepwise) {
    GH_CHECK(s->chain.ensure((size_t)(2 * nt + 2) * sizeof(unsigned)));
    unsigned* flags = (unsigned*)s->chain.p + (nt + 1);
    GH_CHECK(launch_trsv_bwd_chain(s->A.d(), (long)s->np, s->dinv.d(), nt, w, x, flags, s->st));
    if (defer) return GH_OK;
    int failed = 0;
    GH_HIP(hipMemcpyAsync(&failed, flags + nt, sizeof(int), hipMemcpyDeviceToHost, s->st));
    GH_HIP(hipStreamSynchronize(s->st));
    if (failed) { gh_set_error("backward solve: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
    return GH_OK;
  }
  for (int64_t j = nt - 1; j >= 0; --j) {
    hipLaunchKernelGGL(trsv_bwd_step, dim3((unsigned)(j + 1)), dim3(256), 0, s->st,
                       s->A.d(), (long)s->np, s->dinv.d() + j * T * T, (long)(j * T), w, x);
  }
  GH_HIP(hipGetLastError());
  return GH_OK;
}

extern "C" int gh_chol_dot_solve(gh_chol* s, const double* y, double* out) {
  GH_CHECK(need_computed(s));
  if (!y || !out) { gh_set_error("null argument"); return GH_ERR_BAD_ARG; }
  // y^T K^-1 y = || L^-1 y ||^2 : one forward sweep (the reference does both, basic.py:102)
  static const bool stepwise = getenv("GEORGE_AMD_TRSV_STEPS") != nullptr;
  // (the chained kernel only READS its right-hand side: a device-resident y of full padded length is used where it lies)
  const bool direct = !stepwise && s->np == s->n && gh_is_device_ptr(y);
  if (!direct) GH_CHECK(load_vec(s, s->v0, y));
  GH_CHECK(s->v1.ensure((size_t)s->np * sizeof(double)));
  const long e = s->opts.profile ? s->next_ev() : -1;
  if (e >= 0) GH_HIP(hipEventRecord(s->ev_pool[e].a, s->st));
  GH_CHECK(trsv_forward(s, direct ? const_cast<double*>(y) : s->v0.d(), s->v1.d(), !stepwise));
  const int* fail = stepwise ? nullptr : (const int*)((unsigned*)s->chain.p + s->np / T);
  GH_CHECK(launch_dot(s->v1.d(), s->v1.d(), (long)s->np, s->scal.d() + 1, s->scal.d() + 72, s->st, fail));
  if (e >= 0) GH_HIP(hipEventRecord(s->ev_pool[e].b, s->st));
  double back[3] = {0.0, 0.0, 0.0};                     // [0] quadratic form, [1] (failure word of compute()), [2] the chain's time-out flag
  GH_HIP(hipMemcpyAsync(back, s->scal.d() + 1, 3 * sizeof(double), hipMemcpyDeviceToHost, s->st));
  GH_HIP(hipStreamSynchronize(s->st));
  if (e >= 0) { float ms = 0; GH_HIP(hipEventElapsedTime(&ms, s->ev_pool[e].a, s->ev_pool[e].b)); s->prof.ms_solve = ms; }
  if (!stepwise && back[2] != 0.0) { gh_set_error("forward solve: a workgroup waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
  *out = back[0];
  return GH_OK;
}

// B (np x rp, row-major, zero padded) <- L^-1 B  (forward) and optionally L^-T (backward)
// Two-level blocking: 128-row steps (multiplication by the stored diagonal inverse + a small
// update) inside super-blocks of SB = 8 tiles, then ONE K = 128*SB update of everything below (above)
// the super-block -- the right-hand side is swept N/(128*SB) times instead of N/128 times.
// `tri`: B is the identity being overwritten by L^-1 (forward only): block row j is non-zero in
// columns [0, (j+1)*128) only, so every product is clipped to those columns.
static int trsm_multi(gh_chol* s, double* B, int64_t rp, bool forward, bool backward, bool tri = false) {
  // (tiles per super-block; measured at N = 32768 with 4096 right-hand sides, both sweeps: 2 -> 167 ms, 4 -> 158.5,
  //  8 -> 151.6, 16 -> 149.5; no difference at N = 8192; the switch that overrode it went in round 4)
  const int64_t SB = 8;
  const int64_t np = s->np, nt = np / T;
  const double* L = s->A.d();
  auto mm = [&](double* Cp, const double* Ap, int64_t lda, bool a_km, const double* Bp, int64_t M, int64_t N, int64_t K,
                double alpha, double beta) -> int {
    if (M <= 0 || N <= 0) return GH_OK;
    GhGemm g{};
    g.C = Cp; g.ldc = rp; g.A = Ap; g.lda = lda; g.B = Bp; g.ldb = rp;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.a_km = a_km; g.b_km = false;
    return gh_launch_gemm(g, s->st);
  };
  if (forward) {
    for (int64_t J = 0; J < nt; J += SB) {
      const int64_t Je = std::min<int64_t>(J + SB, nt);
      for (int64_t j = J; j < Je; ++j) {
        double* Bj = B + j * T * rp;
        const int64_t nc = tri ? (j + 1) * T : rp;
        GH_CHECK(mm(Bj, s->dinv.d() + j * T * T, T, true, Bj, T, nc, T, 1.0, 0.0));               // B_j <- L_jj^-1 B_j
        GH_CHECK(mm(B + (j + 1) * T * rp, L + (j + 1) * T * np + j * T, np, true, Bj,
                    (Je - j - 1) * T, nc, T, -1.0, 1.0));                                           // rows of the super-block
      }
      const int64_t nc = tri ? Je * T : rp;
      GH_CHECK(mm(B + Je * T * rp, L + Je * T * np + J * T, np, true, B + J * T * rp,
                  (nt - Je) * T, nc, (Je - J) * T, -1.0, 1.0));                                     // everything below
    }
  }
  if (backward) {
    for (int64_t Je = nt; Je > 0; Je -= SB) {
      const int64_t J = std::max<int64_t>(Je - SB, 0);
      for (int64_t j = Je - 1; j >= J; --j) {
        double* Bj = B + j * T * rp;
        GH_CHECK(mm(Bj, s->dinv.d() + j * T * T, T, false, Bj, T, rp, T, 1.0, 0.0));              // B_j <- L_jj^-T B_j
        GH_CHECK(mm(B + J * T * rp, L + j * T * np + J * T, np, false, Bj, (j - J) * T, rp, T, -1.0, 1.0));
      }
      GH_CHECK(mm(B, L + J * T * np, np, false, B + J * T * rp, J * T, rp, (Je - J) * T, -1.0, 1.0));   // everything above
    }
  }
  return GH_OK;
}

extern "C" int gh_chol_solve(gh_chol* s, const double* b, int64_t nrhs, double* out) {
  GH_CHECK(need_computed(s));
  if (!b || !out || nrhs <= 0) { gh_set_error("bad argument to solve"); return GH_ERR_BAD_ARG; }
  const int64_t n = s->n, np = s->np;
  if (nrhs == 1) {
    GH_CHECK(load_vec(s, s->v0, b));
    GH_CHECK(s->v1.ensure((size_t)np * sizeof(double)));
    GH_CHECK(s->v2.ensure((size_t)np * sizeof(double)));
    GH_CHECK(trsv_forward(s, s->v0.d(), s->v1.d()));
    GH_CHECK(trsv_backward(s, s->v1.d(), s->v2.d()));
    return gh_from_device(out, s->v2.d(), (size_t)n, s->st);
  }
  const int64_t rp = gh_round_up(nrhs, T);
  GH_CHECK(s->rhs.ensure((size_t)np * rp * sizeof(double)));
  GH_HIP(hipMemsetAsync(s->rhs.d(), 0, (size_t)np * rp * sizeof(double), s->st));
  GH_HIP(hipMemcpy2DAsync(s->rhs.d(), rp * sizeof(double), b, nrhs * sizeof(double), nrhs * sizeof(double), n,
                          gh_is_device_ptr(b) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st));
  GH_CHECK(trsm_multi(s, s->rhs.d(), rp, true, true));
  GH_HIP(hipMemcpy2DAsync(out, nrhs * sizeof(double), s->rhs.d(), rp * sizeof(double), nrhs * sizeof(double), n,
                          gh_is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->st));
  GH_HIP(hipStreamSynchronize(s->st));
  return GH_OK;
}

extern "C" int gh_chol_apply_sqrt(gh_chol* s, const double* r, int64_t nrows, double* out) {
  GH_CHECK(need_computed(s));
  if (!r || !out || nrows <= 0) { gh_set_error("bad argument to apply_sqrt"); return GH_ERR_BAD_ARG; }
  // out = r @ U with U = L^T (basic.py:114):  out[s][j] = sum_{k <= j} r[s][k] L[j][k]
  const int64_t n = s->n, np = s->np, rr = gh_round_up(nrows, T);
  GH_CHECK(s->rhs.ensure((size_t)rr * np * sizeof(double)));
  GH_CHECK(s->work.ensure((size_t)rr * np * sizeof(double)));
  GH_HIP(hipMemsetAsync(s->rhs.d(), 0, (size_t)rr * np * sizeof(double), s->st));
  GH_HIP(hipMemcpy2DAsync(s->rhs.d(), np * sizeof(double), r, n * sizeof(double), n * sizeof(double), nrows,
                          gh_is_device_ptr(r) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st));
  GhGemm g{};
  g.C = s->work.d(); g.ldc = np; g.A = s->rhs.d(); g.lda = np; g.B = s->A.d(); g.ldb = np;
  g.M = rr; g.N = np; g.K = np; g.alpha = 1.0; g.beta = 0.0; g.a_km = true; g.b_km = true; g.khi_col = true;
  GH_CHECK(gh_launch_gemm(g, s->st));
  GH_HIP(hipMemcpy2DAsync(out, n * sizeof(double), s->work.d(), np * sizeof(double), n * sizeof(double), nrows,
                          gh_is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->st));
  GH_HIP(hipStreamSynchronize(s->st));
  return GH_OK;
}

// W (np x np) <- K^-1, lower triangle valid.  Uses K^-1 = L^-T L^-1:
//   Linv = L^-1 (forward substitution on the identity), K^-1 = Linv^T Linv (k >= max(i, j)).
static int inverse_lower(gh_chol* s, double* W /* np*np */, double* Linv /* np*np scratch */) {
  const int64_t np = s->np, nt = np / T;
  const double* L = s->A.d();
  GH_HIP(hipMemsetAsync(Linv, 0, (size_t)np * np * sizeof(double), s->st));
  hipLaunchKernelGGL(eye_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s->st, Linv, (long)np, (long)np);
  GH_HIP(hipGetLastError());
  // forward substitution exploiting the lower-triangular right-hand side (tri = true)
  GH_CHECK(trsm_multi(s, Linv, np, true, false, true));
  GhGemm q{};
  q.C = W; q.ldc = np; q.A = Linv; q.lda = np; q.B = Linv; q.ldb = np;
  q.M = np; q.N = np; q.K = np; q.alpha = 1.0; q.beta = 0.0; q.a_km = false; q.b_km = false;
  q.lower = true; q.klo_max = true;
  return gh_launch_gemm(q, s->st);
}

extern "C" int gh_chol_get_inverse(gh_chol* s, double* out) {
  GH_CHECK(need_computed(s));
  if (!out) { gh_set_error("null output"); return GH_ERR_BAD_ARG; }
  const int64_t n = s->n, np = s->np;
  GH_CHECK(s->work.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(s->work2.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(inverse_lower(s, s->work.d(), s->work2.d()));
  // full symmetric n x n result (work2 is free again)
  double* full = s->work2.d();
  const long tot = (long)n * n;
  hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->st, s->work.d(), (long)np, full, (long)n, (long)n);
  GH_HIP(hipGetLastError());
  return gh_from_device(out, full, (size_t)tot, s->st);
}

extern "C" int gh_chol_predict(gh_chol* s, gh_kernel* k, const double* r, const double* xs, int64_t m,
                               double* mu, double* var, double* cov) {
  GH_CHECK(need_computed(s));
  if (!k || !r || !xs || !mu || m <= 0) { gh_set_error("bad argument to predict"); return GH_ERR_BAD_ARG; }
  if (k->ndim != s->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  GH_CHECK(k->upload());
  const int64_t n = s->n, np = s->np, mp = gh_round_up(m, T);
  hipStream_t st = s->st;
  // z = L^-1 r
  GH_CHECK(load_vec(s, s->v0, r));
  GH_CHECK(s->v1.ensure((size_t)np * sizeof(double)));
  GH_CHECK(trsv_forward(s, s->v0.d(), s->v1.d()));
  // V = L^-1 K(x, xs)   (np x mp): built on the device, forward substitution only, because
  // K*s K^-1 K*s^T = V^T V and K*s K^-1 r = V^T z  (gp.py:532-545 does both sweeps on the host)
  GhBuf xsd;
  const double* xs_dev = xs;
  if (!gh_is_device_ptr(xs)) {
    GH_CHECK(xsd.ensure((size_t)m * s->ndim * sizeof(double)));
    GH_CHECK(gh_to_device(xsd.d(), xs, (size_t)m * s->ndim, st));
    xs_dev = xsd.d();
  }
  GH_CHECK(s->rhs.ensure((size_t)np * mp * sizeof(double)));
  GH_CHECK(gh_launch_kmat(k, s->x.d(), n, xs_dev, m, nullptr, s->rhs.d(), mp, np, mp, 0, 0, false, false, st));
  GH_CHECK(trsm_multi(s, s->rhs.d(), mp, true, false));
  // column reductions
  const int64_t nchunks = std::min<int64_t>(64, np / T);
  const int64_t rows_per = gh_round_up((np + nchunks - 1) / nchunks, 1);
  GH_CHECK(s->scratch.ensure((size_t)(2 * nchunks * mp + 2 * mp) * sizeof(double)));
  double* pmu = s->scratch.d();
  double* pvar = pmu + nchunks * mp;
  double* dmu = pvar + nchunks * mp;
  double* dvar = dmu + mp;
  hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((mp + 255) / 256), (unsigned)nchunks), dim3(256), 0, st,
                     s->rhs.d(), (long)mp, (long)np, (long)rows_per, s->v1.d(), pmu, pvar, (long)mp);
  GH_HIP(hipGetLastError());
  if (var) GH_CHECK(gh_launch_kdiag(k, xs_dev, xs_dev, m, dvar, st));           // gp.py:539
  hipLaunchKernelGGL(colfinal_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
                     pmu, pvar, (long)nchunks, (long)mp, (long)m, dmu, var ? dvar : nullptr);
  GH_HIP(hipGetLastError());
  GH_CHECK(gh_from_device(mu, dmu, (size_t)m, st));
  if (var) GH_CHECK(gh_from_device(var, dvar, (size_t)m, st));
  if (cov) {
    // cov = K(xs, xs) - V^T V      (gp.py:543-545)
    GH_CHECK(s->work.ensure((size_t)mp * mp * sizeof(double)));
    GH_CHECK(gh_launch_kmat(k, xs_dev, m, xs_dev, m, nullptr, s->work.d(), mp, mp, mp, 0, 0, true, false, st));
    GhGemm g{};
    g.C = s->work.d(); g.ldc = mp; g.A = s->rhs.d(); g.lda = mp; g.B = s->rhs.d(); g.ldb = mp;
    g.M = mp; g.N = mp; g.K = np; g.alpha = -1.0; g.beta = 1.0; g.a_km = false; g.b_km = false;
    GH_CHECK(gh_launch_gemm(g, st));
    GH_HIP(hipMemcpy2DAsync(cov, m * sizeof(double), s->work.d(), mp * sizeof(double), m * sizeof(double), m,
                            gh_is_device_ptr(cov) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
  }
  GH_HIP(hipStreamSynchronize(st));
  return GH_OK;
}

extern "C" int gh_chol_grad(gh_chol* s, gh_kernel* k, const uint32_t* which, const double* r,
                            double* grad, double* alpha, double* diagA) {
  GH_CHECK(need_computed(s));
  if (!k || !which || !r || !grad) { gh_set_error("bad argument to grad"); return GH_ERR_BAD_ARG; }
  if (k->ndim != s->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  GH_CHECK(k->upload());
  const int64_t n = s->n, np = s->np;
  hipStream_t st = s->st;
  // alpha = K^-1 r                       (gp.py:429)
  GH_CHECK(load_vec(s, s->v0, r));
  GH_CHECK(s->v1.ensure((size_t)np * sizeof(double)));
  GH_CHECK(s->v2.ensure((size_t)np * sizeof(double)));
  GH_CHECK(trsv_forward(s, s->v0.d(), s->v1.d()));
  GH_CHECK(trsv_backward(s, s->v1.d(), s->v2.d()));
  // K^-1 (lower)                         (gp.py:436)
  GH_CHECK(s->work.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(s->work2.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(inverse_lower(s, s->work.d(), s->work2.d()));
  // 1/2 sum_ij A_ij dK_ij/dtheta, A = alpha alpha^T - K^-1   (gp.py:437,465-466), fused
  GH_CHECK(s->v0.ensure((size_t)std::max<int64_t>(np, GH_MAX_GRAD) * sizeof(double)));
  double* dgrad = s->v0.d();                       // v0 is free again
  double* ddiag = s->v1.d();                       // so is v1
  GH_CHECK(gh_launch_kgrad_reduce(k, which, s->x.d(), n, s->v2.d(), s->work.d(), np, dgrad, ddiag, s->scratch, st));
  if (k->size > 0) GH_CHECK(gh_from_device(grad, dgrad, (size_t)k->size, st));
  if (alpha) GH_CHECK(gh_from_device(alpha, s->v2.d(), (size_t)n, st));
  if (diagA) GH_CHECK(gh_from_device(diagA, ddiag, (size_t)n, st));
  GH_HIP(hipStreamSynchronize(st));
  return GH_OK;
}

// ============================================================ fused objective
// nll and its gradient (gp.py:470-480; the optimiser loop of docs/tutorials/hyper.rst:131-152) as
// ONE call: build K -> factor -> log-det -> z = L^-1 r (used for r^T K^-1 r = |z|^2 AND, through
// the backward sweep, for alpha) -> K^-1 -> 1/2 sum A_ij dK_ij/dtheta.  Nothing is synchronised
// until the very end; only scalars and N-vectors reach the host.  `grad == NULL`: the
// log-likelihood pieces only (compute + dot_solve without the second sweep and the inverse).
extern "C" int gh_chol_objective(gh_chol* s, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                                 const double* yerr, const double* r, const uint32_t* which,
                                 double* logdet, double* quad, double* grad, double* alpha, double* diagA) {
  if (!r || !logdet || !quad) { gh_set_error("bad argument to objective"); return GH_ERR_BAD_ARG; }
  if (grad && !which) { gh_set_error("objective: gradient requested without a parameter mask"); return GH_ERR_BAD_ARG; }
  ComputeCtx c;
  GH_CHECK(compute_enqueue(s, k, x, n, ndim, yerr, c));
  hipStream_t st = s->st;
  if (s->tail && s->tail != st && s->ev_sync[1]) {      // (the solves and the gradient run on the main stream: it joins here)
    GH_HIP(hipEventRecord(s->ev_sync[1], s->tail));
    GH_HIP(hipStreamWaitEvent(st, s->ev_sync[1], 0));
  }
  const int64_t np = s->np, nt = np / T;
  const bool want_alpha = grad || alpha || diagA;
  GH_CHECK(load_vec(s, s->v0, r));
  GH_CHECK(s->v1.ensure((size_t)np * sizeof(double)));
  static const bool stepwise = getenv("GEORGE_AMD_TRSV_STEPS") != nullptr;
  GH_CHECK(trsv_forward(s, s->v0.d(), s->v1.d(), true));
  GH_CHECK(launch_dot(s->v1.d(), s->v1.d(), (long)np, s->scal.d() + 1, s->scal.d() + 72, st,
                      stepwise ? nullptr : (const int*)((const unsigned*)s->chain.p + nt)));           // -> scal[3]
  double* dgrad = nullptr;
  double* ddiag = nullptr;
  if (want_alpha) {
    GH_CHECK(s->v2.ensure((size_t)np * sizeof(double)));
    GH_CHECK(trsv_backward(s, s->v1.d(), s->v2.d(), true));                          // alpha (v1 is consumed)
  }
  if (grad || diagA) {
    GH_CHECK(s->work.ensure((size_t)np * np * sizeof(double)));
    GH_CHECK(s->work2.ensure((size_t)np * np * sizeof(double)));
    GH_CHECK(inverse_lower(s, s->work.d(), s->work2.d()));
    GH_CHECK(s->v0.ensure((size_t)std::max<int64_t>(np, GH_MAX_GRAD) * sizeof(double)));
    dgrad = s->v0.d();
    ddiag = s->v1.d();
    static const uint32_t none[GH_MAX_GRAD] = {0};
    GH_CHECK(gh_launch_kgrad_reduce(k, grad ? which : none, s->x.d(), n, s->v2.d(), s->work.d(), np, dgrad, ddiag, s->scratch, st));
  }
  double host[4] = {0.0, 0.0, 0.0, 0.0};                // log-det, quadratic form, failure word (bits), forward chain's time-out flag
  int fail_b = 0;
  GH_HIP(hipMemcpyAsync(host, s->scal.d(), 4 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (!stepwise && want_alpha)
    GH_HIP(hipMemcpyAsync(&fail_b, (const unsigned*)s->chain.p + (nt + 1) + nt, sizeof(int), hipMemcpyDeviceToHost, st));
  if (grad && k->size > 0) GH_CHECK(gh_from_device(grad, dgrad, (size_t)k->size, st));
  if (alpha) GH_CHECK(gh_from_device(alpha, s->v2.d(), (size_t)n, st));
  if (diagA) GH_CHECK(gh_from_device(diagA, ddiag, (size_t)n, st));
  GH_HIP(hipStreamSynchronize(st));
  long long info_host = 0;
  memcpy(&info_host, &host[2], sizeof(long long));
  GH_CHECK(compute_finish(s, c, host[0], info_host, logdet));
  if ((!stepwise && host[3] != 0.0) || fail_b) { gh_set_error("objective: a chained solve waited more than 2 s for its predecessor"); return GH_ERR_HIP; }
  *quad = host[1];
  return GH_OK;
}

// ============================================================ factor export / import
// The reference's BasicSolver survives pickling COMPUTED (tests/test_pickle.py:21-36: its factor is a
// NumPy array).  Here the factor lives in HBM, so it is packed on the device -- row i of the lower
// triangle at offset i (i + 1) / 2, N (N + 1) / 2 doubles -- and copied out, together with the
// inverses of the 128 x 128 diagonal blocks (Np / 128 x 128 x 128) that every solve multiplies by.
__global__ void pack_lower_kernel(const double* A, long ld, long n, double* out) {
  const long i = blockIdx.x;
  const double* row = A + i * ld;
  double* o = out + i * (i + 1) / 2;
  for (long j = threadIdx.x; j <= i; j += blockDim.x) o[j] = row[j];
}
__global__ void unpack_lower_kernel(const double* in, long n, double* A, long ld, long np) {
  const long i = blockIdx.x;                     // row of the padded matrix
  double* row = A + i * ld;
  if (i < n) {
    const double* src = in + i * (i + 1) / 2;
    for (long j = threadIdx.x; j < np; j += blockDim.x) row[j] = (j <= i) ? src[j] : 0.0;
  } else {
    for (long j = threadIdx.x; j < np; j += blockDim.x) row[j] = (j == i) ? 1.0 : 0.0;      // identity padding
  }
}
extern "C" int64_t gh_chol_factor_size(const gh_chol* s) { return s ? s->n * (s->n + 1) / 2 : 0; }
extern "C" int64_t gh_chol_dinv_size(const gh_chol* s) { return s ? (s->np / T) * T * T : 0; }
extern "C" int gh_chol_export_factor(gh_chol* s, double* packed_lower, double* dinv_out) {
  GH_CHECK(need_computed(s));
  if (!packed_lower || !dinv_out) { gh_set_error("null output"); return GH_ERR_BAD_ARG; }
  const int64_t n = s->n, np = s->np;
  const size_t cnt = (size_t)n * (n + 1) / 2;
  GH_CHECK(s->work.ensure(cnt * sizeof(double)));
  hipLaunchKernelGGL(pack_lower_kernel, dim3((unsigned)n), dim3(256), 0, s->st, s->A.d(), (long)np, (long)n, s->work.d());
  GH_HIP(hipGetLastError());
  GH_CHECK(gh_from_device(packed_lower, s->work.d(), cnt, s->st));
  GH_CHECK(gh_from_device(dinv_out, s->dinv.d(), (size_t)(np / T) * T * T, s->st));
  GH_HIP(hipStreamSynchronize(s->st));
  return GH_OK;
}
extern "C" int gh_chol_import_factor(gh_chol* s, int64_t n, int32_t ndim, const double* x, const double* packed_lower,
                                     const double* dinv_in, double logdet) {
  if (!s || n <= 0 || ndim <= 0 || !x || !packed_lower || !dinv_in) { gh_set_error("bad argument to import_factor"); return GH_ERR_BAD_ARG; }
  GH_CHECK(set_device(s));
  s->computed = false;
  const int64_t np = gh_round_up(n, T);
  s->n = n; s->np = np; s->ndim = ndim; s->info = 0;
  const size_t cnt = (size_t)n * (n + 1) / 2;
  GH_CHECK(s->A.ensure((size_t)np * np * sizeof(double)));
  GH_CHECK(s->dinv.ensure((size_t)(np / T) * T * T * sizeof(double)));
  GH_CHECK(s->x.ensure((size_t)n * ndim * sizeof(double)));
  GH_CHECK(s->scal.ensure(256 * sizeof(double)));
  GH_CHECK(s->work.ensure(cnt * sizeof(double)));
  GH_CHECK(gh_to_device(s->x.d(), x, (size_t)n * ndim, s->st));
  GH_CHECK(gh_to_device(s->work.d(), packed_lower, cnt, s->st));
  GH_CHECK(gh_to_device(s->dinv.d(), dinv_in, (size_t)(np / T) * T * T, s->st));
  hipLaunchKernelGGL(unpack_lower_kernel, dim3((unsigned)np), dim3(256), 0, s->st, s->work.d(), (long)n, s->A.d(), (long)np, (long)np);
  GH_HIP(hipGetLastError());
  GH_HIP(hipStreamSynchronize(s->st));
  s->logdet = logdet;
  s->computed = true;
  return GH_OK;
}
extern "C" void gh_chol_release_buffers(gh_chol* s) {
  // frees everything but the handle itself (streams, events); the next compute() re-allocates
  if (!s) return;
  (void)hipSetDevice(s->opts.device);
  if (s->st) (void)hipStreamSynchronize(s->st);
  s->computed = false;
  for (GhBuf* b : {&s->A, &s->dinv, &s->x, &s->yerr, &s->v0, &s->v1, &s->v2, &s->rhs, &s->work, &s->work2, &s->scratch, &s->chain}) b->release();
}
extern "C" void gh_chol_trim(gh_chol* s) {
  // frees the transient N x N / N x M work buffers of predict / grad / get_inverse, keeps the factor
  if (!s) return;
  (void)hipSetDevice(s->opts.device);
  if (s->st) (void)hipStreamSynchronize(s->st);
  for (GhBuf* b : {&s->rhs, &s->work, &s->work2, &s->scratch}) b->release();
}
