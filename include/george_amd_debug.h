/*
 * george_amd_debug.h -- NOT part of the drop-in boundary (that is george_amd.h): validation switches, stream-placement
 * probes and the on-device micro-benchmarks that pin the roofline denominators.  Exported by libgeorge_amd.so for
 * tests/, bench.py and scripts/; a george maintainer binds nothing from here.
 */
#ifndef GEORGE_AMD_DEBUG_H_
#define GEORGE_AMD_DEBUG_H_

#include "george_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fp64 MFMA / HBM micro-benchmarks used to pin the roofline denominators
 * (SURVEY.md 8d): returns TFLOP/s of a v_mfma_f64_16x16x4_f64-only kernel and
 * GB/s of a 16-B/lane copy. */
int gh_microbench_mfma_f64(double* tflops_out);
/* validation / A-B switch for every GEMM: 0 = plain-VALU kernel (same semantics, cross-checks the
 * MFMA lane maps on the device), anything else = v_mfma_f64_16x16x4 with LDS-DMA operand staging
 * (default); returns the previous setting. */
int gh_debug_set_mfma(int mode);
/* the k-major x k-major GEMM kernel with its slab loop software-pipelined by half a slab (gemm_f64_mfma_dma_sp): 1 on, 0 off,
 * -1 the build's default; returns the previous mode.  Same bits either way (tests/test_gpu_gemm.py). */
int gh_debug_set_gemm_sp(int mode);
/* tile order of the GEMM launches without k clipping: -1 the library's rule (grouped when the column operand exceeds 128 MiB), 0
 * row-major, 1 row groups of 8 walked column-major; returns the previous mode.  Same bits (a tile's arithmetic does not depend on
 * when it runs).  For the traffic A/B of profiles/r06/syrk_traffic_ab.md. */
int gh_debug_set_gemm_grouped(int mode);
/* 1 (default): with look-ahead, a compute()'s inputs and kernel-matrix build are enqueued on the chain stream (whose first panel is
 * the first thing that needs them); 0: on the main stream with a cross-stream hand-over, as until round 6.  Returns the previous setting. */
int gh_debug_set_build_on_chain(int on);
/* 1 (default): with the panel width left to the solver (gh_chol_opts.nb == 0) the outer panels are 2048 columns wide while the
 * trailing matrix behind them has more than 25 600 columns and 1024 after; 0: 1024 throughout; n > 1: the bound is n columns.
 * Returns the previous setting.  Same bits whatever the widths (the update adds the same k in the same order). */
int gh_debug_set_adaptive_panels(int on);
/* HODLR passes that serve two levels at once (round 5): bit 0 = the narrow solve (update of level l + chunk products of the
 * next level in one pass over the rows, "sum + core product" in one launch, symmetric leaf product), bit 1 = the factorisation
 * sweep's update of level l + reduce of the next level in one pass over U; -1: the default (both); returns the previous mask.
 */
int gh_debug_set_hodlr_passes(int mask);
/* 1: HODLR leaves of 129 .. 256 rows through the in-place pivoted Gauss-Jordan (what leaves of more than 256 rows take) instead of
 * the 2 x 2 blocked Cholesky; returns the previous setting (validation arm) */
int gh_debug_set_hodlr_leaf_gj(int on);
/* 1 (default): the ACA of tree levels whose blocks have at most 128 rows and columns runs with one wavefront per node
 * (hodlr_aca_wave_kernel); 0: every level with one workgroup per node.  Same ranks and factors, bit for bit; returns the
 * previous setting. */
int gh_debug_set_hodlr_wave_aca(int on);
/* workgroups the cooperative ACA launch of the top tree levels may use (32 .. 256, default 256: one per CU); returns the previous
 * value.  Same draws; the cluster sizes change the reduction trees, so results agree to rounding, not bit for bit. */
int gh_debug_set_hodlr_coop_wgs(int n);
/* the clusters below the first clustered level of that launch get 1 / div of the workgroups the even-load rule deals them (never
 * fewer than two): 2 (default; < 1 restores it), 1 = the even-load rule of rounds 3-5.  Returns the previous setting.  Same bits. */
int gh_debug_set_hodlr_coop_lower(int div);
/* 1 (default): the one-workgroup ACA launch takes each level's nodes longest first, by the durations the nodes reported in the
 * handle's previous compute(); 0: in tree order.  Returns the previous setting.  Same bits (a node's arithmetic does not depend on
 * when it runs). */
int gh_debug_set_hodlr_lpt(int on);
/* 1 (default): where the factorisation's leaf product is one pass of the 128-row-leaf kernel, the compaction of the ACA scratch
 * writes the level-major copy V only and that product reads it and writes the row-major U for the first time; 0: the compaction
 * writes both and the product works on U in place (rounds 3-5).  Returns the previous setting.  Same bits. */
int gh_debug_set_hodlr_u_from_v(int on);
/* 1 (default): for the levels with many small nodes (>= 32 nodes, cores of <= 32 rows) "sum of the chunk partials + core inverse +
 * core product" is ONE launch with a workgroup per node (hodlr_core_kernel); 0: the three launches of rounds 2-5.  Returns the
 * previous setting.  Same bits. */
int gh_debug_set_hodlr_core_fused(int on);
/* 1 (default): a level that could be clustered but is left with one workgroup per node by the launch's budget (level 5 of C4) is
 * appended to the cooperative launch as one-workgroup segments; 0: it goes to the one-workgroup launch.  Returns the previous setting. */
int gh_debug_set_hodlr_coop_singles(int on);
/* 1 (default): 128-row leaves of kernels with the a + b F(r^2) fast form are evaluated inside the leaf factorisation kernel
 * (potf2_kinv_kernel<true>); 0: a build launch writes them first.  Same bits.  Returns the previous setting. */
int gh_debug_set_hodlr_leaf_fused(int on);
/* which of a dense handle's streams run concurrently (HIP maps streams onto few hardware queues):
 * out[i * 6 + j], i < j, n >= 36: milliseconds for two 300-us spin kernels launched together on
 * streams i and j (0 caller's null stream, 1 main, 2 chain, 3 rows-below, 4 near, 5 CU-masked); -1 = absent */
int gh_debug_stream_overlap(gh_chol* s, double* out, int n);
/* out[i * 6 + j], i != j: ms until a one-workgroup kernel on stream j completes when it is launched right after a grid of
 * 2^18 workgroups (~2 ms) on stream i: small = the two queues dispatch independently (same stream numbering) */
int gh_debug_stream_dispatch(gh_chol* s, double* out, int n);
int gh_microbench_hbm_copy(double* gbps_out);
/* instruction-rate suite (n >= 16): out[0..2] = v_mfma_f64_16x16x4 TFLOP/s, cycles/instr, GHz at
 * 1 wave/SIMD; out[3..5] same at 2 waves/SIMD; out[6] TFLOP/s at 4 waves/SIMD; out[7..9] v_fma_f64
 * TFLOP/s, cycles/instr, GHz at 4 waves/SIMD; out[10] TFLOP/s at 8 waves/SIMD; out[11..12]
 * v_mfma_f64_4x4x4 TFLOP/s and cycles/instr. */
int gh_microbench_suite(double* out, int n);
/* THE fp64 matrix-pipe ceiling: a bare v_mfma_f64_16x16x4_f64 issue loop in inline assembly (accumulators pinned to VGPRs),
 * 64x more (short) workgroups than the chip has slots, resident wavefronts per SIMD pinned by an LDS request.  gh_gemm.hip
 * says why the suite above is NOT a ceiling: its builtin-based loop compiles to 128 accumulator moves per 8 matrix instructions.  out[0..2] = TFLOP/s at 1 / 2 / 4
 * wavefronts per SIMD, out[3] = the best, out[4..6] = milliseconds; n >= 8. */
int gh_microbench_mfma_f64_ceiling(double* out, int n);

/* Timing aids of the sharded dense solver (gh_mgpu_opts.flags, beside the public GH_MGPU_PLAIN_CYCLIC = 1 and GH_MGPU_ONE_COMM = 8):
 * what profiles/r04/scale_model.md was measured with. */
enum {
  GH_MGPU_CHAIN_ONLY   = 2,  /* skip every trailing update except block column k+1 and the bulk gather -- what is left is the
                                critical chain; the results mean nothing (compute() still returns GH_OK), NOT_PD is not raised */
  GH_MGPU_TRACE        = 4   /* record (rank, step, phase, ms, flops or bytes) of every phase: gh_mgpu_get_trace.  With
                                GH_MGPU_COPY the compute phases of all ranks run one at a time and to completion, so
                                that virtual devices sharing one GPU give the durations of a rank alone on its GPU  */
};
/* GH_MGPU_TRACE: rows of 5 doubles (rank, step k, phase, milliseconds, flops or bytes) of the last compute();
 * phases: 0 potrf, 1 column TRSM (+ pack), 2 update of block column k+1, 3 the rest of the trailing update,
 * 4 L_kk transfer, 5 row-panel transfer, 6 panel tile k+1 sent ahead, 7 column-panel gather, 8 kernel-matrix build
 * (step -1).  *n_rows = rows
 * recorded (out may be NULL to ask for the count). */
int  gh_mgpu_get_trace(const gh_mgpu* h, double* out, int64_t max_rows, int64_t* n_rows);

#ifdef __cplusplus
}
#endif
#endif  /* GEORGE_AMD_DEBUG_H_ */
