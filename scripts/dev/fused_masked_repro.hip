// Reconstruction of the round-2 arm that failed ("server and in-panel workers of the fused panel on CU-masked streams:
// non-positive pivot in the second panel on every second compute()"), to see whether the KERNELS and their flag
// protocol fail on masked queues when the driver joins every stream before a panel's flag words are cleared again.
// The kernels are the retired ones (scripts/dev/arms/panel_fused.hip.inc) on top of the product's gh_gemm.hip and
// gh_potf2_body.h; the driver below is new (the original no longer exists).  A right-looking factorisation of an
// np x np SPD matrix in panels of 1024 columns: fused panel -> trailing update (one GEMM) -> next panel, repeated as
// several "computes" on the same streams and flag buffer, compared bit for bit with gh_dev_potrf_block of the library.
//   arms: 0 = normal streams; 1 = server + in-panel workers on streams masked to CUs [0, 32), trailing update on a
//         stream masked to [32, ncu), rows-below workers on a normal high-priority stream;
//         2 = as 1 with the rows-below workers joined one step later (before the block-column update instead of right
//             after the panel)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../george_amd/csrc fused_masked_repro.hip \
//         -L../../george_amd/csrc -lgeorge_amd -Wl,-rpath,'$ORIGIN/../../george_amd/csrc' -o fused_masked_repro
#include "gh_potf2_body.h"
#include "gh_gemm.hip"
#include "arms/panel_fused.hip.inc"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int np = argc > 1 ? atoi(argv[1]) : 4096, NB = 1024, reps = 6;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
  for (int c = 0; c < ncu; ++c) (c < 32 ? lo : hi)[c / 32] |= 1u << (c % 32);
  std::vector<double> Ah((size_t)np * np);
  for (int i = 0; i < np; ++i)
    for (int j = 0; j < np; ++j) {
      const double d = (i - j) / 40.0;
      Ah[(size_t)i * np + j] = 0.7 * exp(-0.5 * d * d) + (i == j ? 0.05 : 0.0);
    }
  double *A, *Aref, *dinv, *dinv_ref;
  long long* info; unsigned* flags;
  const size_t bytes = (size_t)np * np * 8, dbytes = (size_t)(np / 128) * 128 * 128 * 8;
  CK(hipMalloc(&A, bytes)); CK(hipMalloc(&Aref, bytes)); CK(hipMalloc(&dinv, dbytes)); CK(hipMalloc(&dinv_ref, dbytes));
  CK(hipMalloc(&info, 8)); CK(hipMalloc(&flags, 64 * 4));
  // reference: the library's launch chain
  CK(hipMemcpy(Aref, Ah.data(), bytes, hipMemcpyHostToDevice));
  CK(hipMemset(info, 0, 8));
  for (int k0 = 0; k0 < np; k0 += NB) {
    if (gh_dev_potrf_block(Aref + (size_t)k0 * np + k0, np, NB, dinv_ref + (size_t)(k0 / 128) * 128 * 128, (int64_t*)info, k0, nullptr)) return 2;
    const int m = np - (k0 + NB);
    if (m > 0) {
      if (gh_dev_trsm_right(Aref + (size_t)k0 * np + k0, np, dinv_ref + (size_t)(k0 / 128) * 128 * 128, Aref + (size_t)(k0 + NB) * np + k0, np, m, NB, nullptr)) return 2;
      if (gh_dev_gemm_nt(Aref + (size_t)(k0 + NB) * np + k0 + NB, np, Aref + (size_t)(k0 + NB) * np + k0, np, Aref + (size_t)(k0 + NB) * np + k0, np, m, m, NB, 1, nullptr)) return 2;
    }
  }
  CK(hipDeviceSynchronize());
  std::vector<double> Lref((size_t)np * np), L((size_t)np * np);
  CK(hipMemcpy(Lref.data(), Aref, bytes, hipMemcpyDeviceToHost));
  int plo = 0, phi = 0;
  CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  double* Asnap; CK(hipMalloc(&Asnap, bytes));
  double* Ainit; CK(hipMalloc(&Ainit, bytes)); CK(hipMemcpy(Ainit, Ah.data(), bytes, hipMemcpyHostToDevice));
  const bool h2d = argc > 3 && atoi(argv[3]) != 0;
  const bool dump = argc > 4 && atoi(argv[4]) != 0;
  std::vector<double> S((size_t)np * np);
  const int narm = argc > 2 ? atoi(argv[2]) : 8;
  for (int round = 0; round < narm; ++round) {
    const int forced = argc > 5 ? atoi(argv[5]) : -1;
    const int arm = forced >= 0 ? forced : (round == 0 ? 0 : (round == narm - 1 ? 2 : 1));
    hipStream_t sc, si, sw, sm;
    if (arm == 0) {
      CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, phi)); CK(hipStreamCreateWithPriority(&si, hipStreamNonBlocking, phi));
      CK(hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, phi)); CK(hipStreamCreate(&sm));
    } else if (arm == 3) {                                                // si with a DIFFERENT mask (CUs [0, 31)): cannot share sc's queue
      std::vector<uint32_t> lo2 = lo; lo2[0] &= ~(1u << 31);
      CK(hipExtStreamCreateWithCUMask(&sc, words, lo.data())); CK(hipExtStreamCreateWithCUMask(&si, words, lo2.data()));
      CK(hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, phi)); CK(hipExtStreamCreateWithCUMask(&sm, words, hi.data()));
    } else if (arm == 4) {                                                // rows-below workers masked AWAY from the reserved CUs
      CK(hipExtStreamCreateWithCUMask(&sc, words, lo.data())); CK(hipExtStreamCreateWithCUMask(&si, words, lo.data()));
      CK(hipExtStreamCreateWithCUMask(&sw, words, hi.data())); CK(hipExtStreamCreateWithCUMask(&sm, words, hi.data()));
    } else {
      CK(hipExtStreamCreateWithCUMask(&sc, words, lo.data())); CK(hipExtStreamCreateWithCUMask(&si, words, lo.data()));
      CK(hipStreamCreateWithPriority(&sw, hipStreamNonBlocking, phi)); CK(hipExtStreamCreateWithCUMask(&sm, words, hi.data()));
    }
    hipEvent_t e0, e1, e2, e3, ew;
    for (hipEvent_t* e : {&e0, &e1, &e2, &e3, &ew}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int rep = 0; rep < reps; ++rep) {
      const auto t_start = std::chrono::steady_clock::now();
      unsigned fl0[64]; memset(fl0, 0xee, sizeof(fl0)); double ms_p0 = -1.0;
      if (h2d) CK(hipMemcpyAsync(A, Ah.data(), bytes, hipMemcpyHostToDevice, sc));      // (pageable host memory, as the first version of this harness did)
      else CK(hipMemcpyAsync(A, Ainit, bytes, hipMemcpyDeviceToDevice, sc));
      CK(hipMemsetAsync(info, 0, 8, sc));
      CK(hipMemsetAsync(dinv, 0, dbytes, sc));
      // depth-1 look-ahead, as factor_lookahead_deep: chain (sc + worker streams): panel(j) -> U(j, j+1) -> panel(j+1) ...;
      // sm: W(j) = update of the columns from j+2 on, beside panel(j+1)
      bool have_w = false;
      for (int k0 = 0; k0 < np; k0 += NB) {
        CK(hipMemsetAsync(flags, 0, 64 * 4, sc));
        CK(hipEventRecord(e0, sc));
        CK(hipStreamWaitEvent(si, e0, 0)); CK(hipStreamWaitEvent(sw, e0, 0));
        if (gh_launch_panel_fused(A, np, np, k0, NB, dinv + (size_t)(k0 / 128) * 128 * 128, info, flags, sc, si, sw)) return 3;
        CK(hipEventRecord(e1, si)); CK(hipStreamWaitEvent(sc, e1, 0));
        CK(hipEventRecord(e2, sw));
        if (arm != 2) CK(hipStreamWaitEvent(sc, e2, 0));
        const int k1 = k0 + NB, m = np - k1;
        if (k0 == 0 && dump) {                                          // diagnosis: state of the flag words when panel 0 is over
          CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(si)); CK(hipStreamSynchronize(sw));
          ms_p0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
          CK(hipMemcpy(fl0, flags, 256, hipMemcpyDeviceToHost));
        }
        if (m <= 0) break;
        CK(hipEventRecord(e3, sc));                                     // panel k0 complete (arm 2: as far as sc knows)
        CK(hipStreamWaitEvent(sm, e3, 0)); CK(hipStreamWaitEvent(sm, e2, 0));
        if (have_w) CK(hipStreamWaitEvent(sc, ew, 0));                  // block column k1 was inside W of the previous panel
        if (arm == 2) CK(hipStreamWaitEvent(sc, e2, 0));                // (the update READS the rows below: it has to wait; the memset of the NEXT panel is what arm 2 leaves unordered -- see below)
        GhGemm g{};
        g.C = A + (size_t)k1 * np + k1; g.ldc = np; g.A = A + (size_t)k1 * np + k0; g.lda = np; g.B = g.A; g.ldb = np;
        g.M = m; g.N = NB; g.K = NB; g.alpha = -1.0; g.beta = 1.0; g.a_km = true; g.b_km = true; g.lower = false;
        if (gh_launch_gemm(g, sc)) return 4;                            // U(j, j+1) on the chain
        if (k0 == 0) CK(hipMemcpyAsync(Asnap, A, bytes, hipMemcpyDeviceToDevice, sc));      // state in front of panel 1 (W(0) may be running: only block column 1 is looked at)
        const int k2 = k1 + NB, m2 = np - k2;
        if (m2 > 0) {
          GhGemm w{};
          w.C = A + (size_t)k2 * np + k2; w.ldc = np; w.A = A + (size_t)k2 * np + k0; w.lda = np; w.B = w.A; w.ldb = np;
          w.M = m2; w.N = m2; w.K = NB; w.alpha = -1.0; w.beta = 1.0; w.a_km = true; w.b_km = true; w.lower = true;
          if (gh_launch_gemm(w, sm)) return 4;                          // W(j) on sm, beside panel(j+1)
        }
        CK(hipEventRecord(ew, sm));
        have_w = true;
      }
      CK(hipDeviceSynchronize());
      long long inf = 0; unsigned fl[64];
      CK(hipMemcpy(&inf, info, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(fl, flags, 256, hipMemcpyDeviceToHost));
      CK(hipMemcpy(L.data(), A, bytes, hipMemcpyDeviceToHost));
      double worst = 0.0; long bad = 0;
      for (int i = 0; i < np; ++i)
        for (int j = 0; j <= i; ++j) {
          const double a = L[(size_t)i * np + j], b = Lref[(size_t)i * np + j];
          if (!(a == b)) { ++bad; const double d = fabs(a - b); if (d > worst || d != d) worst = d; }
        }
      printf("arm %d (round %d) compute %d: info %lld abort flag %u  lower-triangle entries that differ from the launch chain: %ld (max |diff| %.3e)\n",
             arm, round, rep, inf, fl[PF_ABORT], bad, worst);
      if (bad || rep == 0) {
        printf("    compute took %.1f ms", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
        if (dump) {
          printf("; panel 0 over after %.1f ms with flags DIAG", ms_p0);
          for (int q = 0; q < 8; ++q) printf(" %u", fl0[PF_DIAG + q]);
          printf(" DINV");
          for (int q = 0; q < 8; ++q) printf(" %u", fl0[PF_DINV + q]);
          printf(" ABORT %u", fl0[PF_ABORT]);
        }
        printf("\n");
      }
      if (bad) {
        // where was it wrong in front of panel 1?  panel 0 (columns [0, NB)) against the reference factor; the diagonal block of
        // block column 1 against reference: A[k1:k1+NB, k1:k1+NB] - X X^T recomputed on the host from the REFERENCE panel 0
        CK(hipMemcpy(S.data(), Asnap, bytes, hipMemcpyDeviceToHost));
        long badp = 0;
        for (int i = 0; i < np; ++i) for (int j = 0; j < NB && j <= i; ++j) if (!(S[(size_t)i * np + j] == Lref[(size_t)i * np + j])) ++badp;
        printf("    wrong row blocks of panel 0 (block: wrong column blocks | U = still the ORIGINAL matrix there):");
        for (int rb = 0; rb < np / 128; ++rb) {
          int any = 0; char buf[64]; int at = 0;
          for (int cb = 0; cb < NB / 128 && cb <= rb; ++cb) {
            long w = 0, orig = 0;
            for (int i = rb * 128; i < rb * 128 + 128; ++i)
              for (int j = cb * 128; j < cb * 128 + 128 && j <= i; ++j) {
                if (!(S[(size_t)i * np + j] == Lref[(size_t)i * np + j])) ++w;
                if (S[(size_t)i * np + j] == Ah[(size_t)i * np + j]) ++orig;
              }
            if (w) { any = 1; at += snprintf(buf + at, sizeof(buf) - at, "%d%s ", cb, orig > 8000 ? "U" : ""); }
          }
          if (any) printf(" [%d: %s]", rb, buf);
        }
        printf("\n");
        long badd = 0; double wd = 0.0;
        for (int i = NB; i < 2 * NB && i < np; ++i)
          for (int j = NB; j <= i; ++j) {
            double v = Ah[(size_t)i * np + j];
            for (int k = 0; k < NB; ++k) v -= Lref[(size_t)i * np + k] * Lref[(size_t)j * np + k];
            const double d = fabs(v - S[(size_t)i * np + j]);
            if (d > 1e-9) { ++badd; if (d > wd) wd = d; }
          }
        printf("    snapshot in front of panel 1: panel-0 entries that differ from the reference %ld; updated diagonal block of column 1: %ld entries off by > 1e-9 (max %.3e); A[k1][k1] = %.6e\n",
               badp, badd, wd, S[(size_t)NB * np + NB]);
      }
      fflush(stdout);
    }
    for (hipStream_t s_ : {sc, si, sw, sm}) CK(hipStreamDestroy(s_));
  }
  return 0;
}
