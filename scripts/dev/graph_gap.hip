// Probe: what does a dependent launch cost on this chip / runtime, and does a captured hipGraph make it cheaper?
//   (a) a chain of L dependent small kernels on ONE stream          (what a panel-chain link pays three times)
//   (b) the same chain captured once and replayed as a hipGraph
//   (c) a ping-pong between TWO streams through events              (a cross-stream hand-over per link)
//   (d) the ping-pong captured (fork/join edges become graph edges)
// Kernel bodies: `work` spins for about `ticks` s_memtime ticks on `wg` workgroups, so gap = link - body.
//   hipcc --offload-arch=gfx950 -O2 graph_gap.hip -o graph_gap && ./graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void work(unsigned long long* sink, long long ticks) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] += 1;           // dependent on the previous launch
}

static int chain(hipStream_t st, unsigned long long* d, int L, int wg, long long ticks) {
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL(work, dim3(wg), dim3(256), 0, st, d, ticks);
  return 0;
}
static int pingpong(hipStream_t a, hipStream_t b, std::vector<hipEvent_t>& ev, unsigned long long* d, int L, int wg, long long ticks) {
  for (int i = 0; i < L; ++i) {
    hipStream_t s = (i & 1) ? b : a, o = (i & 1) ? a : b;
    hipLaunchKernelGGL(work, dim3(wg), dim3(256), 0, s, d, ticks);
    if (hipEventRecord(ev[i], s) != hipSuccess) return 1;
    if (hipStreamWaitEvent(o, ev[i], 0) != hipSuccess) return 1;
  }
  return 0;
}

int main() {
  const int L = 512;
  unsigned long long* d = nullptr;
  CK(hipMalloc(&d, 64));
  CK(hipMemset(d, 0, 64));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(L + 1);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int wgs[3] = {1, 64, 512};
  const long long tk[2] = {0, 1000};                                // body ~0 and ~10 us (s_memtime: 100 MHz)
  for (int wi = 0; wi < 3; ++wi) for (int ti = 0; ti < 2; ++ti) {
    const int wg = wgs[wi]; const long long ticks = tk[ti];
    float ms[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {                             // (a)
      CK(hipEventRecord(t0, sa)); chain(sa, d, L, wg, ticks); CK(hipEventRecord(t1, sa)); CK(hipStreamSynchronize(sa));
      CK(hipEventElapsedTime(&ms[0], t0, t1));
    }
    hipGraph_t g; hipGraphExec_t ge;                                // (b)
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    chain(sa, d, L, wg, ticks);
    CK(hipStreamEndCapture(sa, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(t0, sa)); CK(hipGraphLaunch(ge, sa)); CK(hipEventRecord(t1, sa)); CK(hipStreamSynchronize(sa));
      CK(hipEventElapsedTime(&ms[1], t0, t1));
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    for (int rep = 0; rep < 3; ++rep) {                             // (c)
      CK(hipEventRecord(t0, sa));
      CK(hipEventRecord(ev[L], sa)); CK(hipStreamWaitEvent(sb, ev[L], 0));
      if (pingpong(sa, sb, ev, d, L, wg, ticks)) return 1;
      CK(hipEventRecord(ev[L], sb)); CK(hipStreamWaitEvent(sa, ev[L], 0));
      CK(hipEventRecord(t1, sa)); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      CK(hipEventElapsedTime(&ms[2], t0, t1));
    }
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal)); // (d)
    CK(hipEventRecord(ev[L], sa)); CK(hipStreamWaitEvent(sb, ev[L], 0));
    if (pingpong(sa, sb, ev, d, L, wg, ticks)) return 1;
    CK(hipEventRecord(ev[L], sb)); CK(hipStreamWaitEvent(sa, ev[L], 0));
    CK(hipStreamEndCapture(sa, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(t0, sa)); CK(hipGraphLaunch(ge, sa)); CK(hipEventRecord(t1, sa)); CK(hipStreamSynchronize(sa));
      CK(hipEventElapsedTime(&ms[3], t0, t1));
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    printf("wg=%4d body~%2lld us | per link: stream %6.2f us  graph %6.2f us | two-stream events %6.2f us  captured %6.2f us\n",
           wg, ticks / 100, ms[0] * 1e3 / L, ms[1] * 1e3 / L, ms[2] * 1e3 / L, ms[3] * 1e3 / L);
  }
  unsigned long long h = 0;
  CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("launches counted: %llu\n", h);
  return 0;
}
