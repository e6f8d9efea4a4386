// Probe for the round-2 failure "persistent flag-driven kernels on CU-masked streams: non-positive pivot on every
// second compute()" (DESIGN.md section 4).  A persistent kernel that WAITS for a flag only makes progress if the
// kernel that SETS the flag can start while the waiter is resident.  HIP multiplexes streams onto a few hardware
// queues; two streams on one queue run their kernels one after the other, so a waiter on stream i and its setter on
// stream j dead-lock (until the waiter's time-out) whenever i and j share a queue.  This program measures, for
// streams created the way the solver created them (normal, high-priority, CU-masked with the two complementary
// masks), which ordered pairs (waiter i, setter j) complete and which run into the time-out -- repeated over several
// "computes" (the binding is not fixed: it is re-decided as streams go idle and busy).
//   hipcc --offload-arch=gfx950 -O2 masked_queue_probe.hip -o masked_queue_probe && ./masked_queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void waiter(unsigned* flag, int* result, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  int ok = 0;
  while (true) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 1; break; }
    if (wall_clock64() - t0 > timeout_ticks) break;
    __builtin_amdgcn_s_sleep(4);
  }
  *result = ok;
}
__global__ void setter(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32, reserve = 32;
  std::vector<uint32_t> hi_mask(words, 0u), lo_mask(words, 0u);
  for (int c = 0; c < ncu; ++c) (c < reserve ? lo_mask : hi_mask)[c / 32] |= 1u << (c % 32);
  int plo = 0, phi = 0;
  CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  std::vector<hipStream_t> st;
  std::vector<std::string> name;
  hipStream_t s;
  CK(hipStreamCreate(&s)); st.push_back(s); name.push_back("main(blocking)");
  for (int i = 0; i < 3; ++i) { CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, phi)); st.push_back(s); name.push_back("hiprio" + std::to_string(i)); }
  CK(hipExtStreamCreateWithCUMask(&s, words, hi_mask.data())); st.push_back(s); name.push_back("masked[32..256)");
  for (int i = 0; i < 3; ++i) { CK(hipExtStreamCreateWithCUMask(&s, words, lo_mask.data())); st.push_back(s); name.push_back("masked[0..32)#" + std::to_string(i)); }
  const int S = (int)st.size();
  unsigned* flag; int* res;
  CK(hipMalloc(&flag, 64)); CK(hipMalloc(&res, 64));
  const long long timeout = 5000000;                     // 50 ms at 100 MHz
  for (int rep = 0; rep < 3; ++rep) {
    printf("repetition %d: rows = stream of the WAITER, columns = stream of the SETTER; . = hand-over seen, X = waiter timed out\n     ", rep);
    for (int j = 0; j < S; ++j) printf("%2d ", j);
    printf("\n");
    for (int i = 0; i < S; ++i) {
      printf("  %2d ", i);
      for (int j = 0; j < S; ++j) {
        if (i == j) { printf(" - "); continue; }
        CK(hipMemset(flag, 0, 4)); CK(hipMemset(res, 0, 4));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, st[i], flag, res, timeout);
        hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, st[j], flag);
        CK(hipDeviceSynchronize());
        int ok = 0;
        CK(hipMemcpy(&ok, res, 4, hipMemcpyDeviceToHost));
        printf(" %c ", ok ? '.' : 'X');
      }
      printf("  %s\n", name[i].c_str());
    }
  }
  // the solver's pattern: a memset on one stream, an event, waiter and setter on two masked streams behind that event
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  int bad = 0;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipMemsetAsync(res, 0, 4, st[1]));
    CK(hipMemsetAsync(flag, 0, 4, st[1]));
    CK(hipEventRecord(ev, st[1]));
    CK(hipStreamWaitEvent(st[5], ev, 0)); CK(hipStreamWaitEvent(st[6], ev, 0));
    hipLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, st[5], flag, res, timeout);
    hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, st[6], flag);
    CK(hipDeviceSynchronize());
    int ok = 0; CK(hipMemcpy(&ok, res, 4, hipMemcpyDeviceToHost));
    bad += !ok;
  }
  printf("memset -> event -> (waiter on masked#0, setter on masked#1), 20 repetitions: %d timed out\n", bad);
  return 0;
}
