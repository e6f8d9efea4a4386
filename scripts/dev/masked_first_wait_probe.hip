// Does the FIRST operation submitted to a freshly created stream honour hipStreamWaitEvent?  Stream A runs a kernel that
// spins ~3 ms and then writes DONE; event E is recorded behind it; stream B's first-ever operations are
// hipStreamWaitEvent(B, E) and a kernel that reads DONE.  If B's kernel sees DONE == 0 the wait was not honoured.
// Tried for normal, high-priority and CU-masked streams, first use and second use, and with a hipMemsetAsync as the
// operation in front of the event on A (the fused-panel driver's pattern: memset of the flag words, event, waiters).
//   hipcc --offload-arch=gfx950 -O2 masked_first_wait_probe.hip -o masked_first_wait_probe && ./masked_first_wait_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void slow_then_set(unsigned* done, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void read_done(const unsigned* done, unsigned* seen) { *seen = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
  for (int c = 0; c < ncu; ++c) (c < 32 ? lo : hi)[c / 32] |= 1u << (c % 32);
  int plo = 0, phi = 0; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  unsigned *done, *seen; CK(hipMalloc(&done, 64)); CK(hipMalloc(&seen, 64));
  hipEvent_t E; CK(hipEventCreateWithFlags(&E, hipEventDisableTiming));
  const char* kinds[] = {"normal", "high-priority", "CU-masked [0,32)", "CU-masked [32,ncu)"};
  for (int akind = 0; akind < 4; ++akind)
    for (int bkind = 0; bkind < 4; ++bkind) {
      int fails[2] = {0, 0};
      for (int trial = 0; trial < 4; ++trial) {
        hipStream_t A, B;
        auto mk = [&](hipStream_t* s, int kind) -> hipError_t {
          if (kind == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
          if (kind == 1) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, phi);
          return hipExtStreamCreateWithCUMask(s, words, kind == 2 ? lo.data() : hi.data());
        };
        CK(mk(&A, akind)); CK(mk(&B, bkind));
        for (int use = 0; use < 2; ++use) {
          CK(hipMemset(done, 0, 4)); CK(hipMemset(seen, 0xff, 4)); CK(hipDeviceSynchronize());
          hipLaunchKernelGGL(slow_then_set, dim3(1), dim3(64), 0, A, done, 300000LL);      // ~3 ms
          CK(hipEventRecord(E, A));
          CK(hipStreamWaitEvent(B, E, 0));
          hipLaunchKernelGGL(read_done, dim3(1), dim3(64), 0, B, done, seen);
          CK(hipDeviceSynchronize());
          unsigned v = 7; CK(hipMemcpy(&v, seen, 4, hipMemcpyDeviceToHost));
          if (v != 1u) ++fails[use];
        }
        CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
      }
      printf("producer %-20s consumer %-20s : wait NOT honoured on first use %d/4, on second use %d/4\n", kinds[akind], kinds[bkind], fails[0], fails[1]);
    }
  return 0;
}
