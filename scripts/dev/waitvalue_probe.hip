// Does hipStreamWaitValue64 release a stream from a value a KERNEL wrote (no event packet in the producer's queue), and how fast?
//   stream A: producer (spins T us, then stores the flag with a system-scope atomic), then a follow-up kernel -- is there a gap?
//   stream B: hipStreamWaitValue64(flag >= v) + consumer kernel -- when does it start relative to the producer's end?
// compared with hipEventRecord / hipStreamWaitEvent.   hipcc --offload-arch=gfx950 -O2 waitvalue_probe.hip -o /tmp/wvp && /tmp/wvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(long long ticks, unsigned long long* flag, unsigned long long v, long long* stamp) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) {
    stamp[0] = wall_clock64();
    if (flag) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void stamp_kernel(long long* stamp, int slot) { if (threadIdx.x == 0) stamp[slot] = wall_clock64(); }
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned long long* flag = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
  if (e != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc((void**)&flag, 8)); printf("(plain hipMalloc instead)\n"); }
  CK(hipMemset(flag, 0, 8));
  long long* stamp = nullptr;
  CK(hipHostMalloc((void**)&stamp, 64 * sizeof(long long)));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const long long T = 3000;   // 30 us of 10-ns ticks
  for (int mode = 0; mode < 3; ++mode) {       // 0: event, 1: wait-value, 2: nothing between (reference for the follow-up gap)
    double g1 = 0, g2 = 0;
    int ok = 0;
    for (int rep = 0; rep < 12; ++rep) {
      const unsigned long long v = 100 + mode * 100 + rep;
      for (int i = 0; i < 8; ++i) stamp[i] = 0;
      hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, T, mode == 1 ? flag : nullptr, v, stamp);
      if (mode == 0) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
      if (mode == 1) { hipError_t w = hipStreamWaitValue64(b, flag, v, hipStreamWaitValueGte, ~0ull); if (w != hipSuccess) { printf("hipStreamWaitValue64 -> %s\n", hipGetErrorString(w)); return 1; } }
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, stamp, 1);      // the producer stream's next kernel
      if (mode != 2) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, stamp, 2);   // the released stream's kernel
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      if (rep >= 2) { g1 += (stamp[1] - stamp[0]) / 100.0; g2 += (stamp[2] - stamp[0]) / 100.0; ++ok; }
    }
    printf("mode %d (%s): producer's end -> its stream's next kernel %.2f us; -> the other stream's kernel %.2f us\n", mode,
           mode == 0 ? "event record + stream wait" : mode == 1 ? "kernel writes flag + hipStreamWaitValue64" : "nothing", g1 / ok, mode == 2 ? 0.0 : g2 / ok);
  }
  return 0;
}
