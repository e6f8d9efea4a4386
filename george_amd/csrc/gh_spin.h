// gh_spin.h -- the ONE implementation of "a spin-wait that gives up": every device-side wait of this library (the chained
// triangular sweeps of gh_chol.hip, the ACA cluster barrier of gh_hodlr.hip) polls through a GhSpin, so that the time-out
// (GH_SPIN_TIMEOUT_TICKS of the 100 MHz wall clock = 2 s) and the abort word's protocol have one definition:
//
//   * the abort word is an `int` in device memory, 0 while all is well; whoever times out sets it (atomicExch) and every
//     other waiter that sees it non-zero gives up as well -- so one stuck producer releases the whole grid and the host
//     reads ONE word to learn that the result is garbage;
//   * the clock and the abort word are looked at every `period` polls only (the poll itself is the hot path).
//
// Forward-progress rule for users (HISTORY.md, "persistent kernels on CU-masked queues"): wait only for workgroups that are
// resident by construction -- members of the same launch with a smaller block index, or of a launch that fits the chip whole.
#pragma once
#include <hip/hip_runtime.h>

#define GH_SPIN_TIMEOUT_TICKS 200000000LL     // wall_clock64() runs at 100 MHz: 2 s

struct GhSpin {
  long long t0;
  unsigned polls;
  int* abort_word;
  __device__ __forceinline__ explicit GhSpin(int* abort_w) : t0(wall_clock64()), polls(0u), abort_word(abort_w) {}
  // call once per unsuccessful poll; false = stop waiting (timed out here, or somebody else raised the abort word)
  __device__ __forceinline__ bool keep_waiting(unsigned period_mask) {
    if ((++polls & period_mask) != 0u) return true;
    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (wall_clock64() - t0 > GH_SPIN_TIMEOUT_TICKS) { atomicExch(abort_word, 1); return false; }
    return true;
  }
};
