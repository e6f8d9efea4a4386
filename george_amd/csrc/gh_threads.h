// gh_threads.h -- host-thread plumbing shared by the two several-devices-in-one-process solvers
// (gh_mgpu.hip: dense, gh_hodlr.hip: HODLR sub-tree split): the "ranks" are threads of the caller's
// process, one per device.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

// A barrier of `n` host threads that gives up when the handle's abort flag goes up (a rank that
// failed must not leave the others waiting for ever).
struct HostBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 1, waiting = 0;
  unsigned gen = 0;
  std::atomic<int>* abort = nullptr;
  bool wait() {
    if (n <= 1) return !abort->load();
    std::unique_lock<std::mutex> lk(m);
    const unsigned g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return !abort->load(); }
    while (gen == g) {
      cv.wait_for(lk, std::chrono::milliseconds(20));
      if (abort->load()) { cv.notify_all(); return false; }
    }
    return !abort->load();
  }
  // before a new run of the threads: an aborted run leaves the arrival count of the barrier it died in behind
  void reset() { std::lock_guard<std::mutex> lk(m); waiting = 0; }
};
