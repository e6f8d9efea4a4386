// TEST INFRASTRUCTURE, not product: a stand-in for librccl.so that george_amd's sharded solver loads when
// GEORGE_AMD_RCCL_LIB points at it (george_amd/csrc/gh_mgpu.hip, rccl_load).  It gives the eight entry points
// the solver binds (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv,
// ncclAllReduce, ncclGetErrorString) the semantics of the library's point-to-point interface and CHECKS them:
//
//   * ranks of one communicator set may share a device ("virtual ranks": ncclMockSharedDeviceOk), so the W = 2 .. 8
//     send/receive pattern of the solver can run on the one GPU of the test box;
//   * a send from rank a to rank b is matched with the OLDEST unmatched receive on b from a of the SAME communicator
//     set (the library's rule: per pair and communicator, in issue order); element count and type must agree on the
//     two sides, a stream must belong to the communicator's device, peers must exist and differ from the caller;
//   * a call outside a group, and ncclGroupEnd for the calls inside one, return only when every operation has found
//     its partner (host-side blocking is stricter than the library, whose operations wait on the device; a call
//     pattern that completes here with every rank driven by its own host thread completes there when the ranks
//     issue their operations in one common order) -- or fail after MOCK_RCCL_TIMEOUT_S (default 30) seconds naming
//     the operation that nobody answered;
//   * destroying a communicator with unmatched operations is an error;
//   * the payload moves by an ordinary copy on the RECEIVER's stream behind an event of the sender's stream, and the
//     sender's stream waits for the copy: stream order on both sides as with the library.
//
// mock_rccl_stats() / mock_rccl_last_error() let a test read what happened.  Nothing here is optimised.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {
enum { R_OK = 0, R_HIP = 1, R_SYSTEM = 2, R_INTERNAL = 3, R_BAD_ARG = 4, R_BAD_USAGE = 5 };
enum { T_INT8 = 0, T_UINT8 = 1, T_INT32 = 2, T_UINT32 = 3, T_INT64 = 4, T_UINT64 = 5, T_HALF = 6, T_FLOAT = 7, T_DOUBLE = 8 };
enum { K_SEND = 0, K_RECV = 1, K_ALLREDUCE = 2 };

size_t type_size(int t) {
  switch (t) {
    case T_INT8: case T_UINT8: return 1;
    case T_HALF: return 2;
    case T_INT32: case T_UINT32: case T_FLOAT: return 4;
    case T_INT64: case T_UINT64: case T_DOUBLE: return 8;
    default: return 0;
  }
}

struct Set;
struct Comm { Set* set; int rank; int dev; bool alive; };
struct Op {
  int kind = K_SEND;
  const void* src = nullptr;
  void* dst = nullptr;
  size_t count = 0;
  int dtype = 0, redop = 0, peer = -1;
  Comm* comm = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t posted = nullptr, done = nullptr;
  int state = 0;                 // 0 waiting for the partner, 1 matched, 2 failed
  std::string err;
  long long seq = 0;
};
struct Set {
  int id = 0, n = 0, alive = 0;
  std::vector<Comm*> comms;
  std::map<std::pair<int, int>, std::deque<Op*>> sends, recvs;        // (source rank, destination rank) -> unmatched, oldest first
  std::vector<std::deque<Op*>> reduces;                               // per rank
  std::vector<hipEvent_t> events;
};

std::mutex g_mu;
std::condition_variable g_cv;
std::vector<Set*> g_sets;
long long g_seq = 0;
long long g_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // sets, matched pairs, payload bytes, all-reduces, errors, pending now, largest group, groups
std::string g_last_error;
std::mutex g_err_mu;               // g_last_error and g_stat[4] (fail() runs with and without g_mu)
thread_local int t_depth = 0;
thread_local std::vector<Op*> t_group;
thread_local std::string t_error;

int fail(int code, const std::string& what) {       // (g_mu held or not: only strings and counters)
  t_error = what;
  {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_last_error = what;
    g_stat[4]++;
  }
  if (getenv("MOCK_RCCL_VERBOSE")) fprintf(stderr, "mock_rccl: %s\n", what.c_str());
  return code;
}
std::string describe(const Op* o) {
  char b[256];
  const char* k = o->kind == K_SEND ? "send to" : o->kind == K_RECV ? "receive from" : "all-reduce with";
  snprintf(b, sizeof b, "communicator set %d rank %d: %s rank %d, %zu elements of type %d (operation #%lld)", o->comm->set->id, o->comm->rank, k,
           o->peer, o->count, o->dtype, o->seq);
  return b;
}
double timeout_s() {
  const char* e = getenv("MOCK_RCCL_TIMEOUT_S");
  return e ? atof(e) : 30.0;
}
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
hipEvent_t new_event(Set* s, int dev) {
  DeviceGuard g(dev);
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  s->events.push_back(e);
  return e;
}

// g_mu held.  Pair the heads of sends / receives of (a -> b) while both exist.
void match_pair(Set* s, int a, int b) {
  auto& qs = s->sends[{a, b}];
  auto& qr = s->recvs[{a, b}];
  while (!qs.empty() && !qr.empty()) {
    Op* sd = qs.front(); qs.pop_front();
    Op* rv = qr.front(); qr.pop_front();
    g_stat[5] -= 2;
    if (sd->count != rv->count || sd->dtype != rv->dtype) {
      char b2[512];
      snprintf(b2, sizeof b2, "mismatched pair: [%s] met [%s]", describe(sd).c_str(), describe(rv).c_str());
      sd->err = rv->err = b2;
      sd->state = rv->state = 2;
      continue;
    }
    const size_t bytes = sd->count * type_size(sd->dtype);
    hipError_t e = hipSuccess;
    {
      DeviceGuard g(rv->comm->dev);
      e = hipStreamWaitEvent(rv->st, sd->posted, 0);
      // fault injection (tests/test_gpu_mgpu_mock_rccl.py): MOCK_RCCL_STALL=<pair index>:<seconds> holds the receiver's
      // stream for that long in front of the copy of the <pair index>-th matched pair -- a transfer that hangs on the
      // device for a while, as two crossed communicators would, but ends by itself
      {
        static const char* const stall = getenv("MOCK_RCCL_STALL");
        static long long stall_at = -1;
        static double stall_s = 0.0;
        if (stall && stall_at < 0) { stall_at = atoll(stall); const char* c = strchr(stall, ':'); stall_s = c ? atof(c + 1) : 3.0; }
        if (stall && g_stat[1] == stall_at) {
          double* secs = new double(stall_s);
          (void)hipLaunchHostFunc(rv->st, [](void* p) {
            std::this_thread::sleep_for(std::chrono::duration<double>(*(double*)p));
            delete (double*)p;
          }, secs);
        }
      }
      if (e == hipSuccess && bytes) {
        if (sd->comm->dev == rv->comm->dev) e = hipMemcpyAsync(rv->dst, sd->src, bytes, hipMemcpyDeviceToDevice, rv->st);
        else e = hipMemcpyPeerAsync(rv->dst, rv->comm->dev, sd->src, sd->comm->dev, bytes, rv->st);
      }
      hipEvent_t done = new_event(s, rv->comm->dev);
      if (e == hipSuccess && done) e = hipEventRecord(done, rv->st);
      if (!done) e = hipErrorOutOfMemory;
      sd->done = rv->done = done;
    }
    if (e != hipSuccess) {
      sd->err = rv->err = std::string("copy of a matched pair failed: ") + hipGetErrorString(e);
      sd->state = rv->state = 2;
      continue;
    }
    sd->state = rv->state = 1;
    g_stat[1]++;
    g_stat[2] += (long long)bytes;
  }
}
// g_mu held.  Every rank of the set has an all-reduce at the head of its queue: do it (host staged, synchronous).
void match_reduce(Set* s) {
  for (;;) {
    for (int r = 0; r < s->n; ++r) if (s->reduces[r].empty()) return;
    std::vector<Op*> ops(s->n);
    for (int r = 0; r < s->n; ++r) { ops[r] = s->reduces[r].front(); s->reduces[r].pop_front(); }
    g_stat[5] -= s->n;
    bool same = true;
    for (int r = 1; r < s->n; ++r) same = same && ops[r]->count == ops[0]->count && ops[r]->dtype == ops[0]->dtype && ops[r]->redop == ops[0]->redop;
    std::string err;
    if (!same) err = "all-reduce with different count / type / operation across ranks: [" + describe(ops[0]) + "] ...";
    else if (ops[0]->redop != 0 || (ops[0]->dtype != T_DOUBLE && ops[0]->dtype != T_FLOAT && ops[0]->dtype != T_INT32 && ops[0]->dtype != T_INT64))
      err = "the stand-in reduces sums of double / float / int32 / int64 only";
    if (err.empty()) {
      const size_t cnt = ops[0]->count, ts = type_size(ops[0]->dtype);
      std::vector<char> in(cnt * ts), acc(cnt * ts, 0);
      for (int r = 0; r < s->n && err.empty(); ++r) {
        DeviceGuard g(ops[r]->comm->dev);
        hipError_t e = hipStreamSynchronize(ops[r]->st);
        if (e == hipSuccess) e = hipMemcpy(in.data(), ops[r]->src, cnt * ts, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { err = std::string("all-reduce staging failed: ") + hipGetErrorString(e); break; }
        for (size_t i = 0; i < cnt; ++i) {
          switch (ops[0]->dtype) {
            case T_DOUBLE: ((double*)acc.data())[i] += ((double*)in.data())[i]; break;
            case T_FLOAT: ((float*)acc.data())[i] += ((float*)in.data())[i]; break;
            case T_INT32: ((int32_t*)acc.data())[i] += ((int32_t*)in.data())[i]; break;
            default: ((int64_t*)acc.data())[i] += ((int64_t*)in.data())[i]; break;
          }
        }
      }
      for (int r = 0; r < s->n && err.empty(); ++r) {
        DeviceGuard g(ops[r]->comm->dev);
        hipError_t e = hipMemcpy(ops[r]->dst, acc.data(), cnt * ts, hipMemcpyHostToDevice);
        if (e != hipSuccess) err = std::string("all-reduce staging failed: ") + hipGetErrorString(e);
      }
    }
    for (Op* o : ops) { o->state = err.empty() ? 1 : 2; o->err = err; }
    if (err.empty()) g_stat[3]++;
  }
}

int check_args(int kind, const void* buf, size_t count, int dtype, int peer, Comm* c, hipStream_t st) {
  if (!c || !c->alive) return fail(R_BAD_ARG, "operation on a communicator that does not exist (destroyed?)");
  if (type_size(dtype) == 0) return fail(R_BAD_ARG, "unknown element type");
  if (kind != K_ALLREDUCE && (peer < 0 || peer >= c->set->n)) return fail(R_BAD_ARG, "peer rank outside the communicator");
  if (kind != K_ALLREDUCE && peer == c->rank) return fail(R_BAD_USAGE, "send / receive to the calling rank itself");
  if (count && !buf) return fail(R_BAD_ARG, "null buffer with a positive count");
  int dev = -1;
  if (st != nullptr && hipStreamGetDevice(st, &dev) == hipSuccess && dev != c->dev) {
    char b[160];
    snprintf(b, sizeof b, "stream of device %d handed to a communicator of device %d (set %d rank %d)", dev, c->dev, c->set->id, c->rank);
    return fail(R_BAD_ARG, b);
  }
  return R_OK;
}

// the calling thread's operations `ops` (already checked): publish them all, then wait until each found its partner
int post_and_wait(std::vector<Op*>& ops) {
  int rc = R_OK;
  {
    std::unique_lock<std::mutex> lk(g_mu);
    for (Op* o : ops) {
      o->seq = ++g_seq;
      Set* s = o->comm->set;
      if (o->kind != K_ALLREDUCE) {
        o->posted = new_event(s, o->comm->dev);
        DeviceGuard g(o->comm->dev);
        if (!o->posted || hipEventRecord(o->posted, o->st) != hipSuccess) { o->state = 2; o->err = "could not record an event on the caller's stream"; continue; }
      }
      g_stat[5]++;
      if (o->kind == K_SEND) { s->sends[{o->comm->rank, o->peer}].push_back(o); match_pair(s, o->comm->rank, o->peer); }
      else if (o->kind == K_RECV) { s->recvs[{o->peer, o->comm->rank}].push_back(o); match_pair(s, o->peer, o->comm->rank); }
      else { s->reduces[o->comm->rank].push_back(o); match_reduce(s); }
    }
    g_cv.notify_all();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    for (Op* o : ops) {
      while (o->state == 0) {
        if (g_cv.wait_until(lk, deadline) == std::cv_status::timeout && o->state == 0) {
          Set* s = o->comm->set;
          auto drop = [&](std::deque<Op*>& q) { q.erase(std::remove(q.begin(), q.end(), o), q.end()); };
          if (o->kind == K_SEND) drop(s->sends[{o->comm->rank, o->peer}]);
          else if (o->kind == K_RECV) drop(s->recvs[{o->peer, o->comm->rank}]);
          else drop(s->reduces[o->comm->rank]);
          g_stat[5]--;
          o->state = 2;
          o->err = "never answered by the peer: " + describe(o);
        }
      }
      if (o->state == 2) rc = fail(o->err.find("never answered") == 0 ? R_INTERNAL : R_BAD_ARG, o->err);
    }
  }
  for (Op* o : ops) {
    if (o->state == 1 && o->kind == K_SEND && o->done) {                 // the sender's buffer is free once the copy has run
      DeviceGuard g(o->comm->dev);
      if (hipStreamWaitEvent(o->st, o->done, 0) != hipSuccess) rc = fail(R_HIP, "hipStreamWaitEvent on the sender's stream failed");
    }
    delete o;
  }
  ops.clear();
  return rc;
}

int submit(Op* o) {
  if (t_depth > 0) { t_group.push_back(o); return R_OK; }
  std::vector<Op*> one{o};
  return post_and_wait(one);
}
}  // namespace

extern "C" {
typedef Comm* ncclComm_t;

int ncclMockSharedDeviceOk() { return 1; }

int ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1) return fail(R_BAD_ARG, "ncclCommInitAll: bad arguments");
  int have = 0;
  (void)hipGetDeviceCount(&have);
  for (int i = 0; i < ndev; ++i)
    if (devlist && (devlist[i] < 0 || devlist[i] >= have)) return fail(R_BAD_ARG, "ncclCommInitAll: no such device");
  std::lock_guard<std::mutex> lk(g_mu);
  Set* s = new Set;
  s->id = (int)g_sets.size();
  s->n = s->alive = ndev;
  s->reduces.resize(ndev);
  for (int i = 0; i < ndev; ++i) {
    Comm* c = new Comm{s, i, devlist ? devlist[i] : i, true};
    s->comms.push_back(c);
    comms[i] = c;
  }
  g_sets.push_back(s);
  g_stat[0]++;
  return R_OK;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return R_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!c->alive) return fail(R_BAD_ARG, "communicator destroyed twice");
  Set* s = c->set;
  int rc = R_OK;
  auto mine = [&](const std::deque<Op*>& q) { for (Op* o : q) if (o->comm == c) return true; return false; };
  for (auto& kv : s->sends) if (mine(kv.second)) rc = fail(R_BAD_USAGE, "communicator destroyed with an unmatched send");
  for (auto& kv : s->recvs) if (mine(kv.second)) rc = fail(R_BAD_USAGE, "communicator destroyed with an unmatched receive");
  if (!s->reduces[c->rank].empty()) rc = fail(R_BAD_USAGE, "communicator destroyed with an unmatched all-reduce");
  c->alive = false;
  if (--s->alive == 0) {
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    s->events.clear();
  }
  return rc;
}

int ncclGroupStart() { ++t_depth; return R_OK; }

int ncclGroupEnd() {
  if (t_depth <= 0) return fail(R_BAD_USAGE, "ncclGroupEnd without ncclGroupStart");
  if (--t_depth > 0) return R_OK;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_stat[7]++;
    g_stat[6] = std::max<long long>(g_stat[6], (long long)t_group.size());
  }
  std::vector<Op*> ops;
  ops.swap(t_group);
  return ops.empty() ? R_OK : post_and_wait(ops);
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t st) {
  if (int rc = check_args(K_SEND, buf, count, dtype, peer, c, st)) return rc;
  Op* o = new Op;
  o->kind = K_SEND; o->src = buf; o->count = count; o->dtype = dtype; o->peer = peer; o->comm = c; o->st = st;
  return submit(o);
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t st) {
  if (int rc = check_args(K_RECV, buf, count, dtype, peer, c, st)) return rc;
  Op* o = new Op;
  o->kind = K_RECV; o->dst = buf; o->count = count; o->dtype = dtype; o->peer = peer; o->comm = c; o->st = st;
  return submit(o);
}

int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int redop, ncclComm_t c, hipStream_t st) {
  if (int rc = check_args(K_ALLREDUCE, sendbuf, count, dtype, -1, c, st)) return rc;
  if (count && !recvbuf) return fail(R_BAD_ARG, "null buffer with a positive count");
  Op* o = new Op;
  o->kind = K_ALLREDUCE; o->src = sendbuf; o->dst = recvbuf; o->count = count; o->dtype = dtype; o->redop = redop; o->peer = -1; o->comm = c; o->st = st;
  return submit(o);
}

const char* ncclGetErrorString(int code) {
  if (code != R_OK && !t_error.empty()) return t_error.c_str();
  switch (code) {
    case R_OK: return "no error";
    case R_HIP: return "unhandled hip error";
    case R_SYSTEM: return "unhandled system error";
    case R_INTERNAL: return "internal error";
    case R_BAD_ARG: return "invalid argument";
    case R_BAD_USAGE: return "invalid usage";
    default: return "unknown result code";
  }
}

// out[0..7]: communicator sets made, matched send/receive pairs, payload bytes, all-reduces, errors of any kind,
// operations waiting for a partner right now, operations in the largest group, groups closed
void mock_rccl_stats(long long* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::lock_guard<std::mutex> lk2(g_err_mu);
  for (int i = 0; i < 8; ++i) out[i] = g_stat[i];
}
int mock_rccl_last_error(char* out, int cap) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (out && cap > 0) { strncpy(out, g_last_error.c_str(), (size_t)cap - 1); out[cap - 1] = 0; }
  return (int)g_last_error.size();
}
}
