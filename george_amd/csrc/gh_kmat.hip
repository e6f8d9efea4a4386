// gh_kmat.hip -- kernel-matrix build on MI355X (gfx950).
//
// Replaces the serial double loops of src/george/kernel_interface.cpp:47-157
// (one virtual-call chain per pair) by tiled HIP kernels: a 64x64 output tile
// per 256-thread workgroup, the tile's row and column points staged once in LDS,
// every lane walking the same postfix program (gh_eval.h), results written as
// 16-byte (2 x f64) stores, a wavefront covering two full 512-byte row segments.
//
// Roofline: the plain value build is HBM-write bound on paper (8 B/element) but
// one fp64 exp() per element costs ~40 fp64 VALU ops, so the measured bound is
// the fp64 VALU rate; see DESIGN.md.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <math.h>
#include "gh_common.h"
#include <mutex>

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void gh_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
extern "C" const char* gh_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* gh_version(void) { return "george_amd 0.1 (gfx950)"; }
extern "C" int gh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

__global__ void gh_prime_kernel() {}
void gh_prime_device(int device) {
  static std::mutex mu;
  static bool done[64] = {false};
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= 64 || done[device]) return;
  done[device] = true;
  if (getenv("GEORGE_AMD_NO_NULL_PRIME")) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (hipSetDevice(device) == hipSuccess) {
    hipLaunchKernelGGL(gh_prime_kernel, dim3(1), dim3(64), 0, (hipStream_t)nullptr);
    (void)hipDeviceSynchronize();
  }
  (void)hipGetLastError();
  (void)hipSetDevice(cur);
}

bool gh_is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// ------------------------------------------------------------------ block cache (gh_common.h)
#include <map>
#include <mutex>
namespace {
struct BlockCache {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks[16];      // per device, keyed by capacity
  size_t cached_bytes[16] = {0};
  size_t max_cached = (size_t)128 << 30;             // per device (of 288 GB); beyond it blocks are really freed (gh_set_cache_limit)
};
BlockCache g_cache;
}
void* gh_pool_acquire(size_t bytes, size_t* capacity) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    auto& m = g_cache.free_blocks[dev];
    auto it = m.lower_bound(bytes);
    if (it != m.end() && it->first <= 4 * bytes + 4096) {
      void* p = it->second;
      *capacity = it->first;
      g_cache.cached_bytes[dev] -= it->first;
      m.erase(it);
      return p;
    }
  }
  const size_t cap = (bytes + 255) & ~(size_t)255;
  void* p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) {
    // out of device memory with up to 128 GB of released blocks parked here: give them back and try once more
    (void)hipGetLastError();
    gh_pool_trim();
    if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  *capacity = cap;
  return p;
}
// really free every cached block of the current device
void gh_pool_trim() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    for (auto& kv : g_cache.free_blocks[dev]) drop.push_back(kv.second);
    g_cache.free_blocks[dev].clear();
    g_cache.cached_bytes[dev] = 0;
  }
  for (void* q : drop) (void)hipFree(q);
}
extern "C" void gh_release_caches(int32_t device) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
  gh_pool_trim();
  (void)hipSetDevice(prev);
}
// The block cache parks up to 128 GB of released blocks per device and gives them back to the driver only when one of THIS
// library's allocations fails -- another allocator in the process (torch, RCCL) cannot make it do so.  A process that shares
// the device with one sets a smaller limit here (bytes per device; 0 = cache nothing); blocks above the new limit go now.
extern "C" void gh_set_cache_limit(int64_t bytes) {
  int ndev = 0, prev = 0;
  (void)hipGetDevice(&prev);
  if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
  std::vector<std::pair<int, void*>> drop;
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    g_cache.max_cached = bytes > 0 ? (size_t)bytes : 0;
    for (int d = 0; d < 16; ++d) {
      auto& m = g_cache.free_blocks[d];
      while (g_cache.cached_bytes[d] > g_cache.max_cached && !m.empty()) {
        auto it = std::prev(m.end());                   // largest first
        g_cache.cached_bytes[d] -= it->first;
        drop.emplace_back(d, it->second);
        m.erase(it);
      }
    }
  }
  for (auto& q : drop) { if (q.first < ndev && hipSetDevice(q.first) == hipSuccess) (void)hipFree(q.second); }
  if (!drop.empty()) (void)hipSetDevice(prev);
}
// bytes of released blocks parked in the current device's cache (they will be handed out again before anything is allocated)
size_t gh_pool_parked_bytes() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  std::lock_guard<std::mutex> lk(g_cache.mu);
  return g_cache.cached_bytes[dev];
}
void gh_pool_release(void* p, size_t capacity) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  {
    std::lock_guard<std::mutex> lk(g_cache.mu);
    if (g_cache.cached_bytes[dev] + capacity <= g_cache.max_cached) {
      g_cache.free_blocks[dev].emplace(capacity, p);
      g_cache.cached_bytes[dev] += capacity;
      return;
    }
  }
  (void)hipFree(p);
}

int gh_to_device(double* dst, const double* src, size_t count, hipStream_t st) {
  if (count == 0) return GH_OK;
  GH_HIP(hipMemcpyAsync(dst, src, count * sizeof(double),
                        gh_is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  return GH_OK;
}
int gh_from_device(double* dst, const double* src, size_t count, hipStream_t st) {
  if (count == 0) return GH_OK;
  if (gh_is_device_ptr(dst)) {
    GH_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToDevice, st));
  } else {
    GH_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
  }
  return GH_OK;
}

// ------------------------------------------------- program "compilation"
static int n_own_params(int kt) {
  switch (kt) {
    case GH_K_LINEAR: case GH_K_RATQUAD: case GH_K_COSINE: case GH_K_CONSTANT: case GH_K_POLYNOMIAL: return 1;
    case GH_K_LOCALGAUSS: case GH_K_EXPSINE2: return 2;
    default: return 0;
  }
}
static bool is_stationary(int kt) {
  return kt == GH_K_RATQUAD || kt == GH_K_EXP || kt == GH_K_MATERN52 || kt == GH_K_EXPSQUARED || kt == GH_K_MATERN32;
}

// Reduce the postfix program to  a + b * F(r2)  with one unblocked stationary leaf F over an
// isotropic / axis-aligned metric, if it has that shape (constants fold: Constant = naxes * c).
static void detect_fast_form(gh_kernel* k) {
  memset(&k->fast, 0, sizeof(k->fast));
  if (getenv("GEORGE_AMD_NO_FAST_KERNEL")) return;
  struct Aff { bool ok; double a, b; int leaf; };
  std::vector<Aff> st;
  for (size_t i = 0; i < k->nodes.size(); ++i) {
    const GhNode& nd = k->nodes[i];
    if (nd.op == GH_OP_LEAF) {
      if (nd.ktype == GH_K_CONSTANT) st.push_back({true, nd.naxes * nd.q[0], 0.0, -1});
      else if (nd.mtype >= 0 && nd.mtype <= 1 && !nd.blocked) st.push_back({true, 0.0, 1.0, (int)i});
      else st.push_back({false, 0.0, 0.0, -1});
    } else {
      const Aff r = st.back(); st.pop_back();
      const Aff l = st.back(); st.pop_back();
      Aff o{false, 0.0, 0.0, -1};
      if (l.ok && r.ok) {
        if (nd.op == GH_OP_SUM) {
          if (l.leaf < 0 || r.leaf < 0 || l.leaf == r.leaf) o = {true, l.a + r.a, l.b + r.b, l.leaf >= 0 ? l.leaf : r.leaf};
        } else if (l.leaf < 0) {
          o = {true, l.a * r.a, l.a * r.b, r.leaf};
        } else if (r.leaf < 0) {
          o = {true, l.a * r.a, l.b * r.a, l.leaf};
        }
      }
      st.push_back(o);
    }
  }
  if (st.size() != 1 || !st[0].ok || st[0].leaf < 0) return;
  // only the shapes whose rounding the interpreter reproduces exactly: a single scaling and/or
  // a single offset (two leaves of the same kernel summed would fold b = b1 + b2 differently)
  int nstat = 0;
  for (const GhNode& nd : k->nodes) if (nd.op == GH_OP_LEAF && nd.mtype >= 0) ++nstat;
  if (nstat != 1) return;
  const GhNode& lf = k->nodes[st[0].leaf];
  GhFast& f = k->fast;
  f.ktype = lf.ktype; f.mtype = lf.mtype; f.naxes = lf.naxes;
  for (int i = 0; i < lf.naxes; ++i) { f.axes[i] = lf.axes[i]; f.m[i] = lf.m[lf.mtype == 0 ? 0 : i]; }
  f.a = st[0].a; f.b = st[0].b; f.q0 = lf.q[0];
  f.ok = 1;
}

extern "C" int gh_kernel_create(const gh_knode* in, int n_nodes, gh_kernel** out) {
  if (!in || !out || n_nodes <= 0 || n_nodes > GH_MAX_NODES) {
    gh_set_error("invalid kernel"); return GH_ERR_BAD_ARG;
  }
  gh_kernel* k = new gh_kernel();
  k->nodes.resize(n_nodes);
  // postfix validation with a stack of (param offset, param size, ndim)
  struct Ent { int poff, psize, ndim; };
  std::vector<Ent> st;
  int poff = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const gh_knode& s = in[i];
    GhNode& d = k->nodes[i];
    memset(&d, 0, sizeof(d));
    d.op = s.op;
    if (s.op == GH_OP_LEAF) {
      const int kt = s.kernel_type;
      if (kt < 0 || kt > GH_K_DOTPRODUCT) { delete k; gh_set_error("unrecognized kernel"); return GH_ERR_BAD_ARG; }
      if (s.naxes < 0 || s.naxes > GH_MAX_AXES || s.ndim <= 0 || s.ndim > GH_MAX_NDIM) {
        delete k; gh_set_error("kernel dimension out of range (ndim<=%d, naxes<=%d)", GH_MAX_NDIM, GH_MAX_AXES); return GH_ERR_BAD_ARG;
      }
      d.ktype = kt; d.naxes = s.naxes; d.ndim = s.ndim; d.npar = n_own_params(kt);
      if (s.n_params != d.npar) { delete k; gh_set_error("wrong parameter count for kernel_type %d", kt); return GH_ERR_BAD_ARG; }
      for (int a = 0; a < s.naxes; ++a) {
        if (s.axes[a] < 0 || s.axes[a] >= s.ndim) { delete k; gh_set_error("invalid axis"); return GH_ERR_BAD_ARG; }
        d.axes[a] = s.axes[a];
      }
      for (int p = 0; p < d.npar; ++p) d.p[p] = s.params[p];
      d.cst = s.constant;
      if (is_stationary(kt)) {
        d.mtype = s.metric_type;
        int want = d.mtype == 0 ? 1 : d.mtype == 1 ? s.naxes : d.mtype == 2 ? s.naxes * (s.naxes + 1) / 2 : -1;
        if (want < 0 || s.n_metric != want) { delete k; gh_set_error("unrecognized metric"); return GH_ERR_BAD_ARG; }
        d.nmet = want;
        if (d.mtype == 2) {
          // GeneralMetric::set_parameter, metrics.h:171-181: diagonal slots (0,2,5,9,..) hold exp(-v)
          for (int j = 0; j < want; ++j) d.m[j] = s.metric[j];
          for (int j = 0, step = 2; j < want; j += step, ++step) d.m[j] = exp(-s.metric[j]);
        } else {
          for (int j = 0; j < want; ++j) d.m[j] = exp(-s.metric[j]);     // metrics.h:46-49
        }
        d.blocked = s.blocked ? 1 : 0;
        for (int a = 0; a < s.naxes; ++a) { d.lo[a] = s.min_block[a]; d.hi[a] = s.max_block[a]; }
      } else {
        d.mtype = -1;
      }
      // reparameterisations (kernels/*.yml `reparams`)
      switch (kt) {
        case GH_K_RATQUAD:    d.q[0] = exp(d.p[0]); break;                 // alpha
        case GH_K_CONSTANT:   d.q[0] = exp(d.p[0]); break;                 // constant
        case GH_K_COSINE:     d.q[0] = 2 * M_PI * exp(-d.p[0]); break;     // factor
        case GH_K_EXPSINE2:   d.q[0] = M_PI * exp(-d.p[1]); break;         // factor
        case GH_K_LOCALGAUSS: d.q[0] = 0.5 * exp(-d.p[1]); break;          // inv_2w
        case GH_K_LINEAR:     d.q[0] = exp(-d.p[0]); break;                // inv_gamma2
        case GH_K_POLYNOMIAL: d.q[0] = exp(d.p[0]); break;                 // sigma2
        default: break;
      }
      d.poff = poff; d.psize = d.npar + d.nmet; d.psize1 = 0;
      poff += d.psize;
      st.push_back({d.poff, d.psize, d.ndim});
      if ((int)st.size() > GH_MAX_STACK) { delete k; gh_set_error("kernel expression too deep (max %d)", GH_MAX_STACK); return GH_ERR_BAD_ARG; }
    } else if (s.op == GH_OP_SUM || s.op == GH_OP_PRODUCT) {
      if (st.size() < 2) { delete k; gh_set_error("invalid kernel"); return GH_ERR_BAD_ARG; }
      Ent b = st.back(); st.pop_back();
      Ent a = st.back(); st.pop_back();
      if (a.ndim != b.ndim) { delete k; gh_set_error("dimension mismatch"); return GH_ERR_DIM; }   // parser.h:27
      d.poff = a.poff; d.psize = a.psize + b.psize; d.psize1 = a.psize; d.ndim = a.ndim;
      st.push_back({d.poff, d.psize, d.ndim});
    } else {
      delete k; gh_set_error("unrecognized operator"); return GH_ERR_BAD_ARG;
    }
  }
  if (st.size() != 1) { delete k; gh_set_error("invalid kernel"); return GH_ERR_BAD_ARG; }
  k->ndim = st[0].ndim;
  k->size = st[0].psize;
  if (k->size > GH_MAX_GRAD) { delete k; gh_set_error("too many kernel parameters (max %d)", GH_MAX_GRAD); return GH_ERR_BAD_ARG; }
  detect_fast_form(k);
  *out = k;
  return GH_OK;
}
extern "C" void gh_kernel_destroy(gh_kernel* k) { delete k; }
extern "C" int gh_kernel_ndim(const gh_kernel* k) { return k ? k->ndim : -1; }
extern "C" int gh_kernel_size(const gh_kernel* k) { return k ? k->size : -1; }

int gh_kernel::upload() {
  int dev = 0;
  GH_HIP(hipGetDevice(&dev));
  if (d_nodes && dev == device) return GH_OK;
  if (d_nodes) { (void)hipFree(d_nodes); d_nodes = nullptr; }
  GH_HIP(hipMalloc((void**)&d_nodes, nodes.size() * sizeof(GhNode)));
  GH_HIP(hipMemcpy(d_nodes, nodes.data(), nodes.size() * sizeof(GhNode), hipMemcpyHostToDevice));
  device = dev;
  return GH_OK;
}

// ================================================================= kernels
#define KT 64            // kmat tile edge
// lower-triangular tile enumeration: b -> (ti, tj), tj <= ti
__device__ __forceinline__ void tri_index(long b, int& ti, int& tj) {
  long t = (long)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (t * (t + 1) / 2 > b) --t;
  while ((t + 1) * (t + 2) / 2 <= b) ++t;
  ti = (int)t;
  tj = (int)(b - t * (t + 1) / 2);
}

struct KmatArgs {
  const GhNode* prog; int n_nodes; int ndim;
  const double* x1; long n1;
  const double* x2; long n2;
  const double* yerr;
  double* out; long ldo;
  long rows_p, cols_p;     // padded extent to fill
  long row0, col0;         // global offset of out[0][0] (for the diagonal / ordering tests)
  int sym, lower_only;
  int tiles_n;             // tiles per row of the (padded) output
  GhFast fast;             // affine single-leaf form (fast.ok) -> no interpreter, no node loads
};

// One 64 x 64 tile, any kernel expression, any position (edges, padding, diagonal).
// FAST: the kernel has the affine single-leaf form (a.fast): no interpreter in the instantiation
// (half the registers), passes unrolled so that several exp() chains are in flight per lane.
template <bool FAST>
__device__ __forceinline__ void kmat_generic_tile(const KmatArgs& a, const GhNode* __restrict__ prog, int ti, int tj, double* xr, double* xc) {
  const long r0 = (long)ti * KT, c0 = (long)tj * KT;
  const int nd = a.ndim;
  // stage the tile's points (coalesced: consecutive threads read consecutive doubles)
  for (int t = threadIdx.x; t < KT * nd; t += 256) {
    const long r = r0 + t / nd;
    xr[t] = (r < a.n1) ? a.x1[r * nd + (t % nd)] : 0.0;
    const long c = c0 + t / nd;
    xc[t] = (c < a.n2) ? a.x2[c * nd + (t % nd)] : 0.0;
  }
  __syncthreads();
  const int lc = (threadIdx.x & 31) * 2;          // two adjacent columns per lane -> 16-B stores
  const int lr = threadIdx.x >> 5;                // 8 rows per pass
#pragma unroll FAST ? 4 : 1
  for (int pass = 0; pass < KT / 8; ++pass) {
    const int rr = lr + pass * 8;
    const long r = r0 + rr, c = c0 + lc;
    if (r >= a.rows_p || c >= a.cols_p) continue;
    double v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const long cc = c + e;
      const long gr = a.row0 + r, gc = a.col0 + cc;
      if (r < a.n1 && cc < a.n2) {
        const double* p1 = &xr[rr * nd];
        const double* p2 = &xc[(lc + e) * nd];
        // symmetric build: evaluate k(x_min, x_max) as kernel_interface.cpp:68-74 does
        const bool swap = a.sym && (gr > gc);
        double val = FAST ? gh_fast_value(a.fast, swap ? p2 : p1, swap ? p1 : p2)
                          : gh_eval_value(prog, a.n_nodes, swap ? p2 : p1, swap ? p1 : p2);
        if (a.sym && a.yerr && gr == gc) { const double e2 = a.yerr[r]; val += e2 * e2; }   // basic.py:65
        v[e] = val;
      } else {
        v[e] = (a.sym && gr == gc) ? 1.0 : 0.0;   // identity padding keeps the factor and log-det unchanged
      }
    }
    double* o = a.out + r * a.ldo + c;
    if (c + 1 < a.cols_p && ((((size_t)o) & 15) == 0)) {
      *reinterpret_cast<double2*>(o) = make_double2(v[0], v[1]);
    } else {
      o[0] = v[0];
      if (c + 1 < a.cols_p) o[1] = v[1];
    }
  }
}
__device__ __forceinline__ void kmat_tile_of(const KmatArgs& a, int& ti, int& tj) {
  if (a.lower_only) {
    // enumerate 128x128 tiles of the lower triangle (the granularity of the factorisation);
    // blockIdx.y picks one of its four 64x64 sub-tiles, so diagonal 128-tiles are built in full
    int TI, TJ;
    tri_index(blockIdx.x, TI, TJ);
    ti = TI * 2 + (blockIdx.y >> 1);
    tj = TJ * 2 + (blockIdx.y & 1);
  } else { ti = blockIdx.x / a.tiles_n; tj = blockIdx.x % a.tiles_n; }
}
// (the program as a `const __restrict__` kernel parameter of its own: read-only and no alias of the output, so that the walker's
//  node-field loads -- wave-uniform addresses -- become SCALAR loads.  As a member of KmatArgs they were vector loads, ~77 per
//  element for the hyper.rst kernel, each a dependent round trip in front of a branch: -24 % on that build, profiles/r06/composite_kernel.md)
#ifndef GH_KMAT_INTERP_WAVES
#define GH_KMAT_INTERP_WAVES 4   // <= 128 registers: four wavefronts per SIMD hide the walker's dependent scalar loads (-13 %, profiles/r06/composite_kernel.md)
#endif
template <bool FAST>
__global__ __launch_bounds__(256, FAST ? 2 : GH_KMAT_INTERP_WAVES) void kmat_kernel(KmatArgs a, const GhNode* __restrict__ prog) {
  __shared__ double xr[KT * GH_MAX_NDIM];
  __shared__ double xc[KT * GH_MAX_NDIM];
  int ti, tj;
  kmat_tile_of(a, ti, tj);
  kmat_generic_tile<FAST>(a, prog, ti, tj, xr, xc);
}

// Interior tiles of the stationary kernels a + b F(r^2) on ND <= 3 coordinates used as they come (axes =
// 0 .. ND-1, isotropic or axis-aligned metric): the common case of the dense build (BASELINE configs C1-C5)
// with everything the generic tile pays per ELEMENT taken out -- bounds and padding tests, the ordered
// (x_min, x_max) evaluation (d * d is the same either way), the diagonal test, the run-time axis table and
// dimension.  A lane keeps the coordinates of its two columns in registers and reads one row point per
// pass from LDS.  r^2 and F are formed exactly as gh_fast_value forms them.  A tile that touches the
// matrix edge, the padding or the global diagonal goes through kmat_generic_tile.  Counter-backed
// motivation: the generic FAST kernel issues 63 vector instructions per element, 22 of them the exp
// (profiles/r02/pmc_kmat_VALU_N32768.md): the build was VALU-bound at 38 % of the HBM write roofline.
template <int KTYPE, int ND>
__global__ __launch_bounds__(256) void kmat_interior_kernel(KmatArgs a) {
  __shared__ double xr[KT * GH_MAX_NDIM];
  __shared__ double xc[KT * GH_MAX_NDIM];
  int ti, tj;
  kmat_tile_of(a, ti, tj);
  const long r0 = (long)ti * KT, c0 = (long)tj * KT;
  const long gr0 = a.row0 + r0, gc0 = a.col0 + c0;
  const bool interior = r0 + KT <= a.n1 && c0 + KT <= a.n2 && r0 + KT <= a.rows_p && c0 + KT <= a.cols_p &&
                        (!a.sym || gr0 + KT <= gc0 || gc0 + KT <= gr0) && ((((size_t)a.out) | ((size_t)a.ldo * 8)) & 15) == 0;
  if (!interior) { kmat_generic_tile<true>(a, a.prog, ti, tj, xr, xc); return; }
  for (int t = threadIdx.x; t < KT * ND; t += 256) xr[t] = a.x1[r0 * ND + t];
  __syncthreads();
  const int lc = (threadIdx.x & 31) * 2, lr = threadIdx.x >> 5;
  double xa[ND], xb[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { xa[d] = a.x2[(c0 + lc) * ND + d]; xb[d] = a.x2[(c0 + lc + 1) * ND + d]; }
  const GhFast& f = a.fast;
  double* o = a.out + (r0 + lr) * a.ldo + c0 + lc;
#pragma unroll 4
  for (int pass = 0; pass < KT / 8; ++pass, o += 8 * a.ldo) {
    const double* p1 = &xr[(lr + pass * 8) * ND];
    double v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      double r2 = 0.0;
      if (f.mtype == 0) {
        double sum = 0.0;
#pragma unroll
        for (int d = 0; d < ND; ++d) { const double dd = p1[d] - (e ? xb[d] : xa[d]); sum += dd * dd; }
        r2 = sum * f.m[0];
      } else {
#pragma unroll
        for (int d = 0; d < ND; ++d) { const double dd = p1[d] - (e ? xb[d] : xa[d]); r2 += dd * dd * f.m[d]; }
      }
      double w;
      if (KTYPE == GH_K_EXPSQUARED) w = exp(-0.5 * r2);
      else if (KTYPE == GH_K_MATERN32) { const double r = sqrt(3.0 * r2); w = (1.0 + r) * exp(-r); }
      else if (KTYPE == GH_K_MATERN52) { const double r = sqrt(5.0 * r2); w = (1 + r + 5.0 * r2 / 3.0) * exp(-r); }
      else w = exp(-sqrt(r2));
      {
#pragma clang fp contract(off)          // two roundings, like gh_fast_value and the interpreter's * and + nodes
        const double t = f.b * w;
        v[e] = f.a + t;
      }
    }
    *reinterpret_cast<double2*>(o) = make_double2(v[0], v[1]);
  }
}

int gh_launch_kmat(const gh_kernel* k, const double* x1, int64_t n1, const double* x2, int64_t n2,
                   const double* yerr, double* out, int64_t ldo, int64_t rows_p, int64_t cols_p,
                   int64_t row0, int64_t col0, bool sym, bool lower_only, hipStream_t st) {
  if (rows_p <= 0 || cols_p <= 0) return GH_OK;
  KmatArgs a;
  a.prog = k->d_nodes; a.n_nodes = (int)k->nodes.size(); a.ndim = k->ndim;
  a.x1 = x1; a.n1 = n1; a.x2 = x2; a.n2 = n2; a.yerr = yerr;
  a.out = out; a.ldo = ldo; a.rows_p = rows_p; a.cols_p = cols_p; a.row0 = row0; a.col0 = col0;
  a.sym = sym; a.lower_only = lower_only;
  a.fast = k->fast;
  const long tm = (rows_p + KT - 1) / KT, tn = (cols_p + KT - 1) / KT;
  a.tiles_n = (int)tn;
  const long tm128 = (rows_p + 2 * KT - 1) / (2 * KT);
  long nblk = lower_only ? tm128 * (tm128 + 1) / 2 : tm * tn;
  if (nblk > 0x7fffffffL) { gh_set_error("kernel matrix too large"); return GH_ERR_BAD_ARG; }
  const dim3 grid((unsigned)nblk, lower_only ? 4 : 1), block(256);
  // the specialised interior-tile kernel: a + b F(r^2), coordinates used as they come, ndim <= 3, x1 / x2 full arrays
  static const bool no_interior = getenv("GEORGE_AMD_NO_KMAT_INTERIOR") != nullptr;
  bool plain = a.fast.ok && !no_interior && k->ndim <= 3 && a.fast.naxes == k->ndim && (a.fast.mtype == 0 || a.fast.mtype == 1) &&
               (a.fast.ktype == GH_K_EXPSQUARED || a.fast.ktype == GH_K_MATERN32 || a.fast.ktype == GH_K_MATERN52 || a.fast.ktype == GH_K_EXP);
  for (int i = 0; plain && i < a.fast.naxes; ++i) if (a.fast.axes[i] != i) plain = false;
  if (plain) {
#define GH_KMAT_ND(KTYPE)                                                                                   \
    do {                                                                                                    \
      if (k->ndim == 1)      hipLaunchKernelGGL((kmat_interior_kernel<KTYPE, 1>), grid, block, 0, st, a);   \
      else if (k->ndim == 2) hipLaunchKernelGGL((kmat_interior_kernel<KTYPE, 2>), grid, block, 0, st, a);   \
      else                   hipLaunchKernelGGL((kmat_interior_kernel<KTYPE, 3>), grid, block, 0, st, a);   \
    } while (0)
    switch (a.fast.ktype) {
      case GH_K_EXPSQUARED: GH_KMAT_ND(GH_K_EXPSQUARED); break;
      case GH_K_MATERN32:   GH_KMAT_ND(GH_K_MATERN32); break;
      case GH_K_MATERN52:   GH_KMAT_ND(GH_K_MATERN52); break;
      default:              GH_KMAT_ND(GH_K_EXP); break;
    }
#undef GH_KMAT_ND
  }
  else if (a.fast.ok) hipLaunchKernelGGL(kmat_kernel<true>, grid, block, 0, st, a, (const GhNode*)k->d_nodes);
  else                hipLaunchKernelGGL(kmat_kernel<false>, grid, block, 0, st, a, (const GhNode*)k->d_nodes);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

__global__ void kdiag_kernel(const GhNode* __restrict__ prog, int n_nodes, int nd, const double* x1, const double* x2, long n, double* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = gh_eval_value(prog, n_nodes, x1 + i * nd, x2 + i * nd);
}
int gh_launch_kdiag(const gh_kernel* k, const double* x1, const double* x2, int64_t n, double* out, hipStream_t st) {
  if (n <= 0) return GH_OK;
  hipLaunchKernelGGL(kdiag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     k->d_nodes, (int)k->nodes.size(), k->ndim, x1, x2, (long)n, out);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// full (n1, n2, P) parameter-gradient tensor -- API parity with gradient_general/symmetric
// (kernel_interface.cpp:92-125).  HBM-write bound: 8*P bytes per pair.
__global__ __launch_bounds__(256) void kgrad_kernel(const GhNode* __restrict__ prog, int n_nodes, int nd, int P,
                                                    const uint32_t* which, const double* x1, long n1,
                                                    const double* x2, long n2, int sym, double* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n1 * n2) return;
  const long i = idx / n2, j = idx % n2;
  double g[GH_MAX_GRAD];
  const bool swap = sym && (i > j);
  const double* p1 = x1 + i * nd;
  const double* p2 = x2 + j * nd;
  gh_eval_grad(prog, n_nodes, swap ? p2 : p1, swap ? p1 : p2, g);
  double* o = out + idx * P;
  for (int p = 0; p < P; ++p) o[p] = which[p] ? g[p] : 0.0;
}
int gh_launch_kgrad(const gh_kernel* k, const uint32_t* which_host, const double* x1, int64_t n1,
                    const double* x2, int64_t n2, bool sym, double* out, hipStream_t st) {
  const int P = k->size;
  if (n1 * n2 <= 0 || P == 0) return GH_OK;
  uint32_t* d_which = nullptr;
  GH_HIP(hipMalloc((void**)&d_which, sizeof(uint32_t) * P));
  GH_HIP(hipMemcpyAsync(d_which, which_host, sizeof(uint32_t) * P, hipMemcpyHostToDevice, st));
  const long tot = n1 * n2;
  hipLaunchKernelGGL(kgrad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                     k->d_nodes, (int)k->nodes.size(), k->ndim, P, d_which, x1, (long)n1, x2, (long)n2, sym ? 1 : 0, out);
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(st);
  (void)hipFree(d_which);
  GH_HIP(e);
  return GH_OK;
}

// (n1, n2, ndim) coordinate gradients (kernel_interface.cpp:127-157)
__global__ __launch_bounds__(256) void kxgrad_kernel(const GhNode* __restrict__ prog, int n_nodes, int nd, int which_arg,
                                                     const double* x1, long n1, const double* x2, long n2, double* out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n1 * n2) return;
  const long i = idx / n2, j = idx % n2;
  double g1[GH_MAX_NDIM], g2[GH_MAX_NDIM];
  gh_eval_xgrad(prog, n_nodes, nd, x1 + i * nd, x2 + j * nd, g1, g2);
  double* o = out + idx * nd;
  for (int d = 0; d < nd; ++d) o[d] = which_arg == 1 ? g1[d] : g2[d];
}
int gh_launch_kxgrad(const gh_kernel* k, int which_arg, const double* x1, int64_t n1,
                     const double* x2, int64_t n2, double* out, hipStream_t st) {
  const long tot = n1 * n2;
  if (tot <= 0) return GH_OK;
  hipLaunchKernelGGL(kxgrad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                     k->d_nodes, (int)k->nodes.size(), k->ndim, which_arg, x1, (long)n1, x2, (long)n2, out);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// ---- fused gradient reduction: never materialises the (N, N, P) tensor of gp.py:465
// One 64x64 tile of the LOWER triangle per workgroup; per-thread partial sums over its 16
// elements, wavefront shuffle reduction, LDS cross-wave reduction, one partial row per
// workgroup; a second kernel sums the partial rows in fixed order (deterministic).
#ifndef GH_KGRAD_WAVES
#define GH_KGRAD_WAVES 4
#endif
template <int PMAX>
__global__ __launch_bounds__(256, PMAX <= 16 ? GH_KGRAD_WAVES : 1) void kgrad_reduce_kernel(const GhNode* __restrict__ prog, int n_nodes, int nd, int P,
                                                           const uint32_t* which, const double* x, long n,
                                                           const double* alpha, const double* kinv, long ld,
                                                           double* partial, double* diagA) {
  __shared__ double xr[KT * GH_MAX_NDIM];
  __shared__ double xc[KT * GH_MAX_NDIM];
  __shared__ double red[4][PMAX];
  int ti, tj;
  tri_index(blockIdx.x, ti, tj);
  const long r0 = (long)ti * KT, c0 = (long)tj * KT;
  for (int t = threadIdx.x; t < KT * nd; t += 256) {
    const long r = r0 + t / nd;
    xr[t] = (r < n) ? x[r * nd + (t % nd)] : 0.0;
    const long c = c0 + t / nd;
    xc[t] = (c < n) ? x[c * nd + (t % nd)] : 0.0;
  }
  __syncthreads();
  double acc[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) acc[p] = 0.0;
  const int lc = threadIdx.x & 63;
  const int lr = threadIdx.x >> 6;
#pragma unroll 1
  for (int pass = 0; pass < KT / 4; ++pass) {
    const int rr = lr + pass * 4;
    const long r = r0 + rr, c = c0 + lc;
    if (r < n && c <= r) {
      double g[PMAX];
      // ordered arguments (x_min, x_max) = (x_c, x_r) since c <= r  (kernel_interface.cpp:117-121)
      gh_eval_grad(prog, n_nodes, &xc[lc * nd], &xr[rr * nd], g);
      const double kin = kinv[r * ld + c];
      const double aij = alpha[r] * alpha[c] - kin;
      const double w = (r == c) ? 0.5 * aij : aij;
      if (r == c && diagA) diagA[r] = aij;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) if (p < P) acc[p] += w * g[p];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    double v = acc[p];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][p] = v;
  }
  __syncthreads();
  if (threadIdx.x < PMAX && threadIdx.x < P) {
    const int p = threadIdx.x;
    partial[(long)blockIdx.x * P + p] = which[p] ? (red[0][p] + red[1][p]) + (red[2][p] + red[3][p]) : 0.0;
  }
}
__global__ void kgrad_final_kernel(const double* partial, long nblk, int P, double* grad) {
  // one workgroup per parameter; fixed-order tree -> bitwise reproducible
  __shared__ double s[256];
  const int p = blockIdx.x;
  double v = 0.0;
  for (long b = threadIdx.x; b < nblk; b += 256) v += partial[b * P + p];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) grad[p] = s[0];
}

int gh_launch_kgrad_reduce(const gh_kernel* k, const uint32_t* which_host, const double* x, int64_t n,
                           const double* alpha, const double* kinv, int64_t ld, double* grad_dev,
                           double* diagA, GhBuf& scratch, hipStream_t st) {
  const int P = k->size;
  if (n <= 0) return GH_OK;
  const long tm = (n + KT - 1) / KT;
  const long nblk = tm * (tm + 1) / 2;
  const size_t which_bytes = ((sizeof(uint32_t) * (P > 0 ? P : 1) + 15) / 16) * 16;
  GH_CHECK(scratch.ensure(which_bytes + sizeof(double) * nblk * P));
  uint32_t* d_which = (uint32_t*)scratch.p;
  double* partial = (double*)((char*)scratch.p + which_bytes);
  if (P > 0) GH_HIP(hipMemcpyAsync(d_which, which_host, sizeof(uint32_t) * P, hipMemcpyHostToDevice, st));
  const int nn = (int)k->nodes.size();
#define GH_LAUNCH_RED(PM)                                                                        \
  hipLaunchKernelGGL(kgrad_reduce_kernel<PM>, dim3((unsigned)nblk), dim3(256), 0, st, k->d_nodes, \
                     nn, k->ndim, P, d_which, x, (long)n, alpha, kinv, (long)ld, partial, diagA)
  if (P <= 4) GH_LAUNCH_RED(4);
  else if (P <= 16) GH_LAUNCH_RED(16);
  else GH_LAUNCH_RED(GH_MAX_GRAD);
#undef GH_LAUNCH_RED
  GH_HIP(hipGetLastError());
  if (P > 0) {
    hipLaunchKernelGGL(kgrad_final_kernel, dim3(P), dim3(256), 0, st, partial, nblk, P, grad_dev);
    GH_HIP(hipGetLastError());
  }
  return GH_OK;
}

// ============================================================ C-ABI: evaluator
namespace {
struct Staged {            // host-or-device input staged into device memory
  GhBuf buf;
  const double* ptr = nullptr;
  int stage(const double* src, size_t count, hipStream_t st) {
    if (gh_is_device_ptr(src)) { ptr = src; return GH_OK; }
    GH_CHECK(buf.ensure(count * sizeof(double)));
    GH_CHECK(gh_to_device(buf.d(), src, count, st));
    ptr = buf.d();
    return GH_OK;
  }
};
struct OutBuf {            // device result buffer, copied back if the caller's pointer is host memory
  GhBuf buf;
  double* ptr = nullptr;
  double* user = nullptr;
  int begin(double* out, size_t count) {
    user = out;
    if (gh_is_device_ptr(out)) { ptr = out; return GH_OK; }
    GH_CHECK(buf.ensure(count * sizeof(double)));
    ptr = buf.d();
    return GH_OK;
  }
  int finish(size_t count, hipStream_t st) {
    if (ptr != user) return gh_from_device(user, ptr, count, st);
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
  }
};
}  // namespace

static int check_kernel(gh_kernel* k) {
  if (!k) { gh_set_error("invalid kernel"); return GH_ERR_BAD_ARG; }
  if (gh_device_count() <= 0) { gh_set_error("no HIP device available: the george_amd kernels need an MI355X"); return GH_ERR_HIP; }
  return k->upload();
}

extern "C" int gh_kernel_value_general(gh_kernel* k, const double* x1, int64_t n1, const double* x2, int64_t n2, double* out) {
  GH_CHECK(check_kernel(k));
  if (n1 < 0 || n2 < 0) { gh_set_error("negative size"); return GH_ERR_BAD_ARG; }
  if (n1 == 0 || n2 == 0) return GH_OK;
  Staged a, b; OutBuf o;
  GH_CHECK(a.stage(x1, n1 * k->ndim, 0));
  GH_CHECK(b.stage(x2, n2 * k->ndim, 0));
  GH_CHECK(o.begin(out, n1 * n2));
  GH_CHECK(gh_launch_kmat(k, a.ptr, n1, b.ptr, n2, nullptr, o.ptr, n2, n1, n2, 0, 0, false, false, 0));
  return o.finish(n1 * n2, 0);
}
extern "C" int gh_kernel_value_symmetric(gh_kernel* k, const double* x, int64_t n, double* out) {
  GH_CHECK(check_kernel(k));
  if (n < 0) { gh_set_error("negative size"); return GH_ERR_BAD_ARG; }
  if (n == 0) return GH_OK;
  Staged a; OutBuf o;
  GH_CHECK(a.stage(x, n * k->ndim, 0));
  GH_CHECK(o.begin(out, n * n));
  GH_CHECK(gh_launch_kmat(k, a.ptr, n, a.ptr, n, nullptr, o.ptr, n, n, n, 0, 0, true, false, 0));
  return o.finish(n * n, 0);
}
extern "C" int gh_kernel_value_diagonal(gh_kernel* k, const double* x1, const double* x2, int64_t n, double* out) {
  GH_CHECK(check_kernel(k));
  if (n < 0) { gh_set_error("negative size"); return GH_ERR_BAD_ARG; }
  if (n == 0) return GH_OK;
  Staged a, b; OutBuf o;
  GH_CHECK(a.stage(x1, n * k->ndim, 0));
  GH_CHECK(b.stage(x2, n * k->ndim, 0));
  GH_CHECK(o.begin(out, n));
  GH_CHECK(gh_launch_kdiag(k, a.ptr, b.ptr, n, o.ptr, 0));
  return o.finish(n, 0);
}
static int grad_common(gh_kernel* k, const uint32_t* which, const double* x1, int64_t n1,
                       const double* x2, int64_t n2, bool sym, double* out) {
  GH_CHECK(check_kernel(k));
  if (n1 < 0 || n2 < 0 || !which) { gh_set_error("bad argument"); return GH_ERR_BAD_ARG; }
  const size_t tot = (size_t)n1 * n2 * k->size;
  if (tot == 0) return GH_OK;
  Staged a, b; OutBuf o;
  GH_CHECK(a.stage(x1, n1 * k->ndim, 0));
  GH_CHECK(b.stage(x2, n2 * k->ndim, 0));
  GH_CHECK(o.begin(out, tot));
  GH_CHECK(gh_launch_kgrad(k, which, a.ptr, n1, b.ptr, n2, sym, o.ptr, 0));
  return o.finish(tot, 0);
}
extern "C" int gh_kernel_gradient_general(gh_kernel* k, const uint32_t* which, const double* x1, int64_t n1,
                                          const double* x2, int64_t n2, double* out) {
  return grad_common(k, which, x1, n1, x2, n2, false, out);
}
extern "C" int gh_kernel_gradient_symmetric(gh_kernel* k, const uint32_t* which, const double* x, int64_t n, double* out) {
  return grad_common(k, which, x, n, x, n, true, out);
}
static int xgrad_common(gh_kernel* k, int arg, const double* x1, int64_t n1, const double* x2, int64_t n2, double* out) {
  GH_CHECK(check_kernel(k));
  if (n1 < 0 || n2 < 0) { gh_set_error("negative size"); return GH_ERR_BAD_ARG; }
  const size_t tot = (size_t)n1 * n2 * k->ndim;
  if (tot == 0) return GH_OK;
  Staged a, b; OutBuf o;
  GH_CHECK(a.stage(x1, n1 * k->ndim, 0));
  GH_CHECK(b.stage(x2, n2 * k->ndim, 0));
  GH_CHECK(o.begin(out, tot));
  GH_CHECK(gh_launch_kxgrad(k, arg, a.ptr, n1, b.ptr, n2, o.ptr, 0));
  return o.finish(tot, 0);
}
extern "C" int gh_kernel_x1_gradient_general(gh_kernel* k, const double* x1, int64_t n1, const double* x2, int64_t n2, double* out) {
  return xgrad_common(k, 1, x1, n1, x2, n2, out);
}
extern "C" int gh_kernel_x2_gradient_general(gh_kernel* k, const double* x1, int64_t n1, const double* x2, int64_t n2, double* out) {
  return xgrad_common(k, 2, x1, n1, x2, n2, out);
}

extern "C" int gh_dev_kmat_block(gh_kernel* k, const double* x, int64_t n, int32_t ndim, const double* yerr,
                                 int64_t row0, int64_t nrows, int64_t col0, int64_t ncols,
                                 double* out, int64_t ldo, void* stream) {
  GH_CHECK(check_kernel(k));
  if (ndim != k->ndim) { gh_set_error("dimension mismatch"); return GH_ERR_DIM; }
  // x / yerr are the FULL device arrays of n points; the block's points start at row0 / col0.
  // Rows/columns past n are identity padding (1 on the global diagonal, 0 elsewhere).
  const int64_t vr = std::max<int64_t>(0, std::min<int64_t>(nrows, n - row0));
  const int64_t vc = std::max<int64_t>(0, std::min<int64_t>(ncols, n - col0));
  const int64_t r0 = std::min(row0, n > 0 ? n - 1 : 0), c0 = std::min(col0, n > 0 ? n - 1 : 0);
  return gh_launch_kmat(k, x + r0 * ndim, vr, x + c0 * ndim, vc, yerr ? yerr + r0 : nullptr,
                        out, ldo, nrows, ncols, row0, col0, true, false, (hipStream_t)stream);
}

// y = beta*y + alpha * A x  (trans == 0, A is m x n row-major)  or  alpha * A^T x (trans != 0):
// the O(N^2) steps of the distributed triangular solves.  One wavefront per row (coalesced row
// reads + shuffle reduction) or one lane per column (coalesced across lanes) respectively.
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* A, long lda, long m, long n, const double* x,
                                                     double* y, double alpha, double beta) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  double acc = 0.0;
  for (long c = lane; c < n; c += 64) acc += A[row * lda + c] * x[c];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) y[row] = (beta == 0.0 ? 0.0 : beta * y[row]) + alpha * acc;
}
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* A, long lda, long m, long n, const double* x,
                                                     double* y, double alpha, double beta) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double acc = 0.0;
  for (long r = 0; r < m; ++r) acc += A[r * lda + c] * x[r];
  y[c] = (beta == 0.0 ? 0.0 : beta * y[c]) + alpha * acc;
}
extern "C" int gh_dev_gemv(const double* a, int64_t lda, int64_t m, int64_t n, int32_t trans,
                           const double* x, double* y, double alpha, double beta, void* stream) {
  if (m <= 0 || n <= 0) return GH_OK;
  if (!trans) hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, (long)lda, (long)m, (long)n, x, y, alpha, beta);
  else hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (long)lda, (long)m, (long)n, x, y, alpha, beta);
  GH_HIP(hipGetLastError());
  return GH_OK;
}
