// gh_eval.h -- device-side evaluator of a flattened george kernel expression.
//
// The reference evaluates k(x_i, x_j) by a virtual-call chain over a heap tree
// (include/george/kernels.h:21-163 + 13 leaf classes; metrics.h).  Here the tree
// is a postfix array of POD nodes in HBM; every lane of a wavefront walks the
// SAME program, so all control flow is wave-uniform (scalar branches, scalar
// loads of node fields) and only the point coordinates differ per lane.
//
// Formulas follow the reference's YAML kernel specs (kernels/*.yml) and
// metrics.h; file:line citations are next to each case.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include "../../include/george_amd.h"

#ifndef GH_HD
#define GH_HD __host__ __device__ __forceinline__
#endif

// Device-side ("compiled") node: raw parameters plus the reparameterisations the
// reference caches in update_reparams() (templates/kernels.h:430-440).
struct GhNode {
  int op, ktype, mtype, naxes;
  int blocked, npar, nmet, ndim;
  int poff;      // offset of this subtree's first parameter in the full vector
  int psize;     // number of parameters in this subtree
  int psize1;    // operators: size of the left subtree
  int pad_;
  int axes[GH_MAX_AXES];
  double p[GH_MAX_PARAMS];   // raw own parameters
  double q[GH_MAX_PARAMS];   // derived: alpha / factor / inv_gamma2 / sigma2 / constant / inv_2w
  double cst;                // 'order'
  double m[GH_MAX_METRIC];   // metric vector_ AFTER the exp(-v) transform (metrics.h:46-49,171-181)
  double lo[GH_MAX_AXES], hi[GH_MAX_AXES];
};

// ---------------------------------------------------------------- metrics
// r2 and (optionally) d r2 / d metric-params.   metrics.h:76-91 (iso), :108-130 (axis), :183-234 (general)
template <bool GRAD>
GH_HD double gh_metric(const GhNode& nd, const double* x1, const double* x2, double* g) {
  if (nd.mtype == 0) {
    double s = 0.0;
    for (int i = 0; i < nd.naxes; ++i) {
      const int a = nd.axes[i];
      const double d = x1[a] - x2[a];
      s += d * d;
    }
    const double r2 = s * nd.m[0];
    if (GRAD) g[0] = -r2;
    return r2;
  } else if (nd.mtype == 1) {
    double r2 = 0.0;
    for (int i = 0; i < nd.naxes; ++i) {
      const int a = nd.axes[i];
      double d = x1[a] - x2[a];
      d = d * d * nd.m[i];
      r2 += d;
      if (GRAD) g[i] = -d;
    }
    return r2;
  } else {
    const int n = nd.naxes;
    double r[GH_MAX_AXES], lir[GH_MAX_AXES];
    for (int i = 0; i < n; ++i) { const int a = nd.axes[i]; r[i] = x1[a] - x2[a]; }
    // _custom_forward_sub, metrics.h:144-151 (diagonal slots hold 1/L_ii)
    for (int i = 0, k = 0; i < n; ++i) {
      for (int j = 0; j < i; ++j, ++k) r[i] -= nd.m[k] * r[j];
      r[i] *= nd.m[k++];
    }
    double r2 = 0.0;
    for (int i = 0; i < n; ++i) { lir[i] = r[i]; r2 += r[i] * r[i]; }
    if (GRAD) {
      // _custom_backward_sub, metrics.h:153-164
      const int k0 = (n + 1) * n / 2;
      for (int i = n - 1; i >= 0; --i) {
        int k = k0 - n + i;
        for (int j = n - 1; j > i; --j) { r[i] -= nd.m[k] * r[j]; k -= j; }
        r[i] *= nd.m[k];
      }
      // metrics.h:219-226 -- restated as written, including exp(vector_[k]) on the diagonal slots
      for (int i = 0, k = 0; i < n; ++i) {
        g[k] = -2 * r[i] * lir[i] * exp(nd.m[k]);
        k++;
        for (int j = i + 1; j < n; ++j) g[k++] = -2 * r[j] * lir[i];
      }
    }
    return r2;
  }
}

// metric.x1_gradient into g[ndim] (entries off the active axes untouched).  metrics.h:93-99,132-138,236-250
GH_HD void gh_metric_x1(const GhNode& nd, const double* x1, const double* x2, double* g) {
  if (nd.mtype == 0) {
    for (int i = 0; i < nd.naxes; ++i) { const int a = nd.axes[i]; g[a] = nd.m[0] * (x1[a] - x2[a]); }
  } else if (nd.mtype == 1) {
    for (int i = 0; i < nd.naxes; ++i) { const int a = nd.axes[i]; g[a] = nd.m[i] * (x1[a] - x2[a]); }
  } else {
    const int n = nd.naxes;
    double r[GH_MAX_AXES];
    for (int i = 0; i < n; ++i) { const int a = nd.axes[i]; r[i] = x1[a] - x2[a]; }
    for (int i = 0, k = 0; i < n; ++i) {
      for (int j = 0; j < i; ++j, ++k) r[i] -= nd.m[k] * r[j];
      r[i] *= nd.m[k++];
    }
    for (int i = 0; i < n; ++i) g[nd.axes[i]] = r[i];
  }
}

// templates/kernels.h:262-270
GH_HD bool gh_out_of_block(const GhNode& nd, const double* x1, const double* x2) {
  if (!nd.blocked) return false;
  for (int i = 0; i < nd.naxes; ++i) {
    const int a = nd.axes[i];
    if (x1[a] < nd.lo[i] || x1[a] > nd.hi[i] || x2[a] < nd.lo[i] || x2[a] > nd.hi[i]) return true;
  }
  return false;
}

// ------------------------------------------------------ stationary radial part
// value k(r2); if GRAD also dk/dr2 (rg) and dk/d(own param) (pg, RationalQuadratic only)
template <bool GRAD>
GH_HD double gh_radial(const GhNode& nd, double r2, double& rg, double& pg) {
  switch (nd.ktype) {
    case GH_K_EXPSQUARED: {               // kernels/ExpSquared.yml:12,15
      const double e = exp(-0.5 * r2);
      if (GRAD) rg = -0.5 * e;
      return e;
    }
    case GH_K_MATERN32: {                 // kernels/Matern32.yml:13-20
      const double r = sqrt(3.0 * r2);
      const double e = exp(-r);
      if (GRAD) rg = -3.0 * 0.5 * e;
      return (1.0 + r) * e;
    }
    case GH_K_MATERN52: {                 // kernels/Matern52.yml:13-20
      const double r = sqrt(5.0 * r2);
      const double e = exp(-r);
      if (GRAD) rg = -5 * (1 + r) * e / 6.0;
      return (1 + r + 5.0 * r2 / 3.0) * e;
    }
    case GH_K_EXP: {                      // kernels/Exp.yml
      const double r = sqrt(r2);
      const double e = exp(-r);
      if (GRAD) rg = (r2 < DBL_EPSILON) ? 0.0 : -0.5 * e / r;
      return e;
    }
    case GH_K_RATQUAD: {                  // kernels/RationalQuadratic.yml
      const double alpha = nd.q[0];
      const double t1 = 1.0 + 0.5 * r2 / alpha;
      const double v = pow(t1, -alpha);
      if (GRAD) {
        rg = -0.5 * pow(t1, -alpha - 1);
        const double t2 = 2.0 * alpha * t1;
        pg = alpha * v * (r2 / t2 - log(t1));
      }
      return v;
    }
    default:
      return 0.0;
  }
}

// -------------------------------------------------- non-stationary per-axis part
// value f(a,b); if GRAD also param grads pg[0..1]; if XG also df/da, df/db
template <bool GRAD, bool XG>
GH_HD double gh_axis(const GhNode& nd, double a, double b, double* pg, double& da, double& db) {
  switch (nd.ktype) {
    case GH_K_CONSTANT: {                 // kernels/Constant.yml:17-32
      if (GRAD) pg[0] = nd.q[0];
      if (XG) { da = 0.0; db = 0.0; }
      return nd.q[0];
    }
    case GH_K_EMPTY: {
      if (XG) { da = 0.0; db = 0.0; }
      return 0.0;
    }
    case GH_K_DOTPRODUCT: {               // kernels/DotProduct.yml
      if (XG) { da = b; db = a; }
      return a * b;
    }
    case GH_K_COSINE: {                   // kernels/Cosine.yml
      const double f = nd.q[0];
      if (GRAD) { const double r = f * (a - b); pg[0] = r * sin(r); }
      if (XG) { da = -f * sin(f * (a - b)); db = f * sin(f * (a - b)); }
      return cos((a - b) * f);
    }
    case GH_K_EXPSINE2: {                 // kernels/ExpSine2.yml
      const double gamma = nd.p[0], f = nd.q[0];
      const double arg = (a - b) * f;
      const double s = sin(arg);
      const double A = exp(-gamma * s * s);
      if (GRAD) {
        const double c = cos(arg);
        pg[0] = -(s * s) * A;
        pg[1] = 2 * gamma * arg * c * s * A;
      }
      if (XG) {
        const double d = a - b;
        const double t = A * f * gamma * sin(2.0 * f * d);
        da = -t; db = t;
      }
      return A;
    }
    case GH_K_LOCALGAUSS: {               // kernels/LocalGaussian.yml
      const double loc = nd.p[0], inv_2w = nd.q[0];
      const double d1 = a - loc, d2 = b - loc;
      const double arg = (d1 * d1 + d2 * d2) * inv_2w;
      const double e = exp(-arg);
      if (GRAD) { pg[0] = 2 * e * inv_2w * (d1 + d2); pg[1] = e * arg; }
      if (XG) { da = -2.0 * e * d1 * inv_2w; db = -2.0 * e * d2 * inv_2w; }
      return e;
    }
    case GH_K_LINEAR: {                   // kernels/Linear.yml
      const double order = nd.cst, ig = nd.q[0];
      if (order == 0.0) {
        if (GRAD) pg[0] = -ig;
        if (XG) { da = 0.0; db = 0.0; }
        return ig;
      }
      const double v = pow(a * b, order) * ig;
      if (GRAD) pg[0] = -v;
      if (XG) {
        const double pm = pow(a * b, order - 1.0) * ig;
        da = b * order * pm; db = a * order * pm;
      }
      return v;
    }
    case GH_K_POLYNOMIAL: {               // kernels/Polynomial.yml
      const double order = nd.cst, s2 = nd.q[0];
      if (order == 0.0) {
        if (GRAD) pg[0] = 0.0;
        if (XG) { da = 0.0; db = 0.0; }
        return 1.0;
      }
      if (GRAD || XG) {
        const double pm = pow(a * b + s2, order - 1.0);
        if (GRAD) pg[0] = s2 * pm * order;
        if (XG) { da = b * order * pm; db = a * order * pm; }
      }
      return pow(a * b + s2, order);
    }
    default:
      if (XG) { da = 0.0; db = 0.0; }
      return 0.0;
  }
}

// ------------------------------------------------------------------ leaves
GH_HD double gh_leaf_value(const GhNode& nd, const double* x1, const double* x2) {
  if (nd.mtype >= 0) {                    // stationary: templates/kernels.h:260-285
    if (gh_out_of_block(nd, x1, x2)) return 0.0;
    const double r2 = gh_metric<false>(nd, x1, x2, nullptr);
    double rg, pg;
    return gh_radial<false>(nd, r2, rg, pg);
  }
  double v = 0.0, da, db;                 // non-stationary: templates/kernels.h:536-555 (sum over axes)
  for (int i = 0; i < nd.naxes; ++i) {
    const int a = nd.axes[i];
    v += gh_axis<false, false>(nd, x1[a], x2[a], nullptr, da, db);
  }
  return v;
}

// value + gradient wrt this leaf's parameters into g[0 .. psize)
GH_HD double gh_leaf_grad(const GhNode& nd, const double* x1, const double* x2, double* g) {
  if (nd.mtype >= 0) {                    // templates/kernels.h:312-365
    if (gh_out_of_block(nd, x1, x2)) {
      for (int i = 0; i < nd.psize; ++i) g[i] = 0.0;
      return 0.0;
    }
    const double r2 = gh_metric<true>(nd, x1, x2, g + nd.npar);
    double rg = 0.0, pg = 0.0;
    const double v = gh_radial<true>(nd, r2, rg, pg);
    if (nd.npar > 0) g[0] = pg;
    for (int i = 0; i < nd.nmet; ++i) g[nd.npar + i] *= rg;
    return v;
  }
  double v = 0.0, da, db;                 // templates/kernels.h:604-630
  double pg[2];
  for (int p = 0; p < nd.npar; ++p) g[p] = 0.0;
  for (int i = 0; i < nd.naxes; ++i) {
    const int a = nd.axes[i];
    v += gh_axis<true, false>(nd, x1[a], x2[a], pg, da, db);
    for (int p = 0; p < nd.npar; ++p) g[p] += pg[p];
  }
  return v;
}

// value + x1/x2 gradients into gx1[ndim], gx2[ndim]
GH_HD double gh_leaf_xgrad(const GhNode& nd, const double* x1, const double* x2, double* gx1, double* gx2) {
  for (int i = 0; i < nd.ndim; ++i) { gx1[i] = 0.0; gx2[i] = 0.0; }
  if (nd.mtype >= 0) {                    // templates/kernels.h:367-428
    if (gh_out_of_block(nd, x1, x2)) return 0.0;
    const double r2 = gh_metric<false>(nd, x1, x2, nullptr);
    double rg = 0.0, pg = 0.0;
    const double v = gh_radial<true>(nd, r2, rg, pg);
    gh_metric_x1(nd, x1, x2, gx1);
    const double f = 2.0 * rg;
    for (int i = 0; i < nd.ndim; ++i) { gx1[i] *= f; gx2[i] = -gx1[i]; }
    return v;
  }
  double v = 0.0;                         // templates/kernels.h:632-668
  for (int i = 0; i < nd.naxes; ++i) {
    const int a = nd.axes[i];
    double da, db;
    v += gh_axis<false, true>(nd, x1[a], x2[a], nullptr, da, db);
    gx1[a] = da; gx2[a] = db;
  }
  return v;
}

// ------------------------------------------------------- expression walkers
// Register-resident operand stack (no runtime-indexed arrays -> no scratch).
struct GhStack {
  double s0, s1, s2, s3, s4, s5, s6, s7;
  GH_HD void push(double v) { s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v; }
  GH_HD void drop() { s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7; }
};

// ---------------------------------------------------------------- fast affine form
// Most kernels people fit -- `c * Stationary(metric) [+ c']`, every BASELINE config -- reduce to
//     k(x1, x2) = a + b * F(r2(x1, x2))
// with ONE unblocked stationary leaf F over an isotropic / axis-aligned metric.  The host
// recognises that shape when the program is created (gh_kernel_create) and the value kernels then
// take this struct BY VALUE as a launch argument (SGPRs): no node reads from memory, no operand
// stack, one uniform switch.  Same formulas and evaluation order as the interpreter below.
struct GhFast {
  int ok, ktype, mtype, naxes;
  int axes[GH_MAX_AXES];
  double m[GH_MAX_AXES];     // exp(-log_M) per axis (isotropic: m[0])
  double a, b, q0;           // q0: alpha of RationalQuadratic
};

GH_HD double gh_fast_value(const GhFast& f, const double* x1, const double* x2) {
  double r2 = 0.0;
  if (f.mtype == 0) {
    double s = 0.0;
    for (int i = 0; i < f.naxes; ++i) { const double d = x1[f.axes[i]] - x2[f.axes[i]]; s += d * d; }
    r2 = s * f.m[0];
  } else {
    for (int i = 0; i < f.naxes; ++i) { const double d = x1[f.axes[i]] - x2[f.axes[i]]; r2 += d * d * f.m[i]; }
  }
  double v;
  switch (f.ktype) {
    case GH_K_EXPSQUARED: v = exp(-0.5 * r2); break;
    case GH_K_MATERN32: { const double r = sqrt(3.0 * r2); v = (1.0 + r) * exp(-r); break; }
    case GH_K_MATERN52: { const double r = sqrt(5.0 * r2); v = (1 + r + 5.0 * r2 / 3.0) * exp(-r); break; }
    case GH_K_EXP: v = exp(-sqrt(r2)); break;
    default: v = pow(1.0 + 0.5 * r2 / f.q0, -f.q0); break;        // GH_K_RATQUAD
  }
  {
#pragma clang fp contract(off)          // two roundings, like the interpreter's separate * and + nodes
    const double t = f.b * v;
    return f.a + t;
  }
}

// Sum / Product: kernels.h:75-80, 111-116
GH_HD double gh_eval_value(const GhNode* prog, int n_nodes, const double* x1, const double* x2) {
  GhStack st;
  st.s0 = st.s1 = st.s2 = st.s3 = st.s4 = st.s5 = st.s6 = st.s7 = 0.0;
  for (int i = 0; i < n_nodes; ++i) {
    const GhNode& nd = prog[i];
    if (nd.op == GH_OP_LEAF) {
      st.push(gh_leaf_value(nd, x1, x2));
    } else {
      const double b = st.s0, a = st.s1;
      st.drop();
      st.s0 = (nd.op == GH_OP_SUM) ? a + b : a * b;
    }
  }
  return st.s0;
}

// value + full parameter gradient g[0 .. size).  Product rule: kernels.h:117-141.
GH_HD double gh_eval_grad(const GhNode* prog, int n_nodes, const double* x1, const double* x2, double* g) {
  GhStack st;
  st.s0 = st.s1 = st.s2 = st.s3 = st.s4 = st.s5 = st.s6 = st.s7 = 0.0;
  for (int i = 0; i < n_nodes; ++i) {
    const GhNode& nd = prog[i];
    if (nd.op == GH_OP_LEAF) {
      st.push(gh_leaf_grad(nd, x1, x2, g + nd.poff));
    } else {
      const double b = st.s0, a = st.s1;
      st.drop();
      if (nd.op == GH_OP_SUM) {
        st.s0 = a + b;
      } else {
        for (int p = 0; p < nd.psize1; ++p) g[nd.poff + p] *= b;
        for (int p = nd.psize1; p < nd.psize; ++p) g[nd.poff + p] *= a;
        st.s0 = a * b;
      }
    }
  }
  return st.s0;
}

// value + x1 / x2 gradients (each [ndim]); uses a small explicit vector stack.
// Sum: kernels.h:93-108; Product: kernels.h:143-162.
GH_HD double gh_eval_xgrad(const GhNode* prog, int n_nodes, int ndim, const double* x1, const double* x2,
                           double* gx1, double* gx2) {
  double val[GH_MAX_STACK];
  double v1[GH_MAX_STACK][GH_MAX_NDIM], v2[GH_MAX_STACK][GH_MAX_NDIM];
  int sp = 0;
  for (int i = 0; i < n_nodes; ++i) {
    const GhNode& nd = prog[i];
    if (nd.op == GH_OP_LEAF) {
      val[sp] = gh_leaf_xgrad(nd, x1, x2, v1[sp], v2[sp]);
      ++sp;
    } else {
      const int ib = sp - 1, ia = sp - 2;
      if (nd.op == GH_OP_SUM) {
        for (int d = 0; d < ndim; ++d) { v1[ia][d] += v1[ib][d]; v2[ia][d] += v2[ib][d]; }
        val[ia] += val[ib];
      } else {
        const double ka = val[ia], kb = val[ib];
        for (int d = 0; d < ndim; ++d) {
          v1[ia][d] = kb * v1[ia][d] + ka * v1[ib][d];
          v2[ia][d] = kb * v2[ia][d] + ka * v2[ib][d];
        }
        val[ia] = ka * kb;
      }
      --sp;
    }
  }
  for (int d = 0; d < ndim; ++d) { gx1[d] = v1[0][d]; gx2[d] = v2[0][d]; }
  return val[0];
}
