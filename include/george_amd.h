/*
 * george_amd.h -- C ABI of the MI355X-native GP solver backend for dfm/george.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point cites the reference interface it replaces (paths relative
 * to the dfm/george source tree).  The Python host layer (george_amd/ *.py)
 * binds these with ctypes; INTEGRATION.md shows the stub a george maintainer
 * would add.
 *
 * Conventions
 *   - all matrices are C-contiguous (row-major) fp64, exactly as the reference
 *     passes them through pybind11 / NumPy;
 *   - every data pointer may be a HOST pointer or a DEVICE (HBM) pointer; the
 *     library detects which (hipPointerGetAttributes) and stages host buffers
 *     itself.  Results are written to the memory space of the `out` pointer;
 *   - every function returns a status code (GH_OK == 0); gh_last_error() gives
 *     a thread-local message.  No exception crosses the boundary;
 *   - handles are opaque, owned by the library, freed by *_destroy, and not
 *     re-entrant (one call at a time per handle);
 *   - calls are synchronous: on return the result is complete.
 */
#ifndef GEORGE_AMD_H_
#define GEORGE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GH_MAX_AXES    8           /* max active axes of one leaf kernel */
#define GH_MAX_NDIM    16          /* max input dimension */
#define GH_MAX_PARAMS  4
#define GH_MAX_METRIC  36          /* GH_MAX_AXES*(GH_MAX_AXES+1)/2 */
#define GH_MAX_NODES   64
#define GH_MAX_GRAD    64          /* max full_size of a kernel expression */
#define GH_MAX_STACK   8           /* max postfix evaluation depth */

/* status codes.  The Python layer maps NOT_PD -> numpy.linalg.LinAlgError,
 * BAD_ARG -> ValueError, DIM -> RuntimeError("dimension mismatch")
 * (reference include/george/exceptions.h:8-12), NOT_COMPUTED ->
 * RuntimeError("you must call 'compute' first") (exceptions.h:14-18). */
enum {
  GH_OK = 0,
  GH_ERR_NOT_PD = 1,
  GH_ERR_BAD_ARG = 2,
  GH_ERR_HIP = 3,
  GH_ERR_NOT_COMPUTED = 4,
  GH_ERR_DIM = 5,
  GH_ERR_NOMEM = 6,
  GH_ERR_RANK = 7        /* HODLR: a block needs a rank above the solver's ceiling (1024) for the requested tol */
};

/* node operators */
enum { GH_OP_LEAF = 0, GH_OP_SUM = 1, GH_OP_PRODUCT = 2 };

/* leaf kernel ids == the reference's `kernel_type` class attributes
 * (src/george/kernels.py:273,322,391,464,500,546,588,666,712,753,822,897,944) */
enum {
  GH_K_LINEAR = 0, GH_K_RATQUAD = 1, GH_K_EXP = 2, GH_K_LOCALGAUSS = 3,
  GH_K_EMPTY = 4, GH_K_COSINE = 5, GH_K_MATERN52 = 6, GH_K_EXPSINE2 = 7,
  GH_K_CONSTANT = 8, GH_K_EXPSQUARED = 9, GH_K_MATERN32 = 10,
  GH_K_POLYNOMIAL = 11, GH_K_DOTPRODUCT = 12
};

/*
 * One node of a kernel expression, flattened in POSTFIX order (children
 * before their operator).  This POD replaces the heap-allocated C++ object
 * tree that `george::parse_kernel_spec` builds from the Python spec
 * (include/george/parser.h:14-35 operators, :344-403 stationary leaves);
 * the fields are exactly the attributes that parser reads.
 */
typedef struct gh_knode {
  int32_t op;                       /* GH_OP_*                                        */
  int32_t kernel_type;              /* GH_K_* (leaf only)                              */
  int32_t metric_type;              /* 0 isotropic, 1 axis-aligned, 2 general; -1 = non-stationary leaf */
  int32_t ndim;                     /* input dimension                                 */
  int32_t naxes;                    /* number of active axes                           */
  int32_t blocked;                  /* stationary leaves: `blocked` flag               */
  int32_t n_params;                 /* own parameters (e.g. log_alpha)                 */
  int32_t n_metric;                 /* metric parameters                               */
  int32_t axes[GH_MAX_AXES];
  double  params[GH_MAX_PARAMS];    /* own parameters, in parameter_names order        */
  double  constant;                 /* `order` for Linear/Polynomial                   */
  double  metric[GH_MAX_METRIC];    /* metric.get_parameter_vector(include_frozen=True) */
  double  min_block[GH_MAX_AXES];
  double  max_block[GH_MAX_AXES];
} gh_knode;

typedef struct gh_kernel gh_kernel;
typedef struct gh_chol   gh_chol;
typedef struct gh_hodlr  gh_hodlr;

/* ------------------------------------------------------------------ misc */
int         gh_device_count(void);
const char* gh_last_error(void);
const char* gh_version(void);
/* Device memory the library keeps for re-use -- released transient blocks (up to 128 GB per device, gh_set_cache_limit) -- is really freed.  The
 * library does this itself before it reports GH_ERR_NOMEM; an application that needs the memory for its own allocations
 * calls it directly (solver handles keep what they hold: gh_chol_trim / gh_chol_release_buffers / *_destroy). */
void gh_release_caches(int32_t device);
/* upper bound, per device, on the released device blocks the library keeps parked for its next allocation (default 128 GB of
 * the 288; they are otherwise returned to the driver only when one of the library's OWN allocations fails).  A process that
 * shares a GPU with another allocator (torch, RCCL buffers) lowers it; 0 = park nothing.  Blocks above the new bound are
 * freed at once.  (No reference counterpart: NumPy's allocator, basic.py:58.) */
void gh_set_cache_limit(int64_t bytes);
/* ---------------------------------------------- kernel-function evaluator
 * Replaces the pybind11 class KernelInterface, src/george/kernel_interface.cpp:
 *   ctor + parse_kernel_spec  :12-14   -> gh_kernel_create
 *   value_general             :47-60   -> gh_kernel_value_general
 *   value_symmetric           :62-77   -> gh_kernel_value_symmetric
 *   value_diagonal            :79-90   -> gh_kernel_value_diagonal
 *   gradient_general          :92-107  -> gh_kernel_gradient_general
 *   gradient_symmetric        :109-125 -> gh_kernel_gradient_symmetric
 *   x1_gradient_general       :127-141 -> gh_kernel_x1_gradient_general
 *   x2_gradient_general       :143-157 -> gh_kernel_x2_gradient_general
 * `which` is the uint32 mask of kernels.py:120; masked-out slots (left
 * uninitialised by the reference, kernels.h:85-91) are written as 0. */
int  gh_kernel_create(const gh_knode* nodes, int n_nodes, gh_kernel** out);
void gh_kernel_destroy(gh_kernel* k);
int  gh_kernel_ndim(const gh_kernel* k);
int  gh_kernel_size(const gh_kernel* k);          /* full_size */
int  gh_kernel_value_general(gh_kernel* k, const double* x1, int64_t n1,
                             const double* x2, int64_t n2, double* out /* n1*n2 */);
int  gh_kernel_value_symmetric(gh_kernel* k, const double* x, int64_t n, double* out /* n*n */);
int  gh_kernel_value_diagonal(gh_kernel* k, const double* x1, const double* x2,
                              int64_t n, double* out /* n */);
int  gh_kernel_gradient_general(gh_kernel* k, const uint32_t* which,
                                const double* x1, int64_t n1, const double* x2, int64_t n2,
                                double* out /* n1*n2*size */);
int  gh_kernel_gradient_symmetric(gh_kernel* k, const uint32_t* which,
                                  const double* x, int64_t n, double* out /* n*n*size */);
int  gh_kernel_x1_gradient_general(gh_kernel* k, const double* x1, int64_t n1,
                                   const double* x2, int64_t n2, double* out /* n1*n2*ndim */);
int  gh_kernel_x2_gradient_general(gh_kernel* k, const double* x1, int64_t n1,
                                   const double* x2, int64_t n2, double* out /* n1*n2*ndim */);

/* ------------------------------------------------------- dense Cholesky solver
 * Replaces BasicSolver (src/george/solvers/basic.py) and the SciPy/LAPACK
 * dpotrf/dpotrs it calls:
 *   compute        :51-70   (K = kernel(x); K_ii += yerr_i^2; cholesky; log_det)
 *   apply_inverse  :72-87   (cho_solve)            -> gh_chol_solve
 *   dot_solve      :89-102  (y . cho_solve(y))     -> gh_chol_dot_solve
 *   apply_sqrt     :104-114 (r @ U)                -> gh_chol_apply_sqrt
 *   get_inverse    :116-121                        -> gh_chol_get_inverse
 * plus fused device-resident forms of the GP glue around it
 *   GP.predict            src/george/gp.py:482-545 -> gh_chol_predict
 *   GP.grad_log_likelihood (kernel part) gp.py:429-466 -> gh_chol_grad
 * K is built on the device from (kernel, x) and never visits the host. */
typedef struct gh_chol_opts {
  int32_t device;        /* HIP device ordinal                                   */
  int32_t nb;            /* outer panel width (multiple of 128); 0 = default      */
  int32_t profile;       /* 1: record per-kernel-class hipEvent timings           */
  int32_t lookahead;     /* 1: overlap panel factorisation with trailing update   */
  int32_t reserved[4];
} gh_chol_opts;

int  gh_chol_create(const gh_chol_opts* opts, gh_chol** out);
void gh_chol_destroy(gh_chol* s);
/* yerr: (n,) standard deviations ALREADY including white noise (gp.py:330). */
int  gh_chol_compute(gh_chol* s, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                     const double* yerr, double* logdet_out);
int64_t gh_chol_info(const gh_chol* s);     /* 1-based index of the failing pivot after GH_ERR_NOT_PD, else 0 */
int64_t gh_chol_size(const gh_chol* s);
int64_t gh_chol_device_bytes(const gh_chol* s);  /* HBM bytes the handle holds right now (factor + work arrays) */
int  gh_chol_solve(gh_chol* s, const double* b, int64_t nrhs, double* out);   /* b, out: (n, nrhs) row-major; may alias */
int  gh_chol_dot_solve(gh_chol* s, const double* y, double* out);
int  gh_chol_apply_sqrt(gh_chol* s, const double* r, int64_t nrows, double* out); /* r, out: (nrows, n) */
int  gh_chol_get_inverse(gh_chol* s, double* out /* n*n */);
/* mu = K(xs,x) K^-1 r ; var = diag(K(xs,xs)) - diag(K(xs,x) K^-1 K(x,xs)) ;
 * cov = K(xs,xs) - K(xs,x) K^-1 K(x,xs).   var / cov may be NULL. */
int  gh_chol_predict(gh_chol* s, gh_kernel* k, const double* r /* n: y - mean */,
                     const double* xs, int64_t m, double* mu /* m */,
                     double* var /* m or NULL */, double* cov /* m*m or NULL */);
/* alpha = K^-1 r; A = alpha alpha^T - K^-1; grad[p] = 1/2 sum_ij A_ij dK_ij/dtheta_p
 * for the parameters selected by `which` (others 0); diagA (n) = diag(A). */
int  gh_chol_grad(gh_chol* s, gh_kernel* k, const uint32_t* which, const double* r,
                  double* grad /* size */, double* alpha /* n or NULL */, double* diagA /* n or NULL */);
/* Fused hyper-parameter objective: GP.nll + GP.grad_nll (src/george/gp.py:470-480, i.e. compute
 * gp.py:303-337 + log_likelihood :369-397 + the kernel part of grad_log_likelihood :429-466) as ONE
 * device-resident call with one synchronisation: build K, factor, *logdet = log|K|,
 * *quad = r^T K^-1 r, and -- when grad != NULL -- alpha = K^-1 r (from the same forward solve),
 * K^-1 and grad[p] = 1/2 sum_ij A_ij dK_ij/dtheta_p for the parameters selected by `which`.
 * grad / alpha / diagA may be NULL; the handle is left computed (solve / predict may follow). */
int  gh_chol_objective(gh_chol* s, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                       const double* yerr, const double* r /* n: y - mean */, const uint32_t* which,
                       double* logdet, double* quad, double* grad /* size or NULL */,
                       double* alpha /* n or NULL */, double* diagA /* n or NULL */);
/* Checkpointing of the device factor (the reference's BasicSolver pickles COMPUTED,
 * tests/test_pickle.py:21-36, because its factor is a NumPy array, basic.py:68): the lower
 * triangle of L packed by rows (gh_chol_factor_size() = n (n + 1) / 2 doubles) and the inverses of
 * its 128 x 128 diagonal blocks (gh_chol_dinv_size() doubles).  import_factor() rebuilds a computed
 * handle from them plus the inputs x (n, ndim) that predict / grad evaluate kernels against. */
int64_t gh_chol_factor_size(const gh_chol* s);
int64_t gh_chol_dinv_size(const gh_chol* s);
int  gh_chol_export_factor(gh_chol* s, double* packed_lower, double* dinv_out);
int  gh_chol_import_factor(gh_chol* s, int64_t n, int32_t ndim, const double* x,
                           const double* packed_lower, const double* dinv_in, double logdet);
/* memory management of a long-lived handle: trim() frees the transient work buffers of predict /
 * grad / get_inverse (up to 3 x 8 N^2 bytes) and keeps the factor; release_buffers() frees
 * everything but the handle (streams, events) -- the next compute() re-allocates. */
void gh_chol_trim(gh_chol* s);
void gh_chol_release_buffers(gh_chol* s);
/* profile counters of the last compute(): see george_amd/csrc/gh_chol.hip */
typedef struct gh_chol_profile {
  double ms_total;          /* build + factor, device time                     */
  double ms_build;          /* kernel-matrix build                             */
  double ms_panel;          /* potf2 + trsm + inner updates                    */
  double ms_trailing;       /* sum of trailing-update (SYRK) launches          */
  double trailing_flops;    /* algorithmic flops of those launches             */
  int64_t n_trailing;       /* number of trailing-update launches              */
  double ms_solve;          /* last dot_solve / solve                          */
  double ms_update_union;   /* time during which ANY trailing-update launch ran (wide SYRKs on the main
                             * stream + block-column GEMMs on the chain stream): union of their intervals */
  double update_flops;      /* algorithmic flops of all those launches          */
  double reserved[2];
} gh_chol_profile;
int  gh_chol_get_profile(const gh_chol* s, gh_chol_profile* out);
/* per trailing-update launch of the last profiled compute(): (start ms, end ms, algorithmic flops), times from the
 * start of compute() by HIP events on the stream each launch went to; *n_out = number of launches recorded.  The
 * union of these intervals is gh_chol_profile.ms_update_union (bench.py's roofline: reproducible from them). */
int  gh_chol_get_update_intervals(const gh_chol* s, double* out /* 3 * max_launches */, int32_t max_launches, int32_t* n_out);

/* ------------------------------------------------------------ HODLR solver
 * Replaces HODLRSolver (src/george/solvers/hodlr.py:13-76), the pybind11
 * `Solver` (src/george/solvers/_hodlr.cpp:38-110) and hodlr::Node
 * (include/george/hodlr.h:13-258). */
typedef struct gh_hodlr_opts {
  int32_t device;
  int32_t min_size;      /* hodlr.py:43 default 100 */
  int32_t seed;          /* hodlr.py:43 default 42  */
  int32_t max_rank;      /* cap on the ACA rank per block; 0 = default (256) */
  double  tol;           /* hodlr.py:43 default 0.1 */
  int32_t reserved[4];
} gh_hodlr_opts;

int  gh_hodlr_create(const gh_hodlr_opts* opts, gh_hodlr** out);
void gh_hodlr_destroy(gh_hodlr* h);
int  gh_hodlr_compute(gh_hodlr* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                      const double* yerr, double* logdet_out);
int  gh_hodlr_solve(gh_hodlr* h, const double* b, int64_t nrhs, double* out);  /* (n, nrhs) row-major */
int  gh_hodlr_dot_solve(gh_hodlr* h, const double* y, double* out);
int  gh_hodlr_get_inverse(gh_hodlr* h, double* out /* n*n */);
int  gh_hodlr_ranks(const gh_hodlr* h, int32_t* ranks_out, int32_t max_out, int32_t* n_out);

/* --------------------------------------------------- dense solver on several GPUs
 * The BasicSolver protocol of /root/reference/src/george/solvers/basic.py:51-102 (compute, log-determinant,
 * dot_solve, apply_inverse) for ONE process that owns several MI355X: 2-D block-cyclic tiles (tile (I, J)
 * on rank (I mod pr) * pc + (J mod pc)), one host thread and one stream per device, panels moved with RCCL
 * over xGMI (grouped ncclSend / ncclRecv on the communicator of ncclCommInitAll; librccl is dlopen'ed at the
 * first create).  The reference has nothing here: it is single-process, single-host (gp.py:327).  The
 * multi-PROCESS form of the same algorithm (one rank per GPU under torch.distributed; what bench.py
 * --gpus N runs) is george_amd/distributed.py.  Default grid: pr = n_dev, pc = 1 -- whole tile rows per rank in
 * "snake" order (0 1 .. P-1 P-1 .. 1 0: equal shares of the lower triangle) -- because on the xGMI full mesh
 * the critical chain potrf(k) -> TRSM -> block column k+1 -> potrf(k+1) then moves only two nb x nb tiles per
 * step, over all links at once (george_amd/csrc/gh_mgpu.hip; profiles/r04/scale_model.md).  The whole solver
 * protocol is offered: the O(N^2 R) operations are left-looking tile sweeps on the sharded factor. */
typedef struct gh_mgpu gh_mgpu;
enum {
  GH_MGPU_RCCL = 0,      /* RCCL point-to-point over xGMI; one rank per physical device                      */
  GH_MGPU_COPY = 1       /* peer copies behind events; the same device may be listed several times ("virtual
                            devices": exercises the n_dev-rank ownership and ordering logic on one GPU)     */
};
enum {
  GH_MGPU_PLAIN_CYCLIC = 1,  /* pc == 1: tile row I on rank I mod pr instead of the snake order                    */
  /* (2 and 4 are the timing aids GH_MGPU_CHAIN_ONLY / GH_MGPU_TRACE: include/george_amd_debug.h -- a chain-only compute()
   *  returns GH_OK with numbers that mean nothing, which has no place among the drop-in's flags)                          */
  GH_MGPU_ONE_COMM     = 8   /* GH_MGPU_RCCL: the bulk gather on the chain's stream and communicator (one communicator in
                                flight at a time) -- chosen by itself when the two streams of a rank do not dispatch
                                independently (gh_mgpu_comm_mode tells which)                                         */
};
typedef struct gh_mgpu_opts {
  int32_t n_dev;             /* 1..16 */
  int32_t devices[16];       /* HIP device ordinals, rank i runs on devices[i] */
  int32_t pr, pc;            /* process grid, pr * pc == n_dev; 0, 0: n_dev x 1 (whole tile rows per rank) */
  int32_t nb;                /* tile edge, multiple of 128; 0: 1024 from N = 24576 up, else 512 */
  int32_t transport;         /* GH_MGPU_RCCL | GH_MGPU_COPY */
  int32_t flags;             /* GH_MGPU_PLAIN_CYCLIC | GH_MGPU_ONE_COMM (| the timing aids of george_amd_debug.h) */
  int32_t reserved[3];
} gh_mgpu_opts;
int  gh_mgpu_create(const gh_mgpu_opts* opts, gh_mgpu** out);      /* communicators + an all-reduce self-check */
void gh_mgpu_destroy(gh_mgpu* h);
/* basic.py:51-70 on the grid: every rank builds its own tiles from (kernel, x, yerr) -- host pointers --
 * and the factor stays sharded; GH_ERR_NOT_PD + gh_mgpu_info() as gh_chol_compute */
int  gh_mgpu_compute(gh_mgpu* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                     const double* yerr, double* logdet_out);
int64_t gh_mgpu_info(const gh_mgpu* h);
int  gh_mgpu_grid(const gh_mgpu* h, int32_t* pr, int32_t* pc, int32_t* nb);
int  gh_mgpu_comm_mode(const gh_mgpu* h);   /* 2: chain and bulk gather on their own communicators; 1: GH_MGPU_ONE_COMM; 0: GH_MGPU_COPY */
int  gh_mgpu_owner(const gh_mgpu* h, int64_t tile_row, int64_t tile_col);              /* rank that holds tile (I, J); -1 on bad arguments */
/* all of the following take HOST pointers; every right-hand side of a call is swept together (chunks of 2048 columns) */
int  gh_mgpu_dot_solve(gh_mgpu* h, const double* y, double* out);                     /* basic.py:89-102 */
int  gh_mgpu_solve(gh_mgpu* h, const double* b, int64_t nrhs, double* out);           /* basic.py:72-87; (n, nrhs) row-major, may alias */
int  gh_mgpu_apply_sqrt(gh_mgpu* h, const double* r, int64_t nrows, double* out);     /* basic.py:104-114; r, out: (nrows, n) */
int  gh_mgpu_get_inverse(gh_mgpu* h, double* out /* n*n */);                          /* basic.py:116-121 */
/* gp.py:482-545 on the sharded factor (forward sweep only: K* K^-1 K*^T = V^T V with V = L^-1 K(x, xs));
 * var / cov may be NULL; cov is offered for m <= 2048 */
int  gh_mgpu_predict(gh_mgpu* h, gh_kernel* k, const double* r /* n: y - mean */, const double* xs, int64_t m,
                     double* mu /* m */, double* var /* m or NULL */, double* cov /* m*m or NULL */);
/* --------------------------------------------------- HODLR solver, tree split over several GPUs
 * hodlr::Node (include/george/hodlr.h:29-254) with the top log2(n_dev) levels of the tree shared and the
 * n_dev sub-trees below them -- independent of each other (hodlr.h:75-103: a node's factorisation touches
 * its own rows only) -- one per device.  ONE process, a host thread per device.  Per compute():
 *   - the ACA of each of the n_dev - 1 top nodes runs on one device against the whole point set, and every
 *     device pulls the rows it owns of each ancestor's factors (peer copies, N x r doubles per node in all);
 *   - every device builds and factors its sub-tree (the single-GPU code on rows [row0, row0 + n_p)), carrying
 *     the ancestors' U rows along as extra right-hand-side columns;
 *   - a top node's 2r x 2r core needs V^T U summed over the devices below it: 2r x C doubles exchanged
 *     through pinned host memory per level -- the only data-path exchange, also in every solve.
 * Same node-by-node random streams as the single-GPU solver (a node's generator is keyed by its position
 * in the global tree), so ranks and results agree with gh_hodlr_* to rounding.  n_dev must be a power of
 * two and every top node internal (N / n_dev >= 2 min_size); x, yerr, b, y are HOST pointers.  The same
 * device may be listed several times ("virtual devices": the split is then exercised on one GPU).
 * The reference has nothing here (single process, single thread). */
typedef struct gh_hodlr_mgpu gh_hodlr_mgpu;
typedef struct gh_hodlr_mgpu_opts {
  int32_t n_dev;             /* 1, 2, 4, 8 or 16 */
  int32_t devices[16];       /* HIP device ordinals; sub-tree p (rows in tree order) lives on devices[p] */
  int32_t min_size;          /* as gh_hodlr_opts */
  int32_t seed;
  int32_t max_rank;
  double  tol;
  int32_t reserved[4];
} gh_hodlr_mgpu_opts;
int  gh_hodlr_mgpu_create(const gh_hodlr_mgpu_opts* opts, gh_hodlr_mgpu** out);
void gh_hodlr_mgpu_destroy(gh_hodlr_mgpu* h);
int  gh_hodlr_mgpu_compute(gh_hodlr_mgpu* h, gh_kernel* k, const double* x, int64_t n, int32_t ndim,
                           const double* yerr, double* logdet_out);                          /* hodlr.h:75-103 */
int  gh_hodlr_mgpu_solve(gh_hodlr_mgpu* h, const double* b, int64_t nrhs, double* out);      /* hodlr.h:107-114; (n, nrhs) row-major */
int  gh_hodlr_mgpu_dot_solve(gh_hodlr_mgpu* h, const double* y, double* out);                /* hodlr.h:116-120 */
/* ranks of all internal nodes, level by level, left to right (the order gh_hodlr_ranks uses) */
int  gh_hodlr_mgpu_ranks(const gh_hodlr_mgpu* h, int32_t* ranks_out, int32_t max_out, int32_t* n_out);
/* host logic only, no device needed: the rows of every sub-tree of an n-point tree split over n_dev devices (hodlr.h:47-64
 * applied log2(n_dev) times) and, per sub-tree level, the index of each sub-tree's first internal node in the global
 * level (seed_off: n_dev x max_levels, may be NULL); *n_levels = levels of the deepest sub-tree */
int  gh_hodlr_mgpu_layout(int64_t n, int32_t n_dev, int32_t min_size, int64_t* row0, int64_t* nrows,
                          int32_t* seed_off, int32_t max_levels, int32_t* n_levels);
/* rows [row0[p], row0[p] + nrows[p]) live on devices[p]; arrays of n_dev entries (after compute()) */
int  gh_hodlr_mgpu_rows(const gh_hodlr_mgpu* h, int64_t* row0, int64_t* nrows);

/* ------------------------------------------- device-level tile operations
 * Building blocks of the blocked factorisation on DEVICE pointers and an
 * explicit hipStream_t (passed as void*; NULL = default stream).  Used by the
 * multi-GPU 2-D block-cyclic driver (george_amd/distributed.py), which moves
 * panels between ranks with torch.distributed (RCCL) in between.  Leading
 * dimensions are in elements; all sizes must be multiples of 128. */
/* out[r, c] = k(x[row0+r], x[col0+c]) (+ yerr[row0+r]^2 where row0+r == col0+c); x / yerr are the
 * full n-point device arrays, rows / columns past n are identity padding */
int gh_dev_kmat_block(gh_kernel* k, const double* x, int64_t n, int32_t ndim, const double* yerr,
                      int64_t row0, int64_t nrows, int64_t col0, int64_t ncols,
                      double* out, int64_t ldo, void* stream);
/* y = beta*y + alpha * A x (trans == 0; A is m x n row-major) or alpha * A^T x (trans != 0) */
int gh_dev_gemv(const double* a, int64_t lda, int64_t m, int64_t n, int32_t trans,
                const double* x, double* y, double alpha, double beta, void* stream);
/* z = L^-1 w for the n x n lower-triangular block `l` factored by gh_dev_potrf_block (`dinv` = its
 * diagonal-block inverses): one chained launch (gh_chol.hip, trsv_fwd_chain_direct; w and z must be different arrays).  `scratch` needs
 * (n/128 + 1) * 4 bytes of device memory, zeroed here.  w is read only. */
int gh_dev_trsv_lower(const double* l, int64_t ldl, const double* dinv, int64_t n,
                      const double* w, double* z, void* scratch, void* stream);
/* x = L^-T w for the same block (trsv_bwd_chain); same scratch; w is read only.  After either call
 * ((int32_t*)scratch)[n/128] != 0 means a workgroup gave up waiting for its predecessor (2 s). */
int gh_dev_trsv_lower_t(const double* l, int64_t ldl, const double* dinv, int64_t n,
                        const double* w, double* x, void* scratch, void* stream);
/* in-place lower Cholesky of the n x n block `a`; dinv receives the inverses of
 * its 128x128 diagonal blocks, (n/128) x 128 x 128; *info_dev (device int64) is
 * set to base_index + failing pivot (1-based) when not positive definite. */
int gh_dev_potrf_block(double* a, int64_t lda, int64_t n, double* dinv,
                       int64_t* info_dev, int64_t base_index, void* stream);
/* a21 (m x n) <- a21 * L11^-T using L11 (n x n, lower) and its diagonal-block inverses */
int gh_dev_trsm_right(const double* l11, int64_t ld11, const double* dinv,
                      double* a21, int64_t lda, int64_t m, int64_t n, void* stream);
/* c (m x n) -= a (m x k) * b (n x k)^T ; lower != 0: only tiles on/below the diagonal of the
 * square c are updated (SYRK-shaped, a and b rows aligned with c rows/cols). */
int gh_dev_gemm_nt(double* c, int64_t ldc, const double* a, int64_t lda,
                   const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k,
                   int32_t lower, void* stream);
/* the same update on a STAIRCASE-shaped c, as ONE launch: row group g (group_rows rows of c and of a, g = 0 .. n_groups-1) is
 * updated over its first group_cols[g] columns, c[g-th rows, 0:group_cols[g]) -= a[g-th rows] * b[0:group_cols[g])^T; group_cols
 * (a HOST array) is non-decreasing, everything a multiple of 128.  The per-rank trailing update of a solver that owns whole tile
 * rows: every tile row reaches as far as its own diagonal tile.  Bit-identical to one gh_dev_gemm_nt per group. */
int gh_dev_gemm_nt_stair(double* c, int64_t ldc, const double* a, int64_t lda, const double* b, int64_t ldb,
                         int64_t group_rows, int32_t n_groups, const int64_t* group_cols, int64_t k, void* stream);
/* general form: c = beta*c + alpha * sum_k A(m,k) B(n,k); by default A(m,k) = a[m*lda + k] and
 * B(n,k) = b[n*ldb + k] ("k-major"); flags select the transposed layouts, SYRK-style lower-only
 * tile sets and k-range clipping for triangular operands. */
enum {
  GH_GEMM_A_MMAJOR = 1,   /* A(m,k) = a[k*lda + m]                                   */
  GH_GEMM_B_NMAJOR = 2,   /* B(n,k) = b[k*ldb + n]                                   */
  GH_GEMM_LOWER    = 4,   /* m >= n: only tiles on/below the diagonal (a lower trapezoid) */
  GH_GEMM_KLO_MAX  = 8,   /* k starts at max(tile row0, tile col0)                   */
  GH_GEMM_KHI_COL  = 16,  /* k ends at tile col0 + 128                               */
  GH_GEMM_KHI_ROW  = 32   /* k ends at tile row0 + 128                               */
};
int gh_dev_gemm(double* c, int64_t ldc, const double* a, int64_t lda,
                const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k,
                double alpha, double beta, int32_t flags, void* stream);
/* sum_i 2*log(a[i*lda+i]) over the n diagonal entries, accumulated into *out_dev */
int gh_dev_logdet_accum(const double* a, int64_t lda, int64_t n, double* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* GEORGE_AMD_H_ */
