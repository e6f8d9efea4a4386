"""ctypes binding of ``libgeorge_amd.so`` (the C ABI declared in
``include/george_amd.h``).

There is NO CPU fallback: if the HIP library is missing, importing this module
raises, and every entry point returns ``GH_ERR_HIP`` (raised here as
``RuntimeError``) when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgeorge_amd.so")

GH_MAX_AXES, GH_MAX_NDIM, GH_MAX_PARAMS, GH_MAX_METRIC = 8, 16, 4, 36
GH_MAX_NODES, GH_MAX_GRAD, GH_MAX_STACK = 64, 64, 8
GH_OK, GH_ERR_NOT_PD, GH_ERR_BAD_ARG, GH_ERR_HIP, GH_ERR_NOT_COMPUTED, GH_ERR_DIM, GH_ERR_NOMEM, GH_ERR_RANK = range(8)
GH_OP_LEAF, GH_OP_SUM, GH_OP_PRODUCT = 0, 1, 2


class gh_knode(C.Structure):
    _fields_ = [
        ("op", C.c_int32), ("kernel_type", C.c_int32), ("metric_type", C.c_int32),
        ("ndim", C.c_int32), ("naxes", C.c_int32), ("blocked", C.c_int32),
        ("n_params", C.c_int32), ("n_metric", C.c_int32),
        ("axes", C.c_int32 * GH_MAX_AXES),
        ("params", C.c_double * GH_MAX_PARAMS),
        ("constant", C.c_double),
        ("metric", C.c_double * GH_MAX_METRIC),
        ("min_block", C.c_double * GH_MAX_AXES),
        ("max_block", C.c_double * GH_MAX_AXES),
    ]


class gh_chol_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("nb", C.c_int32), ("profile", C.c_int32),
                ("lookahead", C.c_int32), ("reserved", C.c_int32 * 4)]


class gh_chol_profile(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_build", C.c_double), ("ms_panel", C.c_double),
                ("ms_trailing", C.c_double), ("trailing_flops", C.c_double), ("n_trailing", C.c_int64),
                ("ms_solve", C.c_double), ("ms_update_union", C.c_double), ("update_flops", C.c_double),
                ("reserved", C.c_double * 2)]


class gh_hodlr_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("min_size", C.c_int32), ("seed", C.c_int32),
                ("max_rank", C.c_int32), ("tol", C.c_double), ("reserved", C.c_int32 * 4)]


class gh_mgpu_opts(C.Structure):
    _fields_ = [("n_dev", C.c_int32), ("devices", C.c_int32 * 16), ("pr", C.c_int32), ("pc", C.c_int32),
                ("nb", C.c_int32), ("transport", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 3)]


class gh_hodlr_mgpu_opts(C.Structure):
    _fields_ = [("n_dev", C.c_int32), ("devices", C.c_int32 * 16), ("min_size", C.c_int32), ("seed", C.c_int32),
                ("max_rank", C.c_int32), ("tol", C.c_double), ("reserved", C.c_int32 * 4)]


GH_MGPU_RCCL, GH_MGPU_COPY = 0, 1
GH_MGPU_PLAIN_CYCLIC, GH_MGPU_CHAIN_ONLY, GH_MGPU_TRACE, GH_MGPU_ONE_COMM = 1, 2, 4, 8

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "george_amd: %s not found -- build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C george_amd/csrc)" % LIB_PATH)

def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process each
    claim the device and the second one to initialise sees "No HIP GPUs".  Because our library
    must interoperate with torch (device tensors, torch.distributed/RCCL), make both resolve to
    ONE runtime: if torch ships a bundled runtime, load it (RTLD_GLOBAL) before libgeorge_amd.so
    so the dynamic loader binds our DT_NEEDED libamdhip64 to it.  GEORGE_AMD_HIP_RUNTIME=system
    skips this (stand-alone use against /opt/rocm)."""
    if os.environ.get("GEORGE_AMD_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


_share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

_vp, _dp, _i64, _i32 = C.c_void_p, C.c_void_p, C.c_int64, C.c_int32   # data pointers are passed as raw addresses

# name -> (restype, argtypes); this table is what tests/test_abi.py checks against the header
SIGNATURES = {
    "gh_device_count": (C.c_int, []),
    "gh_last_error": (C.c_char_p, []),
    "gh_version": (C.c_char_p, []),
    "gh_release_caches": (None, [C.c_int32]),
    "gh_set_cache_limit": (None, [C.c_int64]),
    "gh_microbench_mfma_f64": (C.c_int, [C.POINTER(C.c_double)]),
    "gh_microbench_hbm_copy": (C.c_int, [C.POINTER(C.c_double)]),
    "gh_debug_set_mfma": (C.c_int, [C.c_int]),
    "gh_debug_set_gemm_sp": (C.c_int, [C.c_int]),
    "gh_debug_set_adaptive_panels": (C.c_int, [C.c_int]),
    "gh_debug_set_build_on_chain": (C.c_int, [C.c_int]),
    "gh_debug_set_gemm_grouped": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_passes": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_leaf_gj": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_wave_aca": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_coop_wgs": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_coop_lower": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_lpt": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_u_from_v": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_core_fused": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_coop_singles": (C.c_int, [C.c_int]),
    "gh_debug_set_hodlr_leaf_fused": (C.c_int, [C.c_int]),
    "gh_debug_stream_overlap": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int]),
    "gh_debug_stream_dispatch": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int]),
    "gh_microbench_suite": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "gh_microbench_mfma_f64_ceiling": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "gh_kernel_create": (C.c_int, [C.POINTER(gh_knode), C.c_int, C.POINTER(_vp)]),
    "gh_kernel_destroy": (None, [_vp]),
    "gh_kernel_ndim": (C.c_int, [_vp]),
    "gh_kernel_size": (C.c_int, [_vp]),
    "gh_kernel_value_general": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _dp]),
    "gh_kernel_value_symmetric": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_kernel_value_diagonal": (C.c_int, [_vp, _dp, _dp, _i64, _dp]),
    "gh_kernel_gradient_general": (C.c_int, [_vp, _dp, _dp, _i64, _dp, _i64, _dp]),
    "gh_kernel_gradient_symmetric": (C.c_int, [_vp, _dp, _dp, _i64, _dp]),
    "gh_kernel_x1_gradient_general": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _dp]),
    "gh_kernel_x2_gradient_general": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _dp]),
    "gh_chol_create": (C.c_int, [C.POINTER(gh_chol_opts), C.POINTER(_vp)]),
    "gh_chol_destroy": (None, [_vp]),
    "gh_chol_compute": (C.c_int, [_vp, _vp, _dp, _i64, _i32, _dp, C.POINTER(C.c_double)]),
    "gh_chol_info": (_i64, [_vp]),
    "gh_chol_size": (_i64, [_vp]),
    "gh_chol_device_bytes": (_i64, [_vp]),
    "gh_chol_solve": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_chol_dot_solve": (C.c_int, [_vp, _dp, C.POINTER(C.c_double)]),
    "gh_chol_apply_sqrt": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_chol_get_inverse": (C.c_int, [_vp, _dp]),
    "gh_chol_predict": (C.c_int, [_vp, _vp, _dp, _dp, _i64, _dp, _dp, _dp]),
    "gh_chol_grad": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp, _dp]),
    "gh_chol_objective": (C.c_int, [_vp, _vp, _dp, _i64, _i32, _dp, _dp, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    _dp, _dp, _dp]),
    "gh_chol_factor_size": (_i64, [_vp]),
    "gh_chol_dinv_size": (_i64, [_vp]),
    "gh_chol_export_factor": (C.c_int, [_vp, _dp, _dp]),
    "gh_chol_import_factor": (C.c_int, [_vp, _i64, _i32, _dp, _dp, _dp, C.c_double]),
    "gh_chol_trim": (None, [_vp]),
    "gh_chol_release_buffers": (None, [_vp]),
    "gh_chol_get_profile": (C.c_int, [_vp, C.POINTER(gh_chol_profile)]),
    "gh_chol_get_update_intervals": (C.c_int, [_vp, _dp, _i32, C.POINTER(C.c_int32)]),
    "gh_hodlr_create": (C.c_int, [C.POINTER(gh_hodlr_opts), C.POINTER(_vp)]),
    "gh_hodlr_destroy": (None, [_vp]),
    "gh_hodlr_compute": (C.c_int, [_vp, _vp, _dp, _i64, _i32, _dp, C.POINTER(C.c_double)]),
    "gh_hodlr_solve": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_hodlr_dot_solve": (C.c_int, [_vp, _dp, C.POINTER(C.c_double)]),
    "gh_hodlr_get_inverse": (C.c_int, [_vp, _dp]),
    "gh_hodlr_ranks": (C.c_int, [_vp, C.POINTER(C.c_int32), _i32, C.POINTER(C.c_int32)]),
    "gh_mgpu_create": (C.c_int, [C.POINTER(gh_mgpu_opts), C.POINTER(_vp)]),
    "gh_mgpu_destroy": (None, [_vp]),
    "gh_mgpu_compute": (C.c_int, [_vp, _vp, _dp, _i64, _i32, _dp, C.POINTER(C.c_double)]),
    "gh_mgpu_info": (_i64, [_vp]),
    "gh_mgpu_grid": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gh_mgpu_comm_mode": (C.c_int, [_vp]),
    "gh_mgpu_dot_solve": (C.c_int, [_vp, _dp, C.POINTER(C.c_double)]),
    "gh_mgpu_solve": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_mgpu_owner": (C.c_int, [_vp, _i64, _i64]),
    "gh_mgpu_apply_sqrt": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_mgpu_get_inverse": (C.c_int, [_vp, _dp]),
    "gh_mgpu_predict": (C.c_int, [_vp, _vp, _dp, _dp, _i64, _dp, _dp, _dp]),
    "gh_mgpu_get_trace": (C.c_int, [_vp, _dp, _i64, C.POINTER(C.c_int64)]),
    "gh_hodlr_mgpu_create": (C.c_int, [C.POINTER(gh_hodlr_mgpu_opts), C.POINTER(_vp)]),
    "gh_hodlr_mgpu_destroy": (None, [_vp]),
    "gh_hodlr_mgpu_compute": (C.c_int, [_vp, _vp, _dp, _i64, _i32, _dp, C.POINTER(C.c_double)]),
    "gh_hodlr_mgpu_solve": (C.c_int, [_vp, _dp, _i64, _dp]),
    "gh_hodlr_mgpu_dot_solve": (C.c_int, [_vp, _dp, C.POINTER(C.c_double)]),
    "gh_hodlr_mgpu_ranks": (C.c_int, [_vp, C.POINTER(C.c_int32), _i32, C.POINTER(C.c_int32)]),
    "gh_hodlr_mgpu_rows": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gh_hodlr_mgpu_layout": (C.c_int, [_i64, _i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), _i32,
                                       C.POINTER(C.c_int32)]),
    "gh_dev_kmat_block": (C.c_int, [_vp, _dp, _i64, _i32, _dp, _i64, _i64, _i64, _i64, _dp, _i64, _vp]),
    "gh_dev_gemv": (C.c_int, [_dp, _i64, _i64, _i64, _i32, _dp, _dp, C.c_double, C.c_double, _vp]),
    "gh_dev_potrf_block": (C.c_int, [_dp, _i64, _i64, _dp, _dp, _i64, _vp]),
    "gh_dev_trsm_right": (C.c_int, [_dp, _i64, _dp, _dp, _i64, _i64, _i64, _vp]),
    "gh_dev_gemm_nt": (C.c_int, [_dp, _i64, _dp, _i64, _dp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "gh_dev_gemm_nt_stair": (C.c_int, [_dp, _i64, _dp, _i64, _dp, _i64, _i64, _i32, C.POINTER(C.c_int64), _i64, _vp]),
    "gh_dev_gemm": (C.c_int, [_dp, _i64, _dp, _i64, _dp, _i64, _i64, _i64, _i64, C.c_double, C.c_double, _i32, _vp]),
    "gh_dev_logdet_accum": (C.c_int, [_dp, _i64, _i64, _dp, _vp]),
    "gh_dev_trsv_lower": (C.c_int, [_dp, _i64, _dp, _i64, _dp, _dp, _vp, _vp]),
    "gh_dev_trsv_lower_t": (C.c_int, [_dp, _i64, _dp, _i64, _dp, _dp, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)       # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    msg = lib.gh_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


class RankCeilingError(ValueError):
    """HODLR: a block needs a rank above the solver's ceiling for the requested tolerance
    (HODLRSolver.compute answers with the dense device solver instead where the matrix fits)."""


def check(rc):
    """Map a status code to the exception the reference raises at the same place."""
    if rc == GH_OK:
        return
    msg = last_error()
    if rc == GH_ERR_NOT_PD:
        raise np.linalg.LinAlgError(msg)               # scipy.linalg.cholesky, basic.py:68 / gp.py:356
    if rc == GH_ERR_BAD_ARG:
        raise ValueError(msg)                          # std::invalid_argument, parser.h:16,33
    if rc == GH_ERR_DIM:
        raise RuntimeError("dimension mismatch")       # george::dimension_mismatch, exceptions.h:8-12
    if rc == GH_ERR_NOT_COMPUTED:
        raise RuntimeError("you must call 'compute' first")   # george::not_computed, exceptions.h:14-18
    if rc == GH_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == GH_ERR_RANK:
        raise RankCeilingError(msg)
    raise RuntimeError("george_amd HIP backend failure: " + msg)


def ptr(a):
    """Raw address of a NumPy array, a torch tensor (``data_ptr``) or an int."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
