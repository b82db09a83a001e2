"""ctypes declarations of the C-ABI in include/albatross_amd.h.

Only the struct layouts and the loader live here.  The product wrapper loads
libalbatross_amd.so and nothing else; the struct types are also reused by the
test-side CPU checker's wrapper.
"""
import ctypes as C
import os

AGP_OK = 0
AGP_ERR_INVALID_ARGUMENT = 1
AGP_ERR_NAN_INPUT = 2
AGP_ERR_NOT_POSITIVE_DEFINITE = 3
AGP_ERR_HIP = 4
AGP_ERR_COMM = 5
AGP_ERR_UNSUPPORTED = 6
AGP_ERR_NO_DEVICE = 7

OP_SQUARED_EXPONENTIAL = 1
OP_EXPONENTIAL = 2
OP_MATERN32 = 3
OP_MATERN52 = 4
OP_CONSTANT = 5
OP_INDEPENDENT_NOISE = 6
OP_NUGGET = 7
OP_POLYNOMIAL = 8
OP_SCALING = 9
OP_SUM = 10
OP_PRODUCT = 11
OP_MEASUREMENT_ONLY = 12
OP_TYPE_PAIR = 13

METRIC_EUCLIDEAN = 0
METRIC_RADIAL = 1
METRIC_ANGULAR = 2

HOST = 0
DEVICE = 1

MAX_KERNEL_NODES = 32
MAX_STACK = 8
MAX_DIM = 8
MAX_SCALE_COLUMNS = 4


class KernelNode(C.Structure):
    _fields_ = [
        ("op", C.c_int32),
        ("metric", C.c_int32),
        ("column", C.c_int32),
        ("order", C.c_int32),
        ("params", C.c_double * 4),
    ]


class Features(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("dim", C.c_int32),
        ("n_scale_columns", C.c_int32),
        ("coords", C.c_void_p),
        ("eq_id", C.c_void_p),
        ("scales", C.c_void_p),
        ("is_measurement", C.c_int32),
        ("location", C.c_int32),
    ]


# every symbol include/albatross_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_D = C.POINTER(C.c_double)
EXPORTS = [
    ("agp_context_create", C.c_int, [C.c_int, _PP]),
    ("agp_context_destroy", None, [_P]),
    ("agp_context_synchronize", C.c_int, [_P]),
    ("agp_last_error", C.c_char_p, [_P]),
    ("agp_status_string", C.c_char_p, [C.c_int]),
    ("agp_device_count", C.c_int, []),
    ("agp_device_malloc", C.c_int, [_P, C.c_int64, _PP]),
    ("agp_device_free", C.c_int, [_P, _P]),
    ("agp_memcpy", C.c_int, [_P, _P, _P, C.c_int64, C.c_int]),
    ("agp_kernel_create", C.c_int, [C.POINTER(KernelNode), C.c_int, _PP]),
    ("agp_kernel_destroy", None, [_P]),
    ("agp_gram", C.c_int, [_P, _P, C.POINTER(Features), C.POINTER(Features), _P, C.c_int64, C.c_int]),
    ("agp_fit_create", C.c_int, [_P, _P, C.POINTER(Features), _P, _P, _PP, _P, _P]),
    ("agp_fit_create_mixed", C.c_int, [_P, _P, C.POINTER(Features), _P, _P, C.c_int, C.c_double, _PP, _P, _P, _P, _P]),
    ("agp_fit_destroy", None, [_P]),
    ("agp_fit_size", C.c_int64, [_P]),
    ("agp_fit_failed_pivot", C.c_int64, [_P]),
    ("agp_fit_log_determinant", C.c_int, [_P, _D]),
    ("agp_fit_download_factor", C.c_int, [_P, _P, _P, C.c_int64]),
    ("agp_fit_download_information", C.c_int, [_P, _P, _P]),
    ("agp_nll", C.c_int, [_P, _P, C.POINTER(Features), _P, _P, _D]),
    ("agp_nll_batch", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64, _P, _P]),
    ("agp_fit_create_batch", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, _P, _P]),
    ("agp_solve", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int]),
    ("agp_factor_create", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _PP]),
    ("agp_nll_dense", C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _D]),
    ("agp_fit_inverse_diagonal", C.c_int, [_P, _P, _P, C.c_int]),
    ("agp_loo_marginal", C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    ("agp_ldlt_create", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, _PP, C.POINTER(C.c_int)]),
    ("agp_ldlt_destroy", None, [_P]),
    ("agp_ldlt_size", C.c_int64, [_P]),
    ("agp_ldlt_solve", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int]),
    ("agp_ldlt_sqrt_solve", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int]),
    ("agp_ldlt_vector_d", C.c_int, [_P, _P]),
    ("agp_ldlt_transpositions", C.c_int, [_P, _P]),
    ("agp_ldlt_download", C.c_int, [_P, _P, _P, C.c_int64]),
    ("agp_sparse_fit_create", C.c_int, [_P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P, C.POINTER(Features), C.c_double, C.c_double, _PP, _P, _D]),
    ("agp_sparse_fit_create_sharded", C.c_int, [_P, _P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P, C.POINTER(Features), C.c_double, C.c_double, _PP, _P, _D]),
    ("agp_sparse_fit_update", C.c_int, [_P, _P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P, C.c_double, _PP, _P]),
    ("agp_sparse_fit_from_prediction", C.c_int, [_P, _P, C.POINTER(Features), _P, _P, C.c_int64, C.c_int, C.c_double, _PP, _P, _P]),
    ("agp_sparse_fit_numerical_rank", C.c_int64, [_P]),
    ("agp_sparse_fit_destroy", None, [_P]),
    ("agp_sparse_fit_size", C.c_int64, [_P]),
    ("agp_sparse_fit_information", C.c_int, [_P, _P, _P]),
    ("agp_sparse_nll", C.c_int, [_P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P, C.POINTER(Features), C.c_double, C.c_double, _D]),
    ("agp_sparse_predict_mean", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, C.c_int]),
    ("agp_sparse_predict_marginal", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, C.c_int]),
    ("agp_sparse_predict_joint", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, C.c_int]),
    ("agp_fit_inverse_blocks", C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int]),
    ("agp_held_out_predictions", C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int]),
    ("agp_predict_mean", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, C.c_int]),
    ("agp_predict_marginal", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, C.c_int]),
    ("agp_predict_joint", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, C.c_int]),
    ("agp_gram_combined", C.c_int, [_P, _P, C.POINTER(Features), C.c_int64, _P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P,
                           C.c_int64, C.c_int]),
    ("agp_fit_update", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, _PP, _P, _D]),
    ("agp_solver_from_fit", C.c_int, [_P, _P, _PP]),
    ("agp_solver_from_ldlt", C.c_int, [_P, _P, _PP]),
    ("agp_solver_block_symmetric", C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _PP]),
    ("agp_solver_explained", C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _PP]),
    ("agp_solver_rows", C.c_int64, [_P]),
    ("agp_solver_solve", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int]),
    ("agp_solver_predict", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, C.POINTER(Features), _P, _P, C.c_int, C.c_int]),
    ("agp_solver_update_information", C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    ("agp_solver_predict_combined", C.c_int, [_P, _P, _P, C.POINTER(Features), C.c_int64, _P, _P, _P, C.POINTER(Features), C.c_int64, _P, _P,
                                              _P, _P, C.c_int, C.c_int]),
    ("agp_solver_destroy", None, [_P]),
    ("agp_comm_unique_id", C.c_int, [_P]),
    ("agp_comm_create", C.c_int, [_P, C.c_int, C.c_int, _P, _PP]),
    ("agp_comm_create_callbacks", C.c_int, [C.c_int, C.c_int, _P, _PP]),
    ("agp_comm_create_ipc", C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, _PP]),
    ("agp_comm_destroy", None, [_P]),
    ("agp_comm_size", C.c_int, [_P]),
    ("agp_comm_rank", C.c_int, [_P]),
    ("agp_comm_all_reduce_host", C.c_int, [_P, _P, C.c_int64, C.c_int]),
    ("agp_comm_barrier", C.c_int, [_P]),
    ("agp_shard_local_rows", C.c_int64, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    ("agp_shard_global_row", C.c_int64, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64]),
    ("agp_shard_owner", C.c_int, [C.c_int64, C.c_int]),
    ("agp_sharded_fit_create", C.c_int, [_P, _P, _P, C.POINTER(Features), _P, _P, _PP, _P, _D]),
    ("agp_sharded_fit_destroy", None, [_P]),
    ("agp_sharded_fit_failed_pivot", C.c_int64, [_P]),
    ("agp_sharded_fit_replicate", C.c_int, [_P, _P, _PP]),
    ("agp_sharded_predict_marginal", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int]),
    ("agp_sharded_predict_joint", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int]),
    ("agp_sharded_fit_stage", C.c_int, [_P, C.c_int, _D]),
    ("agp_last_stage_ms", C.c_int, [_P, C.c_int, _D]),
    ("agp_set_profiling", C.c_int, [_P, C.c_int]),
]

COMM_ID_BYTES = 128

# agp_comm_callbacks / agp_shard_ops_callbacks (include/albatross_amd.h): collectives and block arithmetic supplied by
# the caller (tests; one-GPU boxes).  Pointers arrive as plain integers (c_void_p).
BROADCAST_FN = C.CFUNCTYPE(C.c_int, _P, _P, C.c_int64, C.c_int)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, _P, _P, _P, C.c_int64)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, _P, _P, C.c_int64, C.c_int)


class CommCallbacks(C.Structure):
    _fields_ = [("user", _P), ("broadcast", BROADCAST_FN), ("all_gather", ALL_GATHER_FN), ("all_reduce", ALL_REDUCE_FN)]


FACTOR_DIAG_FN = C.CFUNCTYPE(C.c_int64, _P, _P, C.c_int64, C.c_int64, _P, _P, _D)
TRSM_ROWS_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P)
GEMM_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int)
COPY2D_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64)
INVERT_DIAG_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64, C.c_int64, _P, _P)
COLVEC_DOT_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_double, C.c_double, _P, _P)
AXPBY_FN = C.CFUNCTYPE(None, _P, C.c_int64, C.c_double, _P, C.c_double, _P, _P)
FILL_ZERO_FN = C.CFUNCTYPE(None, _P, _P, C.c_int64)


class ShardOpsCallbacks(C.Structure):
    _fields_ = [("user", _P), ("factor_diag", FACTOR_DIAG_FN), ("trsm_rows", TRSM_ROWS_FN), ("gemm", GEMM_FN),
                ("copy2d", COPY2D_FN), ("invert_diag", INVERT_DIAG_FN), ("colvec_dot", COLVEC_DOT_FN),
                ("axpby", AXPBY_FN), ("fill_zero", FILL_ZERO_FN)]


LIB_NAME = "libalbatross_amd.so"


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


_lib = None


_debug_lib = None


def load_debug():
    """libalbatross_amd_debug.so: the product objects plus the kernel-level probes `agp_debug_*` (csrc/debug_api.hip).
    For tests/ and scripts/ only - the package itself never loads it.  Handles made by the product library (contexts)
    are plain structs of the same build and are accepted by these entry points."""
    global _debug_lib
    if _debug_lib is None:
        path = os.path.join(os.path.dirname(lib_path()), "libalbatross_amd_debug.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C albatross_amd/csrc`")
        _debug_lib = C.CDLL(path)
        # the test-only entry points with pointer / 64-bit arguments (csrc/debug_api.hip)
        _debug_lib.agp_debug_shard_work_doubles.restype = C.c_int64
        _debug_lib.agp_debug_shard_work_doubles.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int]
        _debug_lib.agp_debug_shard_factor_custom.restype = C.c_int
        _debug_lib.agp_debug_shard_factor_custom.argtypes = [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P, _P, _D,
                                                            C.POINTER(C.c_int64)]
        _debug_lib.agp_debug_comm_create_null.restype = C.c_int
        _debug_lib.agp_debug_comm_create_null.argtypes = [C.c_int, C.c_int, _PP]
    return _debug_lib


def load():
    """Load the HIP library.  There is no CPU fallback: a missing or
    unloadable library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). albatross_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, restype, argtypes in EXPORTS:
        fn = getattr(lib, name)  # AttributeError if a declared symbol is absent
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
