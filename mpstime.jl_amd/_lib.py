"""ctypes binding of libmpstime_hip.so (include/mpstime_hip.h).

The HIP library is the product path: if it is missing or fails to load this
module raises - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpstime_hip.so")

MPST_OK, MPST_ERR_INVALID, MPST_ERR_UNSUPPORTED, MPST_ERR_DEVICE, MPST_ERR_SVD, MPST_ERR_NOMEM = 0, -1, -2, -3, -4, -5
LOSS = {"KLD": 0, "MSE": 1}
OPT = {"TSGO": 0, "GD": 1}
F64, F32, C128, C64 = 0, 1, 2, 3       # mpst_set_dataset's dtype (include/mpstime_hip.h)
ABI_VERSION = 2
MAX_SPECTRUM = 512
KERNEL_CLASSES = ("yhat", "grad", "grad_reduce+update", "gram", "eig_tri", "split", "env", "bt_assemble", "allreduce",
                  "eig_vec", "eig_fin")


class mpst_options(C.Structure):
    _fields_ = [("chi_max", C.c_int32), ("update_iters", C.c_int32), ("loss", C.c_int32), ("optimiser", C.c_int32),
                ("rescale_before", C.c_int32), ("rescale_after", C.c_int32), ("train_classes_separately", C.c_int32),
                ("svd_alg", C.c_int32), ("rebuild_caches", C.c_int32), ("track_cost", C.c_int32),
                ("eta", C.c_double), ("cutoff", C.c_double)]


class mpst_sweep_stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("svd_status", C.c_int32), ("max_chi", C.c_int32),
                ("eig_sweeps_total", C.c_int32), ("eig_fallbacks", C.c_int32)]


class mpst_bond_debug(C.Structure):
    _fields_ = [("loss", C.c_double), ("grad_norm", C.c_double), ("bt_norm", C.c_double), ("chi_new", C.c_int32),
                ("n_spectrum", C.c_int32), ("eig_sweeps", C.c_int32), ("reserved", C.c_int32),
                ("spectrum", C.c_double * MAX_SPECTRUM)]


class mpst_encode_opts(C.Structure):
    _fields_ = [("basis", C.c_int32), ("sigmoid_transform", C.c_int32), ("minmax", C.c_int32), ("is_test", C.c_int32),
                ("rescale_out_of_bounds", C.c_int32), ("fit_sigmoid", C.c_int32),
                ("median", C.c_double), ("iqr", C.c_double), ("lo", C.c_double), ("hi", C.c_double),
                ("data_lb", C.c_double), ("data_ub", C.c_double), ("range_a", C.c_double), ("range_b", C.c_double)]


BASIS = {"Legendre_Norm": 0, "Legendre_No_Norm": 1, "Fourier": 2, "Stoudenmire": 3, "Sahand": 4, "Uniform": 5}      # canonical names (options.jl:243-279; :Legendre == :Legendre_No_Norm)

# every symbol include/mpstime_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _dp = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
class ImputeOpts(C.Structure):          # mpst_impute_opts
    _fields_ = [("method", C.c_int32), ("order", C.c_int32), ("get_err", C.c_int32), ("max_trials", C.c_int32),
                ("mean_basis", C.c_int32), ("reserved", C.c_int32), ("rejection_threshold", C.c_double)]


class ImputeModel(C.Structure):         # mpst_impute_model
    _fields_ = [("N", C.c_int64), ("T", C.c_int32), ("d", C.c_int32), ("C", C.c_int32), ("label_site", C.c_int32),
                ("dtype", C.c_int32), ("compute", C.c_int32), ("site", C.POINTER(C.c_void_p)), ("chi", C.POINTER(C.c_int32)),
                ("phi", C.c_void_p), ("label_idx", C.POINTER(C.c_int32))]


SYMBOLS = {
    "mpst_version": (C.c_int, []),
    "mpst_last_error": (C.c_char_p, [_vp]),
    "mpst_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "mpst_destroy": (None, [_vp]),
    "mpst_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "mpst_comm_init": (C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "mpst_comm_library": (C.c_int, [C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "mpst_comm_ipc_export": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    "mpst_comm_ipc_attach": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "mpst_comm_select": (C.c_int, [_vp, C.c_int]),
    "mpst_set_dtype": (C.c_int, [_vp, _i32]),
    "mpst_set_dataset": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(_i32), _i64, _i32, _i32, _i32, _i32, C.POINTER(_i64)]),
    "mpst_encode_dataset": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(_i32), _i64, _i32, _i32, _i32, C.POINTER(mpst_encode_opts),
                                      C.POINTER(_i64), _dp, _dp]),
    "mpst_encode_values": (C.c_int, [_vp, _dp, _i64, _i32, _i32, C.POINTER(mpst_encode_opts), _vp, _dp, _dp]),
    "mpst_get_encoded": (C.c_int, [_vp, C.c_int, _dp]),
    "mpst_set_options": (C.c_int, [_vp, C.POINTER(mpst_options)]),
    "mpst_set_mps": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i32), _i32, _i32]),
    "mpst_get_chi": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "mpst_get_mps": (C.c_int, [_vp, C.POINTER(_vp)]),
    "mpst_build_caches": (C.c_int, [_vp]),
    "mpst_sweep": (C.c_int, [_vp, C.POINTER(mpst_sweep_stats)]),
    "mpst_sweep_batch": (C.c_int, [C.POINTER(_vp), _i32, C.POINTER(mpst_sweep_stats)]),
    "mpst_sweep_batch_multi": (C.c_int, [C.POINTER(_vp), _i32, C.POINTER(C.c_int32), C.POINTER(mpst_sweep_stats)]),
    "mpst_set_batch_hint": (C.c_int, [_vp, _i32]),
    "mpst_get_loss_trace": (C.c_int, [_vp, _dp]),
    "mpst_bond_step": (C.c_int, [_vp, _i32, _i32, C.POINTER(mpst_bond_debug)]),
    "mpst_eval": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.POINTER(_i64)]),
    "mpst_classify": (C.c_int, [_vp, C.c_int, C.POINTER(_i32), _dp]),
    "mpst_normalize": (C.c_int, [_vp]),
    "mpst_impute": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint8), _dp, _dp, _i32, C.POINTER(ImputeOpts), _dp, _dp, _dp, _dp]),
    "mpst_impute_model_run": (C.c_int, [_vp, C.POINTER(ImputeModel), C.POINTER(C.c_uint8), _dp, _vp, _i32, C.POINTER(ImputeOpts), _dp, _dp,
                                        _dp, _dp]),
    "mpst_get_impute_phases": (C.c_int, [_vp, _dp]),
    "mpst_get_impute_info": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "mpst_selftest_mfma": (C.c_int, [_vp, _dp, _dp, _i32, _dp]),
    "mpst_selftest_eig": (C.c_int, [_vp, _dp, _i32, _i32, _dp, _dp, C.POINTER(_i32)]),
    "mpst_set_profile": (C.c_int, [_vp, C.c_uint32]),
    "mpst_get_profile": (C.c_int, [_vp, _dp, C.POINTER(_i64)]),
    "mpst_get_eig_phases": (C.c_int, [_vp, _dp]),
    "mpst_get_tail_phases": (C.c_int, [_vp, _dp]),
    "mpst_get_info": (C.c_int, [_vp, C.POINTER(_i32)]),
    "mpst_get_info_n": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
}

_lib = None


def load():
    """Load libmpstime_hip.so and bind every declared symbol.  Raises if the
    library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``
    or ``make -C mpstime.jl_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP sweep engine has not been built "
                           "(make -C mpstime.jl_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mpst_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI version {lib.mpst_version()}, this binding was written for {ABI_VERSION}: rebuild "
                           "(make -C mpstime.jl_amd/csrc)")
    _lib = lib
    return lib


class MPSTError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[mpst {code}] {msg}")
        self.code = code


class SVDError(MPSTError):
    """Bond-tensor decomposition failed - the failure class `tune` retries on
    (src/Training/hyperparameters/tuning.jl:73-86)."""
