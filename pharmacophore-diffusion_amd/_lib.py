"""ctypes binding of csrc/libpfdyn.so (C ABI: include/pfdyn.h).

The shared library is the product; this module only marshals pointers.  It fails loudly when
the library has not been built -- there is no Python / CPU fallback for the compute path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFDYN_LIB selects an alternative build of the same library (A/B kernel experiments)
LIB_PATH = os.environ.get("PFDYN_LIB") or os.path.join(_HERE, "csrc", "libpfdyn.so")

PF_ABI_VERSION = 1
PF_NORM_MEAN, PF_NORM_VALUE, PF_NORM_GRAPH = 0, 1, 2


class PfConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("pharm_nf", ctypes.c_int32), ("rec_nf", ctypes.c_int32),
        ("vector_size", ctypes.c_int32), ("n_hidden_scalars", ctypes.c_int32), ("n_convs", ctypes.c_int32),
        ("n_message_gvps", ctypes.c_int32), ("n_update_gvps", ctypes.c_int32), ("n_noise_gvps", ctypes.c_int32),
        ("message_norm_mode", ctypes.c_int32), ("message_norm_value", ctypes.c_float),
        ("ff_k", ctypes.c_int32), ("pf_k", ctypes.c_int32),
        ("cutoff_pp", ctypes.c_float), ("cutoff_pf", ctypes.c_float), ("cutoff_fp", ctypes.c_float),
        ("cutoff_ff", ctypes.c_float), ("rbf_dmax", ctypes.c_float), ("rbf_dim", ctypes.c_int32),
    ]


class PfStepCoef(ctypes.Structure):
    _fields_ = [("t", ctypes.c_float), ("alpha_t_given_s", ctypes.c_float), ("var_terms", ctypes.c_float),
                ("sigma", ctypes.c_float), ("ep_zt", ctypes.c_float), ("ep_pred", ctypes.c_float)]


# every symbol include/pfdyn.h declares: (name, restype, argtypes)
_P, _I32, _I64, _F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
SYMBOLS = {
    "pf_version": (ctypes.c_char_p, []),
    "pf_last_error": (ctypes.c_char_p, [_P]),
    "pf_create": (ctypes.c_int, [ctypes.POINTER(PfConfig), ctypes.POINTER(_P)]),
    "pf_destroy": (None, [_P]),
    "pf_set_weight": (ctypes.c_int, [_P, ctypes.c_char_p, _P, _I32, ctypes.POINTER(_I64)]),
    "pf_commit_weights": (ctypes.c_int, [_P]),
    "pf_set_pocket_batch": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "pf_set_pocket_batch_host": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "pf_set_pocket_groups": (ctypes.c_int, [_P, _I32, _P]),
    "pf_declare_onehot_features": (ctypes.c_int, [_P, _I32]),
    "pf_build_pp_edges": (_I64, [_P, _I32, _P, _P, _I32, _P, _P, _I64, _P]),
    "pf_dynamics_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "pf_sample_begin": (ctypes.c_int, [_P, _P, _P, _P]),
    "pf_denoise_step": (ctypes.c_int, [_P, ctypes.POINTER(PfStepCoef), _P, _I32, _I32, _P]),
    "pf_sample_end": (ctypes.c_int, [_P, _F, _P, _P, _P]),
    "pf_sample_status": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "pf_prepare_timesteps": (ctypes.c_int, [_P, _P, _I32, _P]),
    "pf_sample_frame": (ctypes.c_int, [_P, _F, _P, _P, _P]),
    "pf_sample": (ctypes.c_int, [_P, _I32, ctypes.POINTER(PfStepCoef), _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _P]),
    "pf_param_count": (ctypes.c_int, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I32)]),
    "pf_param_layout": (ctypes.c_int, [_P, _I32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "pf_train_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _F, ctypes.c_uint32, _P, _P, _P]),
    "pf_train_backward": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "pf_train_loss_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_float, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_float, ctypes.c_uint32, _P, _P]),
    "pf_train_loss_backward": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "pf_train_loss_backward_out": (ctypes.c_int, [_P, _P, _P, _P]),
    "pf_set_flat_params": (ctypes.c_int, [_P, _P, _P]),
    "pf_get_flat_params": (ctypes.c_int, [_P, _P, _P]),
    "pf_adam_step": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _P]),
    "pf_train_set_precision": (ctypes.c_int, [_P, _I32]),
    "pf_train_get_precision": (ctypes.c_int, [_P, ctypes.POINTER(_I32)]),
    "pf_debug_set_dropout_masks": (ctypes.c_int, [_P, _P]),
    "pf_debug_dropout_mask": (ctypes.c_int, [_P, _I32, _I32, _F, ctypes.c_uint32, _P, _P]),
    "pf_debug_get_edges": (_I64, [_P, _I32, _P, _P, _I64, _P]),
    "pf_debug_conv_layer": (ctypes.c_int, [_P, _I32] + [_P] * 11),
    "pf_profile_enable": (ctypes.c_int, [_P, ctypes.c_uint32]),
    "pf_profile_read": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64), _P]),
    "pf_profile_read_train": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64), _P]),
    "pf_debug_work": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(_I64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64), _P]),
    "pf_debug_counts": (ctypes.c_int, [_P, ctypes.POINTER(_I64), _P]),
    "pf_debug_kernel_family": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    "pf_debug_last_eps": (ctypes.c_int, [_P, _P, _P, _P]),
    "pf_debug_xchg_timeouts": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "pf_debug_xchg_fault": (ctypes.c_int, [_P, _I32, _I32]),
    "pf_debug_ahead": (ctypes.c_int, [_P, ctypes.POINTER(_I64), _P]),
    "pf_debug_l0_hoist": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "pf_debug_chain": (ctypes.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libpfdyn.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pharmacophore-diffusion_amd/csrc`). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PfError(RuntimeError):
    pass


def check(lib, handle, rc, what):
    if rc < 0:
        msg = lib.pf_last_error(handle)
        raise PfError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
