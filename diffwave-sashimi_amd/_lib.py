"""ctypes binding of libdws.so (include/dws.h).  There is no CPU fallback: if
the library is missing or fails to load, importing the engine raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DWS_LIB: another build of the same library (tools/ab_lib.sh times two builds back to back on one box); the product
# path is the in-tree libdws.so
LIB_PATH = os.environ.get("DWS_LIB") or os.path.join(HERE, "libdws.so")

DWS_OK, DWS_ERR_INVALID, DWS_ERR_UNSUPPORTED, DWS_ERR_HIP, DWS_ERR_STATE = 0, -1, -2, -3, -4
DWS_KIND_WAVENET, DWS_KIND_SASHIMI = 1, 2
DWS_MAX_POOL = 8

c_f32p = ctypes.c_void_p  # device pointers travel as integers


class ModelDesc(ctypes.Structure):
    """Mirror of ``dws_model_desc`` (include/dws.h); field names are the YAML keys."""
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("in_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32),
        ("diffusion_step_embed_dim_in", ctypes.c_int32),
        ("diffusion_step_embed_dim_mid", ctypes.c_int32),
        ("diffusion_step_embed_dim_out", ctypes.c_int32),
        ("unconditional", ctypes.c_int32),
        ("mel_upsample", ctypes.c_int32 * 2),
        ("mel_bands", ctypes.c_int32),
        ("res_channels", ctypes.c_int32), ("skip_channels", ctypes.c_int32),
        ("num_res_layers", ctypes.c_int32), ("dilation_cycle", ctypes.c_int32),
        ("d_model", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("n_pool", ctypes.c_int32),
        ("pool", ctypes.c_int32 * DWS_MAX_POOL), ("expand", ctypes.c_int32), ("ff", ctypes.c_int32),
        ("unet", ctypes.c_int32), ("L", ctypes.c_int32),
    ]


_SIGS = {
    "dws_last_error": (ctypes.c_char_p, []),
    "dws_abi_version": (ctypes.c_int, []),
    "dws_arch": (ctypes.c_char_p, []),
    "dws_cauchy_sym_fwd": (ctypes.c_int, [c_f32p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "dws_cauchy_sym_bwd": (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "dws_cauchy_fwd": (ctypes.c_int, [c_f32p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "dws_cauchy_bwd": (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "dws_model_create": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.POINTER(ctypes.c_void_p)]),
    "dws_model_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "dws_model_num_params": (ctypes.c_int, [ctypes.c_void_p]),
    "dws_model_param_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_int)]),
    "dws_model_set_param": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p]),
    "dws_model_update_params": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p),
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "dws_model_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]),
    "dws_model_commit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "dws_model_prepare": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]),
    "dws_model_set_condition": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_void_p]),
    "dws_model_forward": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "dws_model_forward_train": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "dws_model_backward": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "dws_model_get_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_f32p, ctypes.c_int64, ctypes.c_void_p]),
    "dws_model_get_grads": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p),
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "dws_model_set_grad_sinks": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                                ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]),
    "dws_model_grad_group_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "dws_model_grad_ready_seq": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.POINTER(ctypes.c_int32)]),
    "dws_model_read_tap": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_f32p, ctypes.c_int64, ctypes.c_void_p]),
    "dws_sampler_run": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.POINTER(ctypes.c_float),
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                       ctypes.c_int32, c_f32p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_void_p]),
    "dws_sampler_steps": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.POINTER(ctypes.c_float),
                                         ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64,
                                         ctypes.c_int32, ctypes.c_void_p]),
    "dws_mel_spectrogram": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_int64, c_f32p, c_f32p, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    "dws_gemm_bf16x6": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "dws_gemm_f16x3": (ctypes.c_int, [c_f32p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_float] * 2 + [ctypes.c_void_p]),
    "dws_profile_enable": (ctypes.c_int, [ctypes.c_char_p]),
    "dws_profile_query": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]),
    "dws_profile_query_each": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "dws_profile_disable": (ctypes.c_int, []),
}

# every symbol include/dws.h declares
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load libdws.so and bind every entry point; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP engine is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(status):
    """0 -> None; DWS_ERR_UNSUPPORTED -> NotImplementedError (`cauchy.py:72-77,95-101`);
    anything else -> RuntimeError (the reference's TORCH_CHECK failures, `cauchy.cpp:6-7`)."""
    if status == DWS_OK:
        return
    msg = load().dws_last_error().decode("utf-8", "replace")
    if status == DWS_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"libdws error {status}: {msg}")


def ptr(t):
    """Device/host pointer of a torch tensor as an integer (0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
