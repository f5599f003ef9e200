"""``cauchy_mult`` -- the module name the reference imports (`extensions/cauchy/cauchy.py:5`:
``from cauchy_mult import cauchy_mult_fwd, cauchy_mult_bwd, cauchy_mult_sym_fwd, cauchy_mult_sym_bwd``; `models/s4.py:35-42`
keys ``has_cauchy_extension`` on that import), here as a ctypes binding of libdws.so instead of the pybind/CUDA
extension `extensions/cauchy/cauchy.cpp:86-95`.

Standalone on purpose (ctypes + torch only, no package-relative import): copy or symlink this one file next to the
reference's `extensions/cauchy/cauchy.py` (or put this directory on ``sys.path``) and `models/s4.py` picks the HIP
kernels up unchanged.  The library is found through ``$DWS_LIB`` or next to this package (``../libdws.so``).

Same contract as the reference's entry points: complex64 CUDA/HIP tensors, ``v, w [B, N]``, ``z [L]``; outputs are
freshly allocated on ``v``'s device; work is enqueued on the current stream, no synchronisation; shape / device / dtype
violations raise RuntimeError (the reference's TORCH_CHECK, `cauchy.cpp:6-7,58-64`), sizes the kernels are not built
for raise NotImplementedError (`cauchy.py:72-77,95-101`).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DWS_LIB") or os.path.join(_HERE, "..", "libdws.so")
if not os.path.exists(_LIB_PATH):
    # ImportError is what `models/s4.py:35-42` catches to fall back (and warn) -- there is no silent CPU path here
    raise ImportError(f"cauchy_mult: {_LIB_PATH} not found (build libdws.so or set DWS_LIB)")
_lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
_lib.dws_last_error.restype = ctypes.c_char_p
_P, _I = ctypes.c_void_p, ctypes.c_int64
for _n, _k in (("dws_cauchy_fwd", 4), ("dws_cauchy_sym_fwd", 4), ("dws_cauchy_bwd", 6), ("dws_cauchy_sym_bwd", 6)):
    getattr(_lib, _n).argtypes = [_P] * _k + [_I, _I, _I, _P]
    getattr(_lib, _n).restype = ctypes.c_int


def _status(st):
    if st == -2:                                   # DWS_ERR_UNSUPPORTED
        raise NotImplementedError(_lib.dws_last_error().decode("utf-8", "replace"))
    if st != 0:
        raise RuntimeError(_lib.dws_last_error().decode("utf-8", "replace"))


def _arg(name, t, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.complex64:
        raise RuntimeError(f"{name} must be complex64")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fwd(fn, v, z, w):
    if v.dim() != 2 or z.dim() != 1:
        raise RuntimeError("v must be [B, N] and z must be [L]")
    (B, N), L = v.shape, z.shape[0]
    v, z, w = _arg("v", v), _arg("z", z), _arg("w", w, (B, N))
    out = torch.empty((B, L), dtype=v.dtype, device=v.device)
    _status(fn(v.data_ptr(), z.data_ptr(), w.data_ptr(), out.data_ptr(), B, N, L, _stream()))
    return out


def _bwd(fn, v, z, w, dout):
    if v.dim() != 2 or z.dim() != 1:
        raise RuntimeError("v must be [B, N] and z must be [L]")
    (B, N), L = v.shape, z.shape[0]
    v, z, w, dout = _arg("v", v), _arg("z", z), _arg("w", w, (B, N)), _arg("dout", dout, (B, L))
    dv, dw = torch.empty_like(v), torch.empty_like(w)
    _status(fn(v.data_ptr(), z.data_ptr(), w.data_ptr(), dout.data_ptr(), dv.data_ptr(), dw.data_ptr(), B, N, L, _stream()))
    return dv, dw


def cauchy_mult_fwd(v, z, w):                      # `cauchy.cpp:25-36`
    return _fwd(_lib.dws_cauchy_fwd, v, z, w)


def cauchy_mult_bwd(v, z, w, dout):                # `cauchy.cpp:38-53`
    return _bwd(_lib.dws_cauchy_bwd, v, z, w, dout)


def cauchy_mult_sym_fwd(v, z, w):                  # `cauchy.cpp:55-66`
    return _fwd(_lib.dws_cauchy_sym_fwd, v, z, w)


def cauchy_mult_sym_bwd(v, z, w, dout):            # `cauchy.cpp:68-82`
    return _bwd(_lib.dws_cauchy_sym_bwd, v, z, w, dout)
