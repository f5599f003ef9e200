"""Drop-in for the reference's Cauchy operator: the pybind module ``cauchy_mult``
(``extensions/cauchy/cauchy.cpp:86-95``) and its Python wrapper
(``extensions/cauchy/cauchy.py:46-111``), backed by the gfx950 kernels in
libdws.so (``dws_cauchy_*`` in include/dws.h).

  cauchy_mult_fwd(v[B,N], z[L], w[B,N]) -> [B,L]             (complex64, CUDA/HIP tensors)
  cauchy_mult_bwd(v, z, w, dout[B,L])   -> (dv[B,N], dw[B,N])
  cauchy_mult_sym_fwd / cauchy_mult_sym_bwd: same shapes, N = half state
  cauchy_mult(v, z, w, symmetric=True): broadcasting front end with autograd
"""
import torch

from .. import _lib


def _chk(name, t, shape=None):
    # the reference's CHECK_DEVICE / TORCH_CHECK (`cauchy.cpp:6-7,58-64`) -> RuntimeError
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on CUDA")
    if t.dtype != torch.complex64:
        raise RuntimeError(f"{name} must be complex64 (the reference kernels are hard-wired to c10::complex<float>)")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


def _fwd(fn_name, v, z, w):
    if v.dim() != 2 or z.dim() != 1:
        raise RuntimeError("v must be [B, N] and z must be [L]")
    B, N = v.shape
    L = z.shape[0]
    v = _chk("v", v)
    z = _chk("z", z, (L,))
    w = _chk("w", w, (B, N))
    out = torch.empty((B, L), dtype=v.dtype, device=v.device)
    fn = getattr(_lib.load(), fn_name)
    _lib.check(fn(v.data_ptr(), z.data_ptr(), w.data_ptr(), out.data_ptr(), B, N, L, _lib.current_stream()))
    return out


def _bwd(fn_name, v, z, w, dout):
    if v.dim() != 2 or z.dim() != 1:
        raise RuntimeError("v must be [B, N] and z must be [L]")
    B, N = v.shape
    L = z.shape[0]
    v = _chk("v", v)
    z = _chk("z", z, (L,))
    w = _chk("w", w, (B, N))
    dout = _chk("dout", dout, (B, L))
    dv = torch.empty((B, N), dtype=v.dtype, device=v.device)
    dw = torch.empty((B, N), dtype=v.dtype, device=v.device)
    fn = getattr(_lib.load(), fn_name)
    _lib.check(fn(v.data_ptr(), z.data_ptr(), w.data_ptr(), dout.data_ptr(), dv.data_ptr(), dw.data_ptr(),
                  B, N, L, _lib.current_stream()))
    return dv, dw


def cauchy_mult_fwd(v, z, w):
    return _fwd("dws_cauchy_fwd", v, z, w)


def cauchy_mult_bwd(v, z, w, dout):
    return _bwd("dws_cauchy_bwd", v, z, w, dout)


def cauchy_mult_sym_fwd(v, z, w):
    return _fwd("dws_cauchy_sym_fwd", v, z, w)


def cauchy_mult_sym_bwd(v, z, w, dout):
    return _bwd("dws_cauchy_sym_bwd", v, z, w, dout)


class CauchyMultiply(torch.autograd.Function):
    """``extensions/cauchy/cauchy.py:66-87``.  The reference only instantiates
    N == 64 and L % 32 == 0 for this (unused) variant; the HIP kernel has no such
    restriction, only N <= 1024."""

    @staticmethod
    def forward(ctx, v, z, w):
        if not (v.is_cuda and z.is_cuda and w.is_cuda):
            raise NotImplementedError("Only support CUDA tensors")
        ctx.save_for_backward(v, z, w)
        return cauchy_mult_fwd(v, z, w)

    @staticmethod
    def backward(ctx, dout):
        v, z, w = ctx.saved_tensors
        dv, dw = cauchy_mult_bwd(v, z, w, dout)
        return dv, None, dw


class CauchyMultiplySymmetric(torch.autograd.Function):
    """``extensions/cauchy/cauchy.py:90-111``."""

    @staticmethod
    def forward(ctx, v, z, w):
        L = z.shape[-1]
        max_L_value = 32 * 1024 * 64 * 1024
        if L > max_L_value:
            raise NotImplementedError(f"Only support L values <= {max_L_value}")
        if not (v.is_cuda and z.is_cuda and w.is_cuda):
            raise NotImplementedError("Only support CUDA tensors")
        ctx.save_for_backward(v, z, w)
        return cauchy_mult_sym_fwd(v, z, w)

    @staticmethod
    def backward(ctx, dout):
        v, z, w = ctx.saved_tensors
        dv, dw = cauchy_mult_sym_bwd(v, z, w, dout)
        return dv, None, dw


def _cauchy_mult(v, z, w, symmetric=True):
    if not symmetric:
        return CauchyMultiply.apply(v, z, w)
    return CauchyMultiplySymmetric.apply(v, z, w)


def cauchy_mult(v, z, w, symmetric=True):
    """Shape front end of ``extensions/cauchy/cauchy.py:46-63``: broadcast ``v``
    and ``w``, squeeze ``z`` to 1-D, flatten the leading dims, reshape back."""
    v, w = torch.broadcast_tensors(v, w)
    shape = v.shape
    z = z.squeeze()
    assert len(z.shape) == 1
    v = v.contiguous()
    w = w.contiguous()
    z = z.contiguous()
    N = v.size(-1)
    assert w.size(-1) == N
    y = _cauchy_mult(v.view(-1, N), z, w.view(-1, N), symmetric=symmetric)
    return y.view(*shape[:-1], z.size(-1))
