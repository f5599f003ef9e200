"""Oracle: Cauchy multiply (``extensions/cauchy``).  Test infrastructure only.

The contract is the CUDA extension's semantics (``cauchy_cuda.cu:331``):
``out[b,l] = sum_n v/(z-w) + conj(v)/(z-conj(w))`` over the HALF state -- not the
fork's pure-torch fallback ``cauchy_naive`` (``models/s4.py:109-116``) which
drops the conjugate half (semantic trap 1)."""
import torch


def cauchy_sym_formula(v_half, z, w_half):
    """The reference's own known-answer formula (``extensions/cauchy/cauchy.py:19-26``,
    used in fp64 by ``test_cauchy.py:66``), restated for half-state inputs.
    v_half, w_half: (B, N/2) complex; z: (L) complex.  Returns (B, L)."""
    vv = v_half.unsqueeze(1)          # b 1 n
    zz = z.unsqueeze(-1)              # l 1
    ww = w_half.unsqueeze(1)
    return 2 * ((zz * vv.real - vv.real * ww.real - vv.imag * ww.imag)
                / (zz * zz - 2 * zz * ww.real + ww.abs().square())).sum(dim=-1)


def cauchy_sym_direct(v_half, z, w_half):
    """The extension's expression evaluated term by term (``cauchy_cuda.cu:331``)."""
    d1 = z.unsqueeze(-1) - w_half.unsqueeze(1)
    d2 = z.unsqueeze(-1) - w_half.conj().unsqueeze(1)
    return (v_half.unsqueeze(1) / d1 + v_half.conj().unsqueeze(1) / d2).sum(dim=-1)


def cauchy_direct(v, z, w):
    """Non-symmetric: ``out[b,l] = sum_n v/(z-w)`` (``cauchy.py:17``, ``cauchy_cuda.cu:44-115``)."""
    return (v.unsqueeze(1) / (z.unsqueeze(-1) - w.unsqueeze(1))).sum(dim=-1)


def cauchy_sym_bwd(v_half, z, w_half, dout):
    """dv, dw of the symmetric kernel as the extension defines them
    (``cauchy_cuda.cu:420-447``): the gradient w.r.t. the HALF inputs in torch's
    conjugate-gradient convention."""
    wc = w_half.conj().unsqueeze(1)                   # b 1 n
    den1 = z.conj().unsqueeze(-1) - wc                # conj(z) - conj(w)
    den2 = z.unsqueeze(-1) - wc                       # z - conj(w)
    t1 = dout.unsqueeze(-1) / den1
    t2 = dout.conj().unsqueeze(-1) / den2
    dv = (t1 + t2).sum(dim=1)
    dw = (t1 / den1 + t2 / den2).sum(dim=1) * v_half.conj()
    return dv, dw


def cauchy_bwd(v, z, w, dout):
    """Non-symmetric backward (``cauchy_cuda.cu:178-208``)."""
    q = 1.0 / (z.unsqueeze(-1) - w.unsqueeze(1)).conj()
    p = dout.unsqueeze(-1) * q
    return p.sum(dim=1), (p * q).sum(dim=1) * v.conj()


def generate_data(batch_size, N, L, symmetric=True, seed=2357):
    """The reference test's distribution (``extensions/cauchy/test_cauchy.py:11-23,60-61``),
    drawn on the CPU generator.  Returns half-state (v, z, w) when symmetric."""
    g = torch.Generator().manual_seed(seed)
    if not symmetric:
        v = torch.randn(batch_size, N, dtype=torch.complex64, generator=g)
        w = torch.randn(batch_size, N, dtype=torch.complex64, generator=g)
        z = torch.randn(L, dtype=torch.complex64, generator=g)
        return v, z, w
    assert N % 2 == 0
    v_half = torch.randn(batch_size, N // 2, dtype=torch.complex64, generator=g)
    w_half = torch.randn(batch_size, N // 2, dtype=torch.complex64, generator=g)
    z = torch.exp(1j * torch.randn(L, dtype=torch.float32, generator=g))
    return v_half, z, w_half
