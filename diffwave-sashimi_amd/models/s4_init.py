"""Host-side, one-time S4 (NPLR / HiPPO-LegS) parameter math of the SaShiMi shim:
initialisation (``models/s4.py:251-406,1192-1227,607-641``), the FFT nodes
(``:553-571``) and the first-forward ``C -> C~`` transform (``:524-551``).
None of this is on the per-step path; it runs on the CPU with torch."""
import math

import numpy as np
import torch


def nplr_legs(N):
    """HiPPO-LegS in normal-plus-low-rank form (``s4.py:266-274,316-318,342-406``).

    ``A[i,j] = -sqrt((2i+1)(2j+1))`` (i>j), ``A[i,i] = -(i+1)``, ``B_i = sqrt(2i+1)``,
    rank-1 term ``P_i = sqrt(i + 1/2)``: ``A + P P^T = -1/2 I + S`` with S skew-symmetric,
    so ``eigh(-i (A + P P^T))`` (float64) gives the imaginary parts.  Returns the
    negative-imaginary half: ``w (N/2)``, ``P (1, N/2)``, ``B (N/2)`` as complex64."""
    q = torch.arange(N, dtype=torch.float64)
    r = torch.sqrt(2 * q + 1)
    A = -torch.tril(r[:, None] * r[None, :], -1) - torch.diag(q + 1)
    B = r.clone()
    P = torch.sqrt(0.5 + q)
    AP = A + P[:, None] * P[None, :]
    w_re = torch.mean(torch.diagonal(AP))
    w_im, V = torch.linalg.eigh(AP.to(torch.cdouble) * -1j)
    idx = torch.argsort(w_im)
    w_im, V = w_im[idx][: N // 2], V[:, idx][:, : N // 2]
    Vh = V.conj().T
    w = (w_re + 1j * w_im).to(torch.cfloat)
    Bv = (Vh @ B.to(torch.cdouble)).to(torch.cfloat)
    Pv = (Vh @ P.to(torch.cdouble)).to(torch.cfloat).unsqueeze(0)
    return w, Pv, Bv


def init_s4_params(H, N=64, dt_min=0.001, dt_max=0.1):
    """Random S4 parameters with the reference's distributions and storage
    conventions (``s4.py:1196-1218,631-641``): ``log_dt ~ U(ln dt_min, ln dt_max)``,
    ``C ~ CN(0,1)`` of shape (2, H, N/2) stored as the real view of conj(C);
    w/P/B tiled to H copies; ``inv_w_real = log(-clamp(Re w, max=-1e-3))``."""
    log_dt = torch.rand(H) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)
    w, P, B = nplr_legs(N)
    C = torch.randn(2, H, N // 2, dtype=torch.cfloat)
    Bt = B.unsqueeze(0).repeat(H, 1).unsqueeze(0)            # (1, H, N/2)
    Pt = P.unsqueeze(1).repeat(1, H, 1)                      # (1, H, N/2)
    wt = w.unsqueeze(0).repeat(H, 1)                         # (H, N/2)
    w_real = torch.clamp(wt.real, max=-1e-3)
    return dict(
        C=torch.view_as_real(C.conj().resolve_conj()).contiguous(),
        log_dt=log_dt,
        B=torch.view_as_real(Bt.contiguous()).contiguous(),
        P=torch.view_as_real(Pt.contiguous()).contiguous(),
        inv_w_real=torch.log(-w_real),
        w_imag=wt.imag.clone(),
    )


def omega_z(L):
    """FFT nodes computed with the reference's own expression (``s4.py:561-565``):
    complex64 ``pow`` of a float64-derived base.  They are rounding-sensitive (the
    Nyquist node has 1 + omega ~ 0), so the engine takes them from here instead
    of recomputing them with "better" arithmetic."""
    omega = torch.tensor(np.exp(-2j * np.pi / L), dtype=torch.cfloat)
    omega = omega ** torch.arange(0, L // 2 + 1)
    z = 2 * (1 - omega) / (1 + omega)
    return omega, z


def setup_C(C, P, inv_w_real, w_imag, log_dt, L):
    """First-forward transform ``C~ = C (I - dA^L)`` (``s4.py:524-551``) for a kernel
    whose ``L`` buffer is still 0.  Dense float64 evaluation of the bilinear
    discretisation ``dA = (2/dt I - A)^-1 (2/dt I + A)``, ``A = diag(w) - p q^T`` over the
    conjugate-extended state (p = [P, conj P], q = [conj P, P]); the first N columns
    are kept.  Inputs are the stored real views; returns the new stored ``C``."""
    Cc = torch.view_as_complex(C.detach().cpu().contiguous()).to(torch.cdouble)
    Pc = torch.view_as_complex(P.detach().cpu().contiguous())[0].to(torch.cdouble)
    N = Cc.shape[-1]
    dt = torch.exp(log_dt.detach().cpu().double())
    w = -torch.exp(inv_w_real.detach().cpu().double()) + 1j * w_imag.detach().cpu().double()
    wf = torch.cat([w, w.conj()], -1)
    pf = torch.cat([Pc, Pc.conj()], -1)
    qf = torch.cat([Pc.conj(), Pc], -1)
    A = torch.diag_embed(wf) - pf.unsqueeze(-1) * qf.unsqueeze(-2)
    I = torch.eye(2 * N, dtype=torch.cdouble)
    s = (2.0 / dt).to(torch.cdouble)[:, None, None]
    dA = torch.linalg.solve(s * I - A, s * I + A)
    dA_L = torch.linalg.matrix_power(dA, int(L))
    Cf = torch.cat([Cc, Cc.conj()], -1)
    Ct = (Cf - torch.einsum("chn,hnm->chm", Cf, dA_L))[..., :N].to(torch.cfloat)
    return torch.view_as_real(Ct).contiguous()
