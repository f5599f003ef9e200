"""Oracle: SaShiMi backbone + S4 layer (``models/sashimi.py``, ``models/s4.py``) as
functional torch-CPU ops over a reference-layout ``state_dict``.  Test
infrastructure only.

Semantics are the CUDA-extension ones: the Cauchy step sums BOTH conjugate
halves (``cauchy_cuda.cu:331``), unlike this fork's ``cauchy_naive`` fallback
(``s4.py:109-116``) -- SURVEY.md 8c trap 1.  Like the reference, the S4
convolution kernel is regenerated in every layer on every call (no hoisting),
which is what makes this file usable as the reference-equivalent CPU baseline.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .cauchy import cauchy_sym_direct
from .wavenet import mel_upsample, step_embedding_mlp, wn_conv1d

_r2c = torch.view_as_complex


def transposed_ln(x, m, s):
    """``TransposedLN.forward`` (``models/sashimi.py:17-20``): population std, no eps."""
    sd, mu = torch.std_mean(x, dim=-2, unbiased=False, keepdim=True)
    return (s / sd) * (x - mu + m)


def down_pool(sd, prefix, x, p):
    """``DownPool.forward`` (``sashimi.py:36-39``): ``... h (l s) -> ... (h s) l`` then 1x1 conv."""
    B, H, L = x.shape
    x = x.reshape(B, H, L // p, p).permute(0, 1, 3, 2).reshape(B, H * p, L // p)
    return wn_conv1d(sd, prefix + ".linear.conv", x)


def up_pool(sd, prefix, x, p):
    """``UpPool.forward`` (``sashimi.py:54-58``): 1x1 conv then ``... (h s) l -> ... h (l s)``."""
    x = wn_conv1d(sd, prefix + ".linear.conv", x)
    B, HP, L = x.shape
    return x.reshape(B, HP // p, p, L).permute(0, 1, 3, 2).reshape(B, HP // p, L * p)


def ff(sd, prefix, x):
    """``FF.forward`` (``sashimi.py:60-75``): 1x1 -> GELU(erf) -> 1x1."""
    h = F.gelu(wn_conv1d(sd, prefix + ".ff.0.conv", x))
    return wn_conv1d(sd, prefix + ".ff.2.conv", h)


def omega_z(L, dtype=torch.cfloat):
    """FFT nodes exactly as ``SSKernelNPLR._omega`` computes them (``s4.py:561-565``):
    a complex64 ``pow`` of a float64-derived base -- rounding-sensitive."""
    omega = torch.tensor(np.exp(-2j * np.pi / L), dtype=dtype)
    omega = omega ** torch.arange(0, L // 2 + 1)
    z = 2 * (1 - omega) / (1 + omega)
    return omega, z


def setup_C(C, B, P, inv_w_real, w_imag, log_dt, L):
    """``SSKernelNPLR._setup_C`` for a fresh kernel (buffer ``L == 0``; ``s4.py:524-551``):
    ``C~ = C (I - dA^L)`` over the conjugate-extended state, first N columns kept.
    Restated with dense matrices in complex128 (the reference goes through the
    O(N) DPLR step of ``_setup_linear``/``_step_state_linear``, ``s4.py:815-904``):
    ``dA = (2/dt I - A)^-1 (2/dt I + A)``, ``A = diag(w) - p q^T`` with
    ``p = [P, conj P]``, ``q = [conj P, P]``.
    All inputs are the stored tensors (complex ones as complex64)."""
    N = C.shape[-1]
    dt = torch.exp(log_dt.double())                                   # (H)
    w = (-torch.exp(inv_w_real.double()) + 1j * w_imag.double())      # (H, N)
    wf = torch.cat([w, w.conj()], -1)                                 # (H, 2N)
    Pc = P[0].to(torch.cdouble)
    pf = torch.cat([Pc, Pc.conj()], -1)
    qf = torch.cat([Pc.conj(), Pc], -1)
    A = torch.diag_embed(wf) - pf.unsqueeze(-1) * qf.unsqueeze(-2)    # (H, 2N, 2N)
    I = torch.eye(2 * N, dtype=torch.cdouble)
    s = (2.0 / dt).to(torch.cdouble)[:, None, None]
    dA = torch.linalg.solve(s * I - A, s * I + A)
    dA_L = torch.linalg.matrix_power(dA, L)
    Cf = torch.cat([C.to(torch.cdouble), C.to(torch.cdouble).conj()], -1)   # (c, H, 2N)
    prod = torch.einsum("chn,hnm->chm", Cf, dA_L)
    return (Cf - prod)[..., :N].to(torch.cfloat)


def ss_kernel_nplr(sd, prefix, L):
    """``SSKernelNPLR.forward`` (``s4.py:674-807``) with ``rate=1``, ``state=None``,
    ``rank=1``: returns ``k`` of shape (2, H, L).  ``prefix`` names the
    ``...layer.kernel.kernel`` module.  Honours the ``L`` buffer: 0 means the
    stored ``C`` has not been through ``_setup_C`` yet (``s4.py:686-687``)."""
    C = _r2c(sd[prefix + ".C"].contiguous())
    Bp = _r2c(sd[prefix + ".B"].contiguous())
    P = _r2c(sd[prefix + ".P"].contiguous())
    inv_w_real, w_imag, log_dt = sd[prefix + ".inv_w_real"], sd[prefix + ".w_imag"], sd[prefix + ".log_dt"]
    Lk = int(sd[prefix + ".L"])          # the kernel's own length; L <= Lk is the number of taps asked for
    if Lk == 0:
        C = setup_C(C, Bp, P, inv_w_real, w_imag, log_dt, L)
        Lk = L
    assert Lk >= L, "S4.forward never asks for more taps than the kernel's length (s4.py:1387)"
    dt = torch.exp(log_dt)
    Q = P.conj()
    w = -torch.exp(inv_w_real) + 1j * w_imag
    omega, z = omega_z(Lk)                  # complex64 nodes, as the reference computes them
    if w.dtype != omega.dtype:              # float64 state_dict (conditioning studies): the SAME nodes, widened
        omega, z = omega.to(w.dtype), z.to(w.dtype)
    w = w * dt.unsqueeze(-1)
    Bc = torch.cat([Bp, P], dim=-3)           # (2, H, N)
    Cc = torch.cat([C, Q], dim=-3)            # (3, H, N)
    v = Bc.unsqueeze(-3) * Cc.unsqueeze(-4)   # (2, 3, H, N)
    H, N = w.shape
    wb = w.expand(2, 3, H, N).reshape(-1, N)
    r = cauchy_sym_direct(v.reshape(-1, N), z, wb).reshape(2, 3, H, -1)
    r = r * dt[None, None, :, None]
    k_f = r[:-1, :-1] - r[:-1, -1:] * r[-1:, :-1] / (1 + r[-1:, -1:])
    k_f = k_f * 2 / (1 + omega)
    k = torch.fft.irfft(k_f, n=Lk)[..., :L]   # generated at the kernel's length, truncated to the run (s4.py:805)
    return k[-1]                              # (2, H, L)


def s4_forward(sd, prefix, u):
    """``S4.forward`` (``s4.py:1376-1437``) as configured by ``DiffWaveBlock``
    (``sashimi.py:126``): bidirectional, channels=1, gelu, glu, transposed."""
    L = u.size(-1)
    l_max = int(sd[prefix + ".kernel.kernel.L"]) or L           # S4.L == l_max == the kernel's length after _setup_C
    Lt = min(L, l_max)                                           # `L_kernel` (s4.py:1387): longer inputs keep l_max taps
    k = ss_kernel_nplr(sd, prefix + ".kernel.kernel", Lt)
    k0, k1 = k[0:1], k[1:2]
    kk = F.pad(k0, (0, L)) + F.pad(k1.flip(-1), (L, 0))           # (1, H, Lt + L)
    k_f = torch.fft.rfft(kk, n=Lt + L)
    u_f = torch.fft.rfft(u, n=Lt + L)
    y = torch.fft.irfft(u_f * k_f, n=Lt + L)[..., :L]
    y = y + u * sd[prefix + ".D"].unsqueeze(-1)                   # D: (1, H)
    y = F.gelu(y)
    y = F.conv1d(y, sd[prefix + ".output_linear.0.weight"], sd[prefix + ".output_linear.0.bias"])
    return F.glu(y, dim=-2)


def diffwave_block(sd, prefix, x, emb, mel_spec=None):
    """``DiffWaveBlock.forward`` (``sashimi.py:143-184``)."""
    B, C, L = x.shape
    y = transposed_ln(x, sd[prefix + ".norm1.m"], sd[prefix + ".norm1.s"])
    part_t = F.linear(emb, sd[prefix + ".fc_t.weight"], sd[prefix + ".fc_t.bias"])
    y = y + part_t.unsqueeze(-1)
    y = s4_forward(sd, prefix + ".layer", y)
    if mel_spec is not None:
        m = mel_upsample(sd, prefix, mel_spec, L)
        y = y + wn_conv1d(sd, prefix + ".mel_conv.conv", m)
    y = x + y
    x = y
    y = transposed_ln(y, sd[prefix + ".norm2.m"], sd[prefix + ".norm2.s"])
    y = ff(sd, prefix + ".ff", y)
    return x + y


def layer_plan(cfg):
    """Kinds and (H, L) of d/c/u layers exactly as ``Sashimi.__init__`` builds
    them (``sashimi.py:236-268``)."""
    H, L = cfg["d_model"], cfg["L"]
    pool, expand, n_layers, unet = list(cfg["pool"]), cfg["expand"], cfg["n_layers"], cfg.get("unet", True)
    d, c, u = [], [], []
    for p in pool:
        if unet:
            d += [("block", H, L)] * n_layers
        d.append(("down", H, L, p))
        L //= p
        H *= expand
    c += [("block", H, L)] * n_layers
    for p in pool[::-1]:
        H //= expand
        L *= p
        u.append(("up", H * expand, L // p, p))
        u += [("block", H, L)] * n_layers
    return d, c, u


def sashimi_forward(sd, cfg, audio, diffusion_steps, mel_spec=None, return_pre_final=False):
    """``Sashimi.forward`` (``sashimi.py:277-313``)."""
    unet = cfg.get("unet", True)
    d_layers, c_layers, u_layers = layer_plan(cfg)
    x = F.relu(wn_conv1d(sd, "init_conv.0.conv", audio))
    emb = step_embedding_mlp(sd, "", diffusion_steps, cfg.get("diffusion_step_embed_dim_in", 128))

    def run(kind, prefix, x):
        if kind[0] == "block":
            return diffwave_block(sd, prefix, x, emb, mel_spec=mel_spec)
        if kind[0] == "down":
            return down_pool(sd, prefix, x, kind[3])
        return up_pool(sd, prefix, x, kind[3])

    outputs = []
    for i, kind in enumerate(d_layers):
        outputs.append(x)
        x = run(kind, f"d_layers.{i}", x)
    outputs.append(x)
    for i, kind in enumerate(c_layers):
        x = run(kind, f"c_layers.{i}", x)
    x = x + outputs.pop()
    for i, kind in enumerate(u_layers):
        x = run(kind, f"u_layers.{i}", x)
        if kind[0] == "up" or unet:
            x = x + outputs.pop()
    x = transposed_ln(x, sd["norm.m"], sd["norm.s"])
    y = F.relu(wn_conv1d(sd, "final_conv.0.conv", x))
    out = F.conv1d(y, sd["final_conv.2.conv.weight"], sd["final_conv.2.conv.bias"])
    if return_pre_final:
        return out, y
    return out


class SashimiOracle:
    """Callable with the reference's ``net((audio, t), mel_spec=None)`` surface."""

    def __init__(self, sd, cfg):
        self.sd = {k: v.detach().clone() for k, v in sd.items()}
        self.cfg = dict(cfg)

    def __call__(self, input_data, mel_spec=None):
        audio, steps = input_data
        with torch.no_grad():
            return sashimi_forward(self.sd, self.cfg, audio, steps, mel_spec=mel_spec)
