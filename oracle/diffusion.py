"""Oracle: diffusion-step embedding, schedule tables and the reverse sampler.

Test infrastructure (see ``oracle/__init__.py``).  Restates
``models/utils.py:4-29``, ``utils.py:121-151`` and ``generate.py:23-55``.
"""
import numpy as np
import torch


def calc_diffusion_step_embedding(diffusion_steps, dim_in=128):
    """``models/utils.py:20-27`` without the hard-coded ``.cuda()`` (``:24``).

    ``diffusion_steps`` is ``[B,1]`` float32 (sampling, ``generate.py:50``) or
    int64 (training, ``train.py:218``).  The scalar ``ln(1e4)/(half-1)`` is a
    float64 numpy value; ``arange * -scalar`` is float32 (int64 tensor times a
    python float), so the frequency table is float32.
    """
    assert dim_in % 2 == 0
    half = dim_in // 2
    _embed = np.log(10000) / (half - 1)
    _embed = torch.exp(torch.arange(half) * -_embed)
    _embed = diffusion_steps * _embed
    return torch.cat((torch.sin(_embed), torch.cos(_embed)), 1)


def calc_diffusion_hyperparams(T, beta_0, beta_T, beta=None, fast=False):
    """``utils.py:121-151``: fp32 tables with the reference's sequential
    in-place recurrences (``:144-146``) so the rounding order is identical."""
    if fast and beta is not None:
        Beta = torch.tensor(beta)
        T = len(beta)
    else:
        Beta = torch.linspace(beta_0, beta_T, T)
    Alpha = 1 - Beta
    Alpha_bar = Alpha + 0
    Beta_tilde = Beta + 0
    for t in range(1, T):
        Alpha_bar[t] *= Alpha_bar[t - 1]
        Beta_tilde[t] *= (1 - Alpha_bar[t - 1]) / (1 - Alpha_bar[t])
    Sigma = torch.sqrt(Beta_tilde)
    return {"T": T, "Beta": Beta, "Alpha": Alpha, "Alpha_bar": Alpha_bar, "Sigma": Sigma}


def sampling(net, size, dh, condition=None, x_T=None, noise=None):
    """``generate.py:23-55`` with *injected* noise.

    ``net((x, t), mel_spec=condition) -> eps``.  ``x_T`` is the initial state
    (``generate.py:47``) and ``noise[t]`` the variance term added after step
    ``t`` (``generate.py:54``; ``noise[0]`` is never used).  When they are
    ``None`` they are drawn from torch's CPU generator exactly as the reference
    does, so a seeded call reproduces a seeded reference call.
    """
    T, Alpha, Alpha_bar, Sigma = dh["T"], dh["Alpha"], dh["Alpha_bar"], dh["Sigma"]
    assert len(Alpha) == T and len(Alpha_bar) == T and len(Sigma) == T and len(size) == 3
    x = torch.normal(0, 1, size=size) if x_T is None else x_T.clone()
    with torch.no_grad():
        for t in range(T - 1, -1, -1):
            diffusion_steps = t * torch.ones((size[0], 1))
            eps = net((x, diffusion_steps), mel_spec=condition)
            x = (x - (1 - Alpha[t]) / torch.sqrt(1 - Alpha_bar[t]) * eps) / torch.sqrt(Alpha[t])
            if t > 0:
                z = torch.normal(0, 1, size=size) if noise is None else noise[t]
                x = x + Sigma[t] * z
    return x
