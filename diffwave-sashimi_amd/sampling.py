"""Reverse-diffusion driver with the reference's surface (``generate.py:23-55``,
``utils.py:121-151``) on top of ``dws_sampler_run`` (one step captured as a
hipGraph and replayed T times; schedule, step index and RNG on the device)."""
import ctypes

import numpy as np
import torch

from . import _lib


def calc_diffusion_hyperparams(T, beta_0, beta_T, beta=None, fast=False):
    """``utils.py:121-151``: fp32 tables, sequential in-place recurrences.  All
    tables stay on the host (the engine uploads what it needs once)."""
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


def _host_table(t):
    a = np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def sampling(net, size, diffusion_hyperparams, condition=None, *, x_T=None, noise=None, seed=None,
             use_graph=True):
    """``x_0 = sampling(net, (B, C, L), dh, condition)`` as in ``generate.py:23-55``.

    Extra keyword-only arguments (not in the reference):
      x_T    initial state [B,C,L]; default: drawn on the device from the Philox stream
      noise  injected variance noise [T,B,C,L] (``noise[t]`` is added after step t>0) -- parity mode
      seed   Philox seed for the on-device RNG (default: a fresh draw from torch's CPU generator per call, so
             successive unseeded calls differ -- as the reference's do -- and ``torch.manual_seed`` still governs)
    """
    dh = diffusion_hyperparams
    T, Alpha, Alpha_bar, Sigma = dh["T"], dh["Alpha"], dh["Alpha_bar"], dh["Sigma"]
    assert len(Alpha) == T and len(Alpha_bar) == T and len(Sigma) == T and len(size) == 3
    B, C, L = size
    lib = _lib.load()
    dev = torch.device("cuda")
    if seed is None:
        seed = int(torch.randint(0, 2 ** 63 - 1, (1,), dtype=torch.int64).item())
    with torch.no_grad():
        net._train_generation += 1     # the sampler's forwards overwrite the activations of a pending training forward
        net._sync_params(L)
        net._prepare(B, L)
        net._set_condition(condition)
        if x_T is None:
            x = torch.empty(size, device=dev, dtype=torch.float32)
            init = 1
        else:
            x = x_T.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
            init = 0
        nz = None
        if noise is not None:
            nz = noise.detach().to(device=dev, dtype=torch.float32).contiguous()
            assert tuple(nz.shape) == (T, B, C, L)
        a, pa = _host_table(Alpha)
        ab, pab = _host_table(Alpha_bar)
        sg, psg = _host_table(Sigma)
        _lib.check(lib.dws_sampler_run(net._handle, x.data_ptr(), pa, pab, psg, T, _lib.ptr(nz), seed, init,
                                       1 if use_graph else 0, _lib.current_stream()))
        torch.cuda.current_stream().synchronize()  # nz / tables must outlive the enqueued work
    return x
