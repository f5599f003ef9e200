"""GPU: end-to-end learning sanity through the whole engine (forward_train, every backward kernel, Adam on the
reference-layout parameters, then the hipGraph sampler on the trained weights): a small model of each backbone
trained on random-phase sinusoids (200..800 Hz) must drive the epsilon-MSE from ~1 to < 0.25 within 200 steps
and then generate bounded signals whose dominant frequency lies in the band it was trained on."""
import math

import pytest
import torch
import torch.nn as nn

from tests import cases

pytestmark = pytest.mark.gpu

L = 2048
MODELS = {
    "wavenet": (cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=8, dilation_cycle=8), 2e-3),
    "sashimi": (cases.ss_cfg(d_model=32, n_layers=2, L=L, diffusion_step_embed_dim_mid=128), 1e-3),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_model_learns_sinusoids_and_samples_them(gpu, name):
    from diffwave_sashimi_amd.models import construct_model
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    from diffwave_sashimi_amd.training import training_loss
    cfg, lr = MODELS[name]
    torch.manual_seed(0)
    net = construct_model(dict(cfg)).to(gpu).train()
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    g = torch.Generator().manual_seed(1)
    t = torch.arange(L) / 16000.0

    def batch(B=8):
        f = 200 + 600 * torch.rand(B, 1, generator=g)
        ph = 2 * math.pi * torch.rand(B, 1, generator=g)
        return (0.5 * torch.sin(2 * math.pi * f * t[None] + ph)).unsqueeze(1).to(gpu)

    hist = []
    for _ in range(200):
        opt.zero_grad()
        loss = training_loss(net, nn.MSELoss(), batch(), dh, generator=g)
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
    first, last = sum(hist[:5]) / 5, sum(hist[-20:]) / 20
    assert first > 0.6 and last < 0.25, (first, last)
    net.eval()
    x = sampling(net, (4, 1, L), dh, seed=3)
    assert torch.isfinite(x).all() and 0.15 < float(x.std()) < 0.8 and float(x.abs().max()) < 3.0
    peak = torch.fft.rfft(x[:, 0].cpu(), dim=-1).abs()[:, 1:].argmax(dim=-1).add(1) * 16000.0 / L
    assert all(100.0 <= float(p) <= 1000.0 for p in peak), peak
