"""GPU: WaveNet training path (forward_train + backward through the C ABI) -- parameter gradients of
the `train.py:198-222` loss against torch autograd through the CPU oracle on the same inputs."""
import pytest
import torch
import torch.nn as nn

from oracle import wavenet as own
from tests import cases
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu

TRAIN_CASES = {
    # generic forward kernels (C = 16), dilations 1, 2, 4, 8 with L = 50 (taps fall off both ends)
    "tiny": (cases.wn_cfg(res_channels=16, skip_channels=16, num_res_layers=4, dilation_cycle=4), 3, 50),
    # MFMA forward (C = 64) + backward kernels, L not a multiple of the tile
    "c64": (cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=3, dilation_cycle=3), 2, 200),
    # different res/skip widths
    # MFMA adjoints (tapconv / wgrad kernels, 1 M-tile per wave), different res/skip widths
    "c128_s256": (cases.wn_cfg(res_channels=128, skip_channels=256, num_res_layers=2, dilation_cycle=2), 1, 130),
    # MFMA adjoints with 2 M-tiles per wave, dilations up to 64 > tile, ragged L, B > 1
    "c256": (cases.wn_cfg(res_channels=256, skip_channels=256, num_res_layers=7, dilation_cycle=7), 2, 333),
}


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_wavenet_parameter_gradients_match_autograd(gpu, name):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B, L = TRAIN_CASES[name]
    net = cases.build_ours(cfg, 5).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    g = torch.Generator().manual_seed(9)
    audio = torch.randn(B, 1, L, generator=g) * 0.3
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, generator=torch.Generator().manual_seed(21))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}

    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}

    def oracle_net(inp, mel_spec=None):
        return own.wavenet_forward(sd, cfg, inp[0], inp[1], mel_spec=mel_spec)

    ref_loss = training_loss(oracle_net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(21))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for k, gk in got.items():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        scale = max(float(ref.abs().max()), 1e-6)
        err = float((gk - ref).abs().max()) / scale
        # gradients that are ~0 relative to the largest gradient in the model are compared absolutely
        assert err < 2e-3 or float((gk - ref).abs().max()) < 1e-7, f"{name}: grad of {k}: rel err {err:.3e}"
        worst = max(worst, err if scale > 1e-5 else 0.0)
    print(f"{name}: worst parameter-gradient rel err {worst:.3e}")


def test_training_step_reduces_the_loss_and_eval_path_still_works(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B, L = TRAIN_CASES["c64"]
    net = cases.build_ours(cfg, 6).to(gpu).train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)      # `train.py:91` (lr 2e-4 there)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    losses = []
    for it in range(8):
        opt.zero_grad()
        loss = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] * 0.9, losses
    net.eval()
    with torch.no_grad():
        out = net((audio, torch.zeros(B, 1, device=gpu)))
    assert torch.isfinite(out).all()


COND_TRAIN_CASES = {
    # mel [B, 80, 2] -> 512 upsampled frames >= L = 500 (ragged: the truncation `[:, :, :L]` is exercised)
    "cond_c64": (cases.wn_cfg(unconditional=False, res_channels=64, skip_channels=64, num_res_layers=3, dilation_cycle=3,
                              mel_upsample=[16, 16]), 2, 500, 2),
    "cond_c128": (cases.wn_cfg(unconditional=False, res_channels=128, skip_channels=128, num_res_layers=2, dilation_cycle=2,
                               mel_upsample=[16, 16]), 1, 768, 3),
}


@pytest.mark.parametrize("name", list(COND_TRAIN_CASES))
def test_conditional_wavenet_gradients_match_autograd(gpu, name):
    """Mel-conditional training (`train.py:121-128,221`): gradients of every parameter including each layer's
    upsampler (`upsample_conv2d.{0,1}`, weight-normed ConvTranspose2d) and `mel_conv`."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B, L, Tmel = COND_TRAIN_CASES[name]
    net = cases.build_ours(cfg, 25).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(29)) * 0.3
    mel = torch.cat([cases.mel_inputs(1, Tmel, 31 + i) for i in range(B)])          # one mel per clip
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, mel_spec=mel.to(gpu), generator=torch.Generator().manual_seed(33))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}

    def oracle_net(inp, mel_spec=None):
        return own.wavenet_forward(sd, cfg, inp[0], inp[1], mel_spec=mel_spec)

    ref_loss = training_loss(oracle_net, nn.MSELoss(), audio, dh, mel_spec=mel, generator=torch.Generator().manual_seed(33))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    gmax = max(float(sd[k].grad.abs().max()) for k in got if sd[k].grad is not None)
    bad, seen_cond = [], 0
    for k, gk in got.items():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        seen_cond += ("upsample_conv2d" in k or "mel_conv" in k) and float(ref.abs().max()) > 0
        scale = max(float(ref.abs().max()), 1e-5 * gmax)
        err = float((gk - ref).abs().max()) / scale
        if err >= 2e-3:
            bad.append(f"{k}: rel err {err:.3e} (|ref|max {float(ref.abs().max()):.3e}, |got|max {float(gk.abs().max()):.3e})")
    assert not bad, f"{name}: {len(bad)} of {len(got)} gradients off:\n" + "\n".join(bad[:30])
    assert seen_cond >= 9 * cfg["num_res_layers"]        # every conditioner tensor of every layer carries gradient
