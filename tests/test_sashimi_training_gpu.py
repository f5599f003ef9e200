"""GPU: SaShiMi training path (forward_train + backward through the C ABI): the gradient of every
parameter of the `train.py:198-222` loss -- including the S4 kernel parameters C, B, P, inv_w_real,
w_imag, log_dt behind the Cauchy / Woodbury / FFT chain (`s4.py:704-807`) -- against torch autograd
through the CPU oracle on the same inputs."""
import pytest
import torch
import torch.nn as nn

from oracle import sashimi as oss
from tests import cases
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu

TRAIN_CASES = {
    # H = 32/64/128, L = 1024/256/64 (all through the M = 1024 fused FFT), unet skips on every level
    "d32": (cases.ss_cfg(d_model=32, n_layers=1, L=1024, diffusion_step_embed_dim_mid=64), 2),
    # snet: no blocks on the way down, skips only after UpPool; pool 2 and expand 2 with ff 1
    "snet": (cases.ss_cfg(d_model=32, n_layers=1, L=512, pool=[2, 2], ff=1, unet=False, diffusion_step_embed_dim_mid=64), 3),
    # BASELINE config 5 channel counts (H = 128/256/512), two blocks per level, L = 2048 (M = 2048 at the top)
    "d128": (cases.ss_cfg(d_model=128, n_layers=2, L=2048), 1),
}


def _run(name, gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B = TRAIN_CASES[name]
    L = cfg["L"]
    net = cases.build_ours(cfg, 15).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(19)) * 0.3
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, generator=torch.Generator().manual_seed(23))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    # the first forward ran _setup_C in place (s4.py:531-551), so this state_dict is what a checkpoint holds
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}

    def oracle_net(inp, mel_spec=None):
        return oss.sashimi_forward(sd, cfg, inp[0], inp[1], mel_spec=mel_spec)

    ref_loss = training_loss(oracle_net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(23))
    ref_loss.backward()
    return net, got, sd, float(loss), float(ref_loss)


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_sashimi_parameter_gradients_match_autograd(gpu, name):
    net, got, sd, loss, ref_loss = _run(name, gpu)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    gmax = max(float(sd[k].grad.abs().max()) for k in got if sd[k].grad is not None)
    worst, worst_k, bad = 0.0, None, []
    for k, gk in got.items():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert torch.isfinite(gk).all(), k
        # gradients that are mathematically zero (weight_v of a 1-input weight-normed conv) are rounding
        # noise on both sides: anything below 1e-5 of the model's largest gradient is compared absolutely
        scale = max(float(ref.abs().max()), 1e-5 * gmax)
        err = float((gk - ref).abs().max()) / scale
        if err > worst:
            worst, worst_k = err, k
        # fp32 chains of FFTs and a Cauchy sum on both sides: 5e-3 of the tensor's largest gradient
        if err >= 5e-3:
            bad.append(f"{k}: rel err {err:.3e} (|ref|max {float(ref.abs().max()):.3e}, |got|max {float(gk.abs().max()):.3e})")
    assert not bad, f"{name}: {len(bad)} of {len(got)} gradients off:\n" + "\n".join(bad[:40])
    print(f"{name}: worst parameter-gradient rel err {worst:.3e} ({worst_k})")


def test_sashimi_training_step_reduces_the_loss(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B = TRAIN_CASES["d32"]
    net = cases.build_ours(cfg, 16).to(gpu).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, cfg["L"], generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    losses = []
    for it in range(8):
        opt.zero_grad()
        loss = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0] * 0.9, losses
    net.eval()
    with torch.no_grad():
        out = net((audio, torch.zeros(B, 1, device=gpu)))
    assert torch.isfinite(out).all()


def test_conditional_sashimi_gradients_match_autograd(gpu):
    """Mel-conditional SaShiMi training: every block's conditioner (two weight-normed ConvTranspose2d upsamplers +
    `mel_conv`, pooled stages take the first L_stage upsampled frames, `sashimi.py:160-175`) gets gradients."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg = cases.ss_cfg(unconditional=False, d_model=32, n_layers=1, L=1024, mel_upsample=[16, 16],
                       diffusion_step_embed_dim_mid=64)
    B, L, Tmel = 2, 1024, 4
    net = cases.build_ours(cfg, 35).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(39)) * 0.3
    mel = torch.cat([cases.mel_inputs(1, Tmel, 41 + i) for i in range(B)])
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, mel_spec=mel.to(gpu), generator=torch.Generator().manual_seed(43))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    ref_loss = training_loss(lambda inp, mel_spec=None: oss.sashimi_forward(sd, cfg, inp[0], inp[1], mel_spec=mel_spec),
                             nn.MSELoss(), audio, dh, mel_spec=mel, generator=torch.Generator().manual_seed(43))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * max(1.0, abs(float(ref_loss)))
    gmax = max(float(sd[k].grad.abs().max()) for k in got if sd[k].grad is not None)
    bad, seen_cond = [], 0
    for k, gk in got.items():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        seen_cond += ("upsample_conv2d" in k or "mel_conv" in k) and float(ref.abs().max()) > 0
        scale = max(float(ref.abs().max()), 1e-5 * gmax)
        err = float((gk - ref).abs().max()) / scale
        if err >= 5e-3:
            bad.append(f"{k}: rel err {err:.3e} (|ref|max {float(ref.abs().max()):.3e}, |got|max {float(gk.abs().max()):.3e})")
    assert not bad, f"{len(bad)} of {len(got)} gradients off:\n" + "\n".join(bad[:30])
    assert seen_cond >= 9 * 5          # 5 blocks x (2 upsamplers x (bias, g, v) + mel_conv (bias, g, v))
