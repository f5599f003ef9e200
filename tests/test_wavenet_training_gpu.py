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


def _engine_and_oracle(cfg, B, L, gpu, wseed, aseed, gseed, mel=None, start=0, tries=1):
    """Engine gradients, the oracle's fp32 autograd and its FLOAT64 autograd (rounding-noise yardstick,
    tests/gradcheck.py) on the same weights, audio, steps and noise."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    from tests import gradcheck
    net = cases.build_ours(cfg, wseed)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    # inputs away from every ReLU kink (tests/gradcheck.py: smooth_case), so that the plain 1e-3 bound applies
    audio, gseed, loss_of, truth, kink, tried = gradcheck.smooth_case(cfg, sd, dh, B, L, mel, aseed, gseed, start=start, tries=tries)
    print(f"inputs: try {tried} (audio seed {aseed + 1000 * tried}), largest kink noise {max(kink.values()):.1e}")
    net = net.to(gpu).train()
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, mel_spec=None if mel is None else mel.to(gpu),
                         generator=torch.Generator().manual_seed(gseed))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    loss32, o32 = gradcheck.oracle_grads(cfg, sd, loss_of, torch.float32)
    return got, {k: o32[k] for k in got}, {k: truth[k] for k in got}, float(loss), loss32, kink


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_wavenet_parameter_gradients_match_autograd(gpu, name):
    """Per tensor 1e-3 of its largest gradient (widened to 3x the measured fp32 noise where that is larger)."""
    from tests import gradcheck
    cfg, B, L = TRAIN_CASES[name]
    # first try of tests/gradcheck.smooth_case whose inputs sit furthest from the ReLU kinks (found once, six tries each)
    start = {"c128_s256": 4, "c256": 1}.get(name, 0)
    got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, L, gpu, 5, 9, 21, start=start)
    assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    worst, k = gradcheck.compare(got, o32, truth, label=name, kink=kink)
    print(f"{name}: worst parameter-gradient rel err {worst:.3e} ({k})")


AB_CASES = {
    # 2 M-tiles per wave (C = 256), dilations 1..64, ragged L, B > 1
    "c256": (TRAIN_CASES["c256"][0], 2, 333),
    # the same with L % 4 == 0: the 16-byte LDS-DMA staging of the Winograd data gradient at d >= 4
    "c256_x4": (TRAIN_CASES["c256"][0], 2, 336),
    # 1 M-tile per wave (C = 128), L over several 64-column pair tiles
    "c128_s256": (TRAIN_CASES["c128_s256"][0], 1, 1030),
    # a full dilation cycle: d = 1 .. 2048 with d > L in the last layers, positions past L in the last pair block
    "cycle12": (cases.wn_cfg(res_channels=128, skip_channels=128, num_res_layers=12, dilation_cycle=12), 1, 600),
}


@pytest.mark.parametrize("name", list(AB_CASES))
def test_winograd_and_direct_training_paths_agree(gpu, name):
    """`conv_algo` selects the form of the dilated conv in the training forward AND of its data gradient
    (`csrc/wavenet_backward_wino.hip`: Winograd F(2,3) along the dilation stride; `direct`: three shifted taps).  Same
    weights, same batch: every parameter gradient must agree far inside the 1e-3 gate."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B, L = AB_CASES[name]
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(77))
    grads = {}
    for algo in ("winograd", "direct"):
        net = cases.build_ours(cfg, 5).to(gpu).train()
        net.set_option("conv_algo", algo)
        loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, generator=torch.Generator().manual_seed(3))
        loss.backward()
        grads[algo] = ({k: p.grad.detach().cpu() for k, p in net.named_parameters()}, float(loss))
    (gw, lw), (gd, ld) = grads["winograd"], grads["direct"]
    assert abs(lw - ld) < 1e-5 * max(1.0, abs(ld))
    # (init_conv's weight_v has ONE input channel: its weight-norm gradient is analytically zero, what is left is 1e-10
    # of rounding -- tensors that far below the others carry no signal to compare)
    top = max(float(g.abs().max()) for g in gd.values())
    worst = max((rel_err(gw[k], gd[k]), k) for k in gd if float(gd[k].abs().max()) > 1e-6 * top)
    print(f"{name}: largest winograd-vs-direct gradient difference {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 2e-4, worst
    assert any(not torch.equal(gw[k], gd[k]) for k in gd)          # two different arithmetic paths


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
    from tests import gradcheck
    cfg, B, L, Tmel = COND_TRAIN_CASES[name]
    mel = torch.cat([cases.mel_inputs(1, Tmel, 31 + i) for i in range(B)])          # one mel per clip
    got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, L, gpu, 25, 29, 33, mel=mel, start=4)
    assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    worst, k = gradcheck.compare(got, o32, truth, label=name, kink=kink)
    seen_cond = sum(("upsample_conv2d" in k or "mel_conv" in k) and float(v.abs().max()) > 0 for k, v in o32.items())
    assert seen_cond >= 9 * cfg["num_res_layers"]        # every conditioner tensor of every layer carries gradient
    print(f"{name}: worst parameter-gradient rel err {worst:.3e} ({k})")
