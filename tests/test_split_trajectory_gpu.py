"""GPU: whole reverse-diffusion trajectories (`generate.py:49-54`) through the fp32-equivalent split arithmetic
(`precision="bf16x6"`: 3-term bf16 split, six products, fp32 accumulate) -- the arithmetic the bench times over T = 200
steps must also be SAFE over T = 200 steps: rounding differences of 1e-6 per forward are fed back through the update
T times.  With injected noise (same x_T, same z_t on every path):

  * x_0 of the split path vs x_0 of the exact-f32 path: <= 1e-4 relative
  * both vs the CPU oracle's loop (`oracle/diffusion.py: sampling`, the reference's op order): <= 1e-3

on `wn_c128` (T = 200, the SC09 schedule of BASELINE configs 1-3) and on the mel-conditional `ss_cond_d32` (T = 50, config
4's schedule and channel widths).  Also here: switching the precision between two forwards that are handed the SAME mel
tensor must not drop the conditioner (advisor finding, round 5)."""
import pytest
import torch

from oracle import diffusion as odiff
from oracle import sashimi as oss
from oracle import wavenet as own
from tests import cases
from tests.conftest import REL_TOL, rel_err

pytestmark = pytest.mark.gpu


def _trajectories(gpu, net, oracle_net, size, dh, mel=None, seed=5):
    from diffwave_sashimi_amd.sampling import sampling
    T = dh["T"]
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(*size, generator=g)
    noise = torch.randn(T, *size, generator=g)
    cond = None if mel is None else mel.to(gpu)
    out = {}
    for prec in ("f32", "bf16x6"):
        net.set_option("precision", prec)
        out[prec] = sampling(net, size, dh, condition=cond, x_T=x_T, noise=noise, use_graph=True).cpu()
        assert torch.isfinite(out[prec]).all()
    net.set_option("precision", "f32")
    ref = odiff.sampling(oracle_net, size, dh, condition=mel, x_T=x_T, noise=noise)
    return out, ref


def test_wavenet_T200_trajectory_split_vs_f32_vs_oracle(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c128"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    dh = calc_diffusion_hyperparams(200, 1e-4, 0.02)                     # configs/experiment/sc09.yaml: T=200, 1e-4 .. 0.02
    onet = lambda inp, mel_spec=None: own.wavenet_forward(sd, cfg, inp[0], inp[1])
    out, ref = _trajectories(gpu, net, onet, (B, 1, L), dh)
    e_split_f32 = rel_err(out["bf16x6"], out["f32"])
    e_f32, e_split = rel_err(out["f32"], ref), rel_err(out["bf16x6"], ref)
    print(f"wn_c128 T=200: x_0 bf16x6 vs f32 {e_split_f32:.3e}; vs the oracle loop: f32 {e_f32:.3e}, bf16x6 {e_split:.3e}")
    assert not torch.equal(out["bf16x6"], out["f32"])
    assert e_split_f32 < 1e-4
    assert e_f32 < REL_TOL and e_split < REL_TOL


def test_sashimi_cond_T50_trajectory_split_vs_f32_vs_oracle(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    cfg, B, Tmel, wseed, iseed, _ = cases.SASHIMI_COND_CASES["ss_cond_d32"]
    L = cfg["L"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    mel = cases.mel_inputs(B, Tmel, iseed)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)                      # configs/experiment/ljspeech.yaml: T=50, 1e-4 .. 0.05
    onet = lambda inp, mel_spec=None: oss.sashimi_forward(sd, cfg, inp[0], inp[1], mel_spec=mel_spec)
    out, ref = _trajectories(gpu, net, onet, (B, 1, L), dh, mel=mel)
    e_split_f32 = rel_err(out["bf16x6"], out["f32"])
    e_f32, e_split = rel_err(out["f32"], ref), rel_err(out["bf16x6"], ref)
    print(f"ss_cond_d32 T=50: x_0 bf16x6 vs f32 {e_split_f32:.3e}; vs the oracle loop: f32 {e_f32:.3e}, bf16x6 {e_split:.3e}")
    assert not torch.equal(out["bf16x6"], out["f32"])
    assert e_split_f32 < 1e-4
    assert e_f32 < REL_TOL and e_split < REL_TOL


@pytest.mark.parametrize("backbone", ["wavenet", "sashimi"])
def test_precision_switch_keeps_the_conditioner_of_the_same_mel_tensor(gpu, backbone):
    """net(x, mel); net.set_option(...); net(x, mel) with the SAME mel tensor object: the option marks the engine dirty,
    its commit drops the installed conditioner terms -- the module must hand the mel over again instead of trusting its
    cache (it used to run the second forward unconditionally, silently)."""
    if backbone == "wavenet":
        cfg, B, L, Tmel, wseed, iseed, _ = cases.WAVENET_COND_CASES["wn_cond_c64"]
    else:
        cfg, B, Tmel, wseed, iseed, _ = cases.SASHIMI_COND_CASES["ss_cond_d32"]
        L = cfg["L"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    audio, steps = audio.to(gpu), steps.to(gpu)
    mel = cases.mel_inputs(B, Tmel, iseed).to(gpu)
    with torch.no_grad():
        first = net((audio, steps), mel_spec=mel)
        nomel = net((audio, steps))
        back = net((audio, steps), mel_spec=mel)
        net.set_option("precision", "bf16x6")
        split = net((audio, steps), mel_spec=mel)                 # same tensor object, unchanged version counter
        net.set_option("precision", "f32")
        again = net((audio, steps), mel_spec=mel)
    assert torch.equal(back, first) and torch.equal(again, first)
    assert rel_err(nomel, first) > 1e-2                           # the conditioner matters for this case
    assert rel_err(split, first) < 1e-5 and not torch.equal(split, first)


def test_config2_network_T200_trajectory_at_full_length(gpu):
    """BASELINE config 2's own network (wnet_h256_d36) over its own schedule (T = 200) at its own length (L = 16000), one clip with
    injected noise: x_0 of the split path within 1e-4 of the exact-f32 path's, and both within 1e-3 of the CPU oracle's loop
    (200 oracle forwards of the 36-layer network: ~2 minutes of host time)."""
    import bench
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    cfg = bench.CONFIGS["wnet_h256_d36_T200"]
    mcfg, L = dict(cfg["model"]), cfg["L"]
    net = cases.build_ours(mcfg, 91).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    dh = calc_diffusion_hyperparams(**cfg["diffusion"])
    onet = lambda inp, mel_spec=None: own.wavenet_forward(sd, mcfg, inp[0], inp[1])
    out, ref = _trajectories(gpu, net, onet, (1, 1, L), dh, seed=11)
    e_split_f32 = rel_err(out["bf16x6"], out["f32"])
    e_f32, e_split = rel_err(out["f32"], ref), rel_err(out["bf16x6"], ref)
    print(f"wnet_h256_d36 T=200 L={L}: x_0 bf16x6 vs f32 {e_split_f32:.3e}; vs the oracle loop: f32 {e_f32:.3e}, bf16x6 {e_split:.3e}")
    assert not torch.equal(out["bf16x6"], out["f32"])
    assert e_split_f32 < 1e-4 and e_f32 < REL_TOL and e_split < REL_TOL
