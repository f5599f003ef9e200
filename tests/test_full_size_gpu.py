"""GPU: BASELINE.json's full sizes (B = 16, L = 16000) through size-independent properties: a clip's result does
not depend on its batch neighbours (bitwise), equal inputs give equal outputs, the graph sampler is deterministic,
plus direct parity of ONE clip against the CPU oracle at the full length."""
import pytest
import torch

import bench
from oracle import sashimi as osa
from oracle import wavenet as own
from tests import cases
from tests.conftest import REL_TOL, rel_err

pytestmark = pytest.mark.gpu


def _inputs(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(B, 1, L, generator=g)
    steps = torch.randint(0, 200, (B, 1), generator=g).float()
    audio[7], steps[7] = audio[2], steps[2]          # two identical clips inside the batch
    return audio, steps


def _f64_one_clip(cfg, net, audio, steps, mel=None):
    """float64 evaluation of the oracle graph on ONE clip of the config at its full length: the yardstick both GPU
    arithmetics (exact-f32 MFMA, 3-term bf16 split) are measured against."""
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in net.state_dict().items()}
    with torch.no_grad():
        if cfg["model"]["_name_"] == "wavenet":
            return own.wavenet_forward(sd64, cfg["model"], audio.double(), steps)
        return osa.sashimi_forward(sd64, cfg["model"], audio.double(), steps, mel_spec=None if mel is None else mel.double())


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
@pytest.mark.parametrize("config", ["wnet_h256_d36_T200", "unet_d64_n6_T200"])
def test_config_at_full_size(gpu, config, precision):
    """BASELINE configs 2 and 3 at B = 16, L = 16000, under the exact-f32 path and under the fp32-equivalent 3-term bf16
    split the bench also times at this size (`precision="bf16x6"`): bitwise batch independence / determinism, one clip
    against the fp32 oracle at the full length, and -- for the split -- that clip's error against a float64 evaluation
    within 2x the exact-f32 path's."""
    cfg = bench.CONFIGS[config]
    B, L = cfg["B"], cfg["L"]
    assert (B, L) == (16, 16000)
    torch.manual_seed(3)
    net = cases.build_ours(dict(cfg["model"]), 91).to(gpu)
    net.set_option("precision", precision)
    audio, steps = _inputs(B, L, 92)
    with torch.no_grad():
        full = net((audio.to(gpu), steps.to(gpu)))
        again = net((audio.to(gpu), steps.to(gpu)))
        one = net((audio[3:4].to(gpu), steps[3:4].to(gpu)))
    assert torch.isfinite(full).all() and float(full.abs().max()) > 1e-3
    assert torch.equal(full, again)                              # deterministic
    assert torch.equal(full[2], full[7])                         # equal clips -> equal results, wherever they sit
    assert torch.equal(full[3:4], one)                           # independent of the batch neighbours, bitwise
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}

    def oracle32():
        with torch.no_grad():
            if cfg["model"]["_name_"] == "wavenet":
                return own.wavenet_forward(sd, cfg["model"], audio[3:4], steps[3:4])
            return osa.sashimi_forward(sd, cfg["model"], audio[3:4], steps[3:4])
    ref = cases.cached(("full_size_f32", config), oracle32)       # same seeds under both precisions
    assert rel_err(one.cpu(), ref) < REL_TOL
    if precision != "f32":
        ref64 = cases.cached(("full_size_f64", config), lambda: _f64_one_clip(cfg, net, audio[3:4], steps[3:4]))
        with torch.no_grad():
            net.set_option("precision", "f32")
            one32 = net((audio[3:4].to(gpu), steps[3:4].to(gpu)))
        e_split, e_f32 = rel_err(one.cpu(), ref64), rel_err(one32.cpu(), ref64)
        print(f"{config} clip 3 at L={L}: max-rel error vs float64 f32-MFMA {e_f32:.3e} | {precision} {e_split:.3e}; "
              f"the two GPU paths directly {rel_err(one, one32):.3e}")
        assert not torch.equal(one, one32)                       # the split path really ran
        assert e_split <= 2.0 * e_f32, (e_split, e_f32)


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_config4_at_full_size(gpu, precision):
    """BASELINE config 4 at its workload: unet_d32_n6 mel-conditional, B = 32, L = 16000, mel [32, 80, 63].  Batch
    independence / determinism (bitwise), a mel row conditions only its own clip, a batch-1 mel broadcasts
    (`generate.py:140,155`), one clip against the CPU oracle at the full length, and the T = 50 graph sampler -- under the
    exact-f32 tails and under the 3-term bf16 split (then also: error against float64 within 2x the f32 path's)."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg = bench.CONFIGS["unet_d32_n6_T50_cond"]
    B, L, Tmel = cfg["B"], cfg["L"], cfg["Tmel"]
    assert (B, L, Tmel) == (32, 16000, 63)
    net = cases.build_ours(dict(cfg["model"]), 95).to(gpu)
    net.set_option("precision", precision)
    audio, steps = _inputs(B, L, 96)
    steps = steps.clamp(max=49.0)
    mel = torch.cat([cases.mel_inputs(1, Tmel, 300 + i) for i in range(B)])
    mel[7] = mel[2]
    with torch.no_grad():
        full = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel.to(gpu))
        again = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel.to(gpu))
        one = net((audio[3:4].to(gpu), steps[3:4].to(gpu)), mel_spec=mel[3:4].to(gpu))
        bcast = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel[3:4].to(gpu))
        other = mel.clone()
        other[5] = mel[6]
        changed = net((audio.to(gpu), steps.to(gpu)), mel_spec=other.to(gpu))
    assert torch.isfinite(full).all() and float(full.abs().max()) > 1e-3
    assert torch.equal(full, again) and torch.equal(full[2], full[7]) and torch.equal(full[3:4], one)
    assert torch.equal(bcast[3], full[3]) and not torch.equal(bcast[4], full[4])
    keep = [i for i in range(B) if i != 5]
    assert torch.equal(changed[keep], full[keep]) and not torch.equal(changed[5], full[5])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}

    def oracle32():
        with torch.no_grad():
            return osa.sashimi_forward(sd, cfg["model"], audio[3:4], steps[3:4], mel_spec=mel[3:4])
    ref = cases.cached(("full_size_f32", "c4"), oracle32)
    assert rel_err(one.cpu(), ref) < REL_TOL
    if precision != "f32":
        ref64 = cases.cached(("full_size_f64", "c4"), lambda: _f64_one_clip(cfg, net, audio[3:4], steps[3:4], mel[3:4]))
        with torch.no_grad():
            net.set_option("precision", "f32")
            one32 = net((audio[3:4].to(gpu), steps[3:4].to(gpu)), mel_spec=mel[3:4].to(gpu))
            net.set_option("precision", precision)
        e_split, e_f32 = rel_err(one.cpu(), ref64), rel_err(one32.cpu(), ref64)
        print(f"config 4 clip 3: max-rel error vs float64 f32-MFMA {e_f32:.3e} | {precision} {e_split:.3e}; "
              f"the two GPU paths directly {rel_err(one, one32):.3e}")
        assert not torch.equal(one, one32) and e_split <= 2.0 * e_f32, (e_split, e_f32)
    dh = calc_diffusion_hyperparams(**cfg["diffusion"])
    a = sampling(net, (B, 1, L), dh, condition=mel.to(gpu), seed=5, use_graph=True)
    b = sampling(net, (B, 1, L), dh, condition=mel.to(gpu), seed=5, use_graph=True)
    assert torch.equal(a, b) and torch.isfinite(a).all() and torch.equal(a[2], a[2]) and not torch.equal(a[2], a[3])


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_sampler_at_full_size_is_deterministic(gpu, precision):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg = bench.CONFIGS["wnet_h256_d36_T200"]
    net = cases.build_ours(dict(cfg["model"]), 93).to(gpu)
    net.set_option("precision", precision)
    dh = calc_diffusion_hyperparams(3, 1e-4, 0.05)
    a = sampling(net, (16, 1, 16000), dh, seed=5, use_graph=True)
    b = sampling(net, (16, 1, 16000), dh, seed=5, use_graph=False)
    c = sampling(net, (16, 1, 16000), dh, seed=6, use_graph=True)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()
    assert abs(float(a.std()) - 1.0) < 0.5


def test_config5_training_step_at_full_size(gpu):
    """BASELINE config 5's per-GPU workload: SaShiMi unet_d128_n6, 32 clips of 16000 samples, one training step
    (`train.py:198-222`) through the engine (65 GB of saved activations).  Size-independent properties: the step is
    deterministic (bitwise), every gradient is finite and non-trivial, and -- the loss being a mean over clips -- the
    gradient of the 32-clip batch is the mean of the gradients of its two 16-clip halves (the same x_t, t, z), which is
    also what the 8-rank data-parallel exchange computes."""
    import torch.nn as nn
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import q_sample
    cfg = bench.CONFIGS["unet_d128_n6_T200"]
    B, L = 32, cfg["L"]
    net = cases.build_ours(dict(cfg["model"]), 97).to(gpu).train()
    dh = calc_diffusion_hyperparams(**cfg["diffusion"])
    g = torch.Generator().manual_seed(98)
    audio = (torch.rand(B, 1, L, generator=g) * 2 - 1) * 0.3
    steps = torch.randint(dh["T"], size=(B, 1, 1), generator=g)
    z = torch.normal(0, 1, size=audio.shape, generator=g)
    x_t = q_sample(audio, steps, dh["Alpha_bar"], z).to(gpu)
    steps, z = steps.to(gpu), z.to(gpu)

    def grads(sl):
        net.zero_grad(set_to_none=True)
        loss = nn.MSELoss()(net((x_t[sl], steps[sl].view(-1, 1))), z[sl])
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()}

    l_full, g_full = grads(slice(0, B))
    l_again, g_again = grads(slice(0, B))
    assert l_full == l_again and all(torch.equal(g_full[k], g_again[k]) for k in g_full)
    assert all(torch.isfinite(v).all() for v in g_full.values())
    assert sum(float(v.abs().max()) > 0 for v in g_full.values()) > 0.95 * len(g_full)
    l_a, g_a = grads(slice(0, B // 2))
    l_b, g_b = grads(slice(B // 2, B))
    assert abs(l_full - 0.5 * (l_a + l_b)) < 1e-5 * abs(l_full)
    gmax = max(float(v.abs().max()) for v in g_full.values())
    worst, worst_c = (0.0, None), (0.0, None)
    for k, v in g_full.items():
        half = 0.5 * (g_a[k] + g_b[k])
        scale = max(float(v.abs().max()), 1e-5 * gmax)
        err = float((v - half).abs().max()) / scale
        # the scalar TransposedLN parameters, the 1-input weight_v and the S4 kernel's time scale are sums over all
        # positions / frequencies that cancel to a small remainder: their fp32 noise is measured against float64 in the
        # gradient tests (tests/gradcheck.py; `log_dt` of a 16000-sample stage: the oracle's own fp32 autograd is 1.2e-3
        # off its float64 one for ONE clip); here they only have to agree to a few per cent between two summation orders
        if k.endswith((".m", ".s", "kernel.log_dt")) or k == "init_conv.0.conv.weight_v":
            worst_c = max(worst_c, (err, k))
        else:
            worst = max(worst, (err, k))
    # fp32 sums over 512 000 vs 2 x 256 000 positions in different orders: 1e-3 of each tensor's largest gradient
    assert worst[0] < 1e-3, worst
    assert worst_c[0] < 5e-2, worst_c
    print(f"config 5 at full size: loss {l_full:.5f}, batch-additivity of the gradients to {worst[0]:.2e} ({worst[1]}); "
          f"cancelling scalar sums to {worst_c[0]:.2e} ({worst_c[1]})")
