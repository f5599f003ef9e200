"""GPU parity of the reverse-diffusion sampler (`generate.py:23-55`) with
injected noise, eager and hipGraph replay, plus properties of the Philox path."""
import numpy as np
import pytest
import torch

from oracle import diffusion as odiff
from tests import cases
from tests.conftest import REL_TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["T6", "T50"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_sampling_trajectory_matches_reference(gpu, tag, use_graph):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    g = load_golden("sampler")
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_tiny"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    T, b0, bT = g[f"{tag}/args"]
    dh = calc_diffusion_hyperparams(int(T), float(b0), float(bT), fast=True)
    x0 = sampling(net, (B, 1, L), dh, x_T=torch.from_numpy(g[f"{tag}/x_T"]),
                  noise=torch.from_numpy(g[f"{tag}/noise"]), use_graph=use_graph)
    err = rel_err(x0, g[f"{tag}/x_0"])
    assert err < REL_TOL, f"{tag} graph={use_graph}: {err:.3e}"


def test_schedule_tables_host_side():
    """fp32 tables recomputed on THIS host (bit-exact on the machine that made the
    golden file, see tests/test_oracle_golden.py; a different CPU's vector paths may
    round linspace/sqrt differently by an ulp)."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    g = load_golden("schedule")
    dh = calc_diffusion_hyperparams(200, 1e-4, 0.02, fast=True)
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        assert np.allclose(dh[k].numpy(), g[f"sc09/{k}"], rtol=1e-6, atol=0)


def test_graph_and_eager_agree_bitwise_and_seed_controls_rng(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    dh = calc_diffusion_hyperparams(8, 1e-4, 0.05)
    a = sampling(net, (B, 1, L), dh, seed=7, use_graph=True)
    b = sampling(net, (B, 1, L), dh, seed=7, use_graph=False)
    c = sampling(net, (B, 1, L), dh, seed=8, use_graph=True)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    assert torch.isfinite(a).all()


def test_philox_noise_is_standard_normal(gpu):
    """With a net that outputs 0 (zero-init final conv) and T=1 the sampler returns
    x_T / sqrt(alpha_0): the seeded x_T must be N(0,1)."""
    from diffwave_sashimi_amd.models import construct_model
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg = cases.WAVENET_CASES["wn_tiny"][0]
    torch.manual_seed(0)
    net = construct_model(dict(cfg)).to(gpu).eval()
    dh = calc_diffusion_hyperparams(1, 1e-4, 1e-4)
    x = sampling(net, (4, 1, 65536), dh, seed=123).double().cpu() * float(torch.sqrt(dh["Alpha"][0]))
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01
    assert abs(float((x ** 4).mean()) - 3.0) < 0.1          # kurtosis
    assert abs(float((x[:, :, 1:] * x[:, :, :-1]).mean())) < 0.01  # lag-1 correlation


def test_step_table_size_is_bounded(gpu):
    """The WaveNet step table is T x (layers x channels + correction rows) floats: a T whose table would pass 4 GB is refused
    with DWS_ERR_UNSUPPORTED before anything is allocated (four million steps of a 64-channel network here)."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    T = 4_000_000
    dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    with pytest.raises(NotImplementedError, match="step table"):
        sampling(net, (1, 1, 64), dh, seed=1)
    dh = calc_diffusion_hyperparams(4, 1e-4, 0.05)
    assert torch.isfinite(sampling(net, (1, 1, 64), dh, seed=1)).all()      # the model is usable afterwards


@pytest.mark.parametrize("backbone", ["wavenet", "sashimi"])
def test_step_table_sampler_equals_the_per_step_loop(gpu, backbone):
    """The sampler evaluates the step-only part of the network (embedding, MLP, every layer's fc_t, the layer kernels'
    correction fragments) once for t = 0..T-1 and lets the captured step read row t (device step counter); the plain
    forward evaluates it per clip from the steps it is handed.  Same kernels, one output row per wave either way: the
    sampler's trajectory must equal -- bit for bit -- the loop `generate.py:49-54` written out with module calls."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    if backbone == "wavenet":
        cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_c64"]
    else:
        cfg, B, L, wseed = cases.ss_cfg(d_model=32, n_layers=2, L=1024, diffusion_step_embed_dim_mid=64), 3, 1024, 5
    net = cases.build_ours(cfg, wseed).to(gpu)
    T = 5
    dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    g = torch.Generator().manual_seed(77)
    x_T, noise = torch.randn(B, 1, L, generator=g), torch.randn(T, B, 1, L, generator=g)
    got = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=True)
    al, ab, sg = (dh[k] for k in ("Alpha", "Alpha_bar", "Sigma"))

    def loop():
        # the update in numpy float32 on the host: every product / difference / quotient rounded once, as the engine's
        # `smp_update_kernel` does them (torch's device kernels may turn a scalar divisor into a reciprocal multiply)
        x = x_T.numpy().copy()
        with torch.no_grad():
            for t in range(T - 1, -1, -1):
                eps = net((torch.from_numpy(x).to(gpu), torch.full((B, 1), float(t), device=gpu))).cpu().numpy()
                # fp32 scalars in the reference's order (`generate.py:52`), in numpy: correctly rounded like the C float
                # arithmetic of the engine's table (torch's CPU sqrt was seen 1 ulp off on an AVX-512 host)
                a_t, ab_t = np.float32(al[t]), np.float32(ab[t])
                c1 = (np.float32(1) - a_t) / np.sqrt(np.float32(1) - ab_t)
                c2 = np.sqrt(a_t)
                x = (x - c1 * eps) / c2
                if t > 0:
                    x = x + np.float32(sg[t]) * noise[t].numpy()
        assert x.dtype == np.float32
        return torch.from_numpy(x).to(gpu)

    x = loop()
    assert torch.equal(got, x), float((got - x).abs().max())
    # the table follows the weights: new weights, same T -> rebuilt (a stale table would reproduce the old trajectory)
    with torch.no_grad():
        for p_ in net.parameters():
            if p_.is_floating_point():
                p_.mul_(1.01)
    got2 = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=True)
    assert torch.equal(got2, loop()) and not torch.equal(got2, got)
