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
