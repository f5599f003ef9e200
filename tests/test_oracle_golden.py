"""CPU: pin the oracle (oracle/) against the golden vectors produced from the
reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import diffusion as odiff
from oracle import wavenet as own
from tests import cases
from tests.conftest import load_golden, rel_err


def test_step_embedding_matches_reference():
    g = load_golden("embedding")
    e = odiff.calc_diffusion_step_embedding(torch.from_numpy(g["t_float"]), 128)
    assert torch.equal(e, torch.from_numpy(g["emb_float"]))
    e = odiff.calc_diffusion_step_embedding(torch.from_numpy(g["t_int"]), 128)
    assert torch.equal(e, torch.from_numpy(g["emb_int"]))
    e = odiff.calc_diffusion_step_embedding(torch.from_numpy(g["t_float"]), 64)
    assert torch.equal(e, torch.from_numpy(g["emb_float_64"]))


@pytest.mark.parametrize("tag", ["sc09", "ljspeech", "tiny"])
def test_schedule_tables_bit_exact(tag):
    g = load_golden("schedule")
    T, b0, bT = g[f"{tag}/args"]
    dh = odiff.calc_diffusion_hyperparams(int(T), float(b0), float(bT), fast=True)
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        assert np.array_equal(dh[k].numpy(), g[f"{tag}/{k}"]), k


def test_schedule_fast_beta_hook():
    g = load_golden("schedule")
    dh = odiff.calc_diffusion_hyperparams(200, 1e-4, 0.02, beta=[float(b) for b in g["fast/beta"]], fast=True)
    assert dh["T"] == 6
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        assert np.array_equal(dh[k].numpy(), g[f"fast/{k}"]), k


def test_schedule_known_values():
    # checked values for T=200
    dh = odiff.calc_diffusion_hyperparams(200, 1e-4, 0.02)
    assert abs(float(dh["Alpha_bar"][-1]) - 0.13218278) < 1e-7
    assert abs(float(dh["Sigma"][-1]) - 0.14120138) < 1e-7


@pytest.mark.parametrize("name", list(cases.WAVENET_CASES))
def test_wavenet_oracle_matches_reference(name):
    cfg, B, L, wseed, iseed, store = cases.WAVENET_CASES[name]
    g = load_golden("wavenet")
    ours = cases.build_ours(cfg, wseed)
    sd = {k: v.detach() for k, v in ours.state_dict().items()}
    # the seeded weights are the ones the golden run used
    dig = np.array([sum(float(v.double().sum()) for v in sd.values()),
                    sum(float((v.double() ** 2).sum()) for v in sd.values())])
    assert np.allclose(dig, g[f"{name}/sd_digest"], rtol=1e-12)
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    if store:
        assert np.array_equal(audio.numpy(), g[f"{name}/audio"])
        for k, v in sd.items():
            assert np.array_equal(v.numpy(), g[f"{name}/sd/{k}"]), k
    with torch.no_grad():
        eps, pre = own.wavenet_forward(sd, cfg, audio, steps, return_pre_final=True)
    assert rel_err(eps, g[f"{name}/eps"]) < 1e-5
    dg = cases.summarize(pre, stride=64)
    assert rel_err(dg["strided"], g[f"{name}/pre_final/strided"]) < 1e-5
    assert rel_err(dg["first"], g[f"{name}/pre_final/first"]) < 1e-5
    ref_sumsq = float(np.asarray(g[f"{name}/pre_final/sumsq"]).reshape(-1)[0])
    assert abs(float(np.asarray(dg["sumsq"]).reshape(-1)[0]) - ref_sumsq) <= 1e-5 * ref_sumsq


@pytest.mark.parametrize("name", list(cases.WAVENET_COND_CASES))
def test_wavenet_cond_oracle_matches_reference(name):
    cfg, B, L, Tmel, wseed, iseed, store = cases.WAVENET_COND_CASES[name]
    g = load_golden("wavenet_cond")
    ours = cases.build_ours(cfg, wseed)
    sd = {k: v.detach() for k, v in ours.state_dict().items()}
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    with torch.no_grad():
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed)
            eps = own.wavenet_forward(sd, cfg, audio, steps, mel_spec=mel)
            assert rel_err(eps, g[f"{name}/eps_bm{Bm}"]) < 1e-5
        eps = own.wavenet_forward(sd, cfg, audio, steps)
        assert rel_err(eps, g[f"{name}/eps_nomel"]) < 1e-5


@pytest.mark.parametrize("tag", ["T6", "T50"])
def test_sampler_oracle_matches_reference(tag):
    g = load_golden("sampler")
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_tiny"]
    ours = cases.build_ours(cfg, wseed)
    net = own.WaveNetOracle(ours.state_dict(), cfg)
    T, b0, bT = g[f"{tag}/args"]
    dh = odiff.calc_diffusion_hyperparams(int(T), float(b0), float(bT))
    x0 = odiff.sampling(net, (B, 1, L), dh, x_T=torch.from_numpy(g[f"{tag}/x_T"]),
                        noise=torch.from_numpy(g[f"{tag}/noise"]))
    assert rel_err(x0, g[f"{tag}/x_0"]) < 1e-5
    # seeded (non-injected) mode reproduces the reference's RNG consumption order
    torch.manual_seed(1234)
    x0b = odiff.sampling(net, (B, 1, L), dh)
    assert rel_err(x0b, g[f"{tag}/x_0"]) < 1e-5
