"""Training-loss gradients pinned against the REFERENCE itself (`tests/golden/grads.npz`: backward through the imported
reference modules, `make_golden.py::g_grads`): the oracle's autograd on CPU, and the HIP engine's hand-written
backward on the GPU, on the same weights, audio, (mel), diffusion steps and noise."""
import pytest
import torch
import torch.nn as nn

from oracle import sashimi as osa
from oracle import wavenet as own
from tests import cases
from tests.conftest import load_golden
from tests.golden.make_golden_cases import GRAD_CASES


def _case(g, name):
    cfg, B, L, Tmel = GRAD_CASES[name]
    sd0 = {k: v.detach().clone() for k, v in cases.build_ours(cfg, 311).state_dict().items()}   # as the generator did
    digest = float(torch.stack([v.double().sum() for v in sd0.values() if v.is_floating_point()]).sum())
    assert abs(digest - float(g[f"{name}/sd0_digest"][0])) < 1e-6 * max(1.0, abs(digest)), "seeded weights differ"
    grads = {k[len(name) + 6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"{name}/grad/")}
    audio = torch.from_numpy(g[f"{name}/audio"])
    mel = torch.from_numpy(g[f"{name}/mel"]) if f"{name}/mel" in g else None
    return cfg, B, L, sd0, grads, audio, mel, float(g[f"{name}/loss"][0])


def _loss(net, audio, mel):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    torch.manual_seed(314)                      # the reference run drew t and z from the global RNG after this seed
    return training_loss(net, nn.MSELoss(), audio, dh, mel_spec=mel)


def _compare(got, ref, tol):
    gmax = max(float(v.abs().max()) for v in ref.values())
    bad = []
    for k, r in ref.items():
        scale = max(float(r.abs().max()), 1e-5 * gmax)
        err = float((got[k] - r).abs().max()) / scale
        if err >= tol:
            bad.append(f"{k}: {err:.2e}")
    assert not bad, bad[:20]


@pytest.mark.parametrize("name", list(GRAD_CASES))
def test_oracle_autograd_matches_reference_gradients(name):
    g = load_golden("grads")
    cfg, B, L, sd0, ref, audio, mel, ref_loss = _case(g, name)
    from diffwave_sashimi_amd.models import construct_model
    m = construct_model(dict(cfg))
    m.load_state_dict(sd0)
    if cfg["_name_"] == "sashimi":
        m._setup_C()                            # what the reference's first forward does before its graph is built
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    fwd = own.wavenet_forward if cfg["_name_"] == "wavenet" else osa.sashimi_forward
    loss = _loss(lambda inp, mel_spec=None: fwd(sd, cfg, inp[0], inp[1], mel_spec=mel_spec), audio, mel)
    loss.backward()
    assert abs(float(loss) - ref_loss) < 2e-6 * max(1.0, abs(ref_loss))
    got = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in ref}
    # SaShiMi: the scalar LayerNorm parameters and the 1-input weight_v are heavily cancelling sums (their fp32 noise
    # between two equivalent op orders reaches 5e-3 of the tensor's largest gradient); everything else is < 1e-3
    _compare(got, ref, 2e-3 if cfg["_name_"] == "wavenet" else 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ss", "ss_cond"])
def test_engine_backward_matches_reference_gradients(gpu, name):
    """The tiny SaShiMi fixtures run on the engine's generic forward kernels but need H % 32 == 0 for the MFMA adjoints:
    they are expected to raise; the channel counts the engine trains at are checked against the oracle (which the test
    above pins to these reference gradients) in test_*_training_gpu.py."""
    g = load_golden("grads")
    cfg, B, L, sd0, ref, audio, mel, ref_loss = _case(g, name)
    from diffwave_sashimi_amd.models import construct_model
    net = construct_model(dict(cfg)).to(gpu).train()
    net.load_state_dict({k: v.to(gpu) for k, v in sd0.items()})
    with pytest.raises(NotImplementedError):
        _loss(net, audio.to(gpu), None if mel is None else mel.to(gpu)).backward()


@pytest.mark.gpu
def test_engine_wavenet_backward_matches_reference_gradients(gpu):
    """WaveNet at C = 16 trains on the engine's generic adjoints: compared with the reference's own gradients."""
    g = load_golden("grads")
    cfg, B, L, sd0, ref, audio, mel, ref_loss = _case(g, "wn")
    from diffwave_sashimi_amd.models import construct_model
    net = construct_model(dict(cfg)).to(gpu).train()
    net.load_state_dict({k: v.to(gpu) for k, v in sd0.items()})
    loss = _loss(net, audio.to(gpu), None)
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    _compare({k: p.grad.detach().cpu() for k, p in net.named_parameters()}, ref, 2e-3)
