"""GPU parity: the HIP WaveNet path (through the C ABI) against the reference's
golden outputs and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import wavenet as own
from tests import cases
from tests.conftest import REL_TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(cases.WAVENET_CASES))
def test_wavenet_forward_matches_reference(gpu, name):
    cfg, B, L, wseed, iseed, store = cases.WAVENET_CASES[name]
    g = load_golden("wavenet")
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    with torch.no_grad():
        eps = net((audio.to(gpu), steps.to(gpu)))
        torch.cuda.synchronize()
        assert eps.shape == (B, 1, L) and eps.dtype == torch.float32
        err = rel_err(eps, g[f"{name}/eps"])
        assert err < REL_TOL, f"{name}: rel err {err:.3e} vs reference"
        # internal activation just before the (tiny-weight) output conv
        pre = net.read_tap("pre_final", (B, cfg["skip_channels"], L))
        dg = cases.summarize(pre.cpu(), stride=64)
        assert rel_err(dg["strided"], g[f"{name}/pre_final/strided"]) < REL_TOL
        assert rel_err(dg["last"], g[f"{name}/pre_final/last"]) < REL_TOL
        # int64 steps (training call convention, `train.py:218,221`) give the same result
        eps_i = net((audio.to(gpu), steps.long().to(gpu)))
        assert torch.equal(eps, eps_i)
        # deterministic
        assert torch.equal(eps, net((audio.to(gpu), steps.to(gpu))))
    print(f"{name}: rel err vs reference {err:.3e}")


@pytest.mark.parametrize("name", ["wn_c64", "wn_c128", "wn_h128_d30", "wn_h256_d36"])
def test_wavenet_bf16x3_precision_mode_matches_reference(gpu, name):
    """Opt-in precision="bf16x3" (3-term bf16 split on the matrix cores, fp32 accumulate) must still
    meet the 1e-3 parity bound against the reference's fp32 forward -- it sits ~100x inside it."""
    cfg, B, L, wseed, iseed, store = cases.WAVENET_CASES[name]
    g = load_golden("wavenet")
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    with torch.no_grad():
        e32 = net((audio.to(gpu), steps.to(gpu)))
        net.set_option("precision", "bf16x3")
        e16 = net((audio.to(gpu), steps.to(gpu)))
        pre = net.read_tap("pre_final", (B, cfg["skip_channels"], L))
        net.set_option("precision", "f32")
        e32b = net((audio.to(gpu), steps.to(gpu)))
    err = rel_err(e16, g[f"{name}/eps"])
    assert err < REL_TOL / 10, f"{name}: bf16x3 rel err {err:.3e}"
    dg = cases.summarize(pre.cpu(), stride=64)
    assert rel_err(dg["strided"], g[f"{name}/pre_final/strided"]) < REL_TOL / 10
    assert not torch.equal(e16, e32)          # it really is a different arithmetic path
    assert torch.equal(e32, e32b)             # and switching back restores the exact-f32 path bit for bit
    print(f"{name}: bf16x3 rel err vs reference {err:.3e}")


@pytest.mark.parametrize("L2,B2", [(1, 1), (63, 2), (129, 2), (1001, 1), (1024, 2)])
def test_wavenet_bf16x3_ragged_lengths_match_oracle(gpu, L2, B2):
    """Lengths that are not multiples of the 128-position tile / of 4 (scalar epilogue path)."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed + 3).to(gpu)
    net.set_option("precision", "bf16x3")
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
    with torch.no_grad():
        ref = own.wavenet_forward(sd, cfg, audio, steps)
        got = net((audio.to(gpu), steps.to(gpu)))
    assert rel_err(got, ref) < REL_TOL / 10, (L2, B2)


def test_bf16x3_rejected_where_not_built(gpu):
    net = cases.build_ours(cases.WAVENET_CASES["wn_tiny"][0], 1).to(gpu)
    with pytest.raises(NotImplementedError):
        net.set_option("precision", "bf16x3")
    with pytest.raises(RuntimeError):
        net.set_option("precision", "fp8")


@pytest.mark.parametrize("name", ["wn_tiny", "wn_c64"])
def test_wavenet_matches_oracle_on_fresh_inputs(gpu, name):
    """Same seeded inputs through the oracle and the HIP path (no golden file involved)."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed + 5).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for L2, B2 in ((1, 1), (63, 3), (64, 1), (65, 2), (1000, 2)):
        audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
        with torch.no_grad():
            ref = own.wavenet_forward(sd, cfg, audio, steps)
            got = net((audio.to(gpu), steps.to(gpu)))
        assert rel_err(got, ref) < REL_TOL, (L2, B2)


def test_load_state_dict_changes_output(gpu):
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    other = cases.build_ours(cfg, wseed + 1)
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        e0 = net((audio.to(gpu), steps.to(gpu)))
        net.load_state_dict(other.state_dict())
        e1 = net((audio.to(gpu), steps.to(gpu)))
        ref = own.wavenet_forward({k: v.detach() for k, v in other.state_dict().items()}, cfg, audio, steps)
    assert not torch.equal(e0, e1)
    assert rel_err(e1, ref) < REL_TOL


def test_zero_init_network_outputs_exact_zero(gpu):
    """`ZeroConv1d` is zero-initialised (`wavenet.py:35-36`): an untrained net returns exactly 0."""
    from diffwave_sashimi_amd.models import construct_model
    cfg = cases.WAVENET_CASES["wn_c64"][0]
    torch.manual_seed(0)
    net = construct_model(dict(cfg)).to(gpu).eval()
    audio, steps = cases.wavenet_inputs(2, 256, 1, 3)
    with torch.no_grad():
        eps = net((audio.to(gpu), steps.to(gpu)))
    assert torch.count_nonzero(eps) == 0


@pytest.mark.parametrize("name", list(cases.WAVENET_COND_CASES))
def test_wavenet_conditional_matches_reference(gpu, name):
    cfg, B, L, Tmel, wseed, iseed, store = cases.WAVENET_COND_CASES[name]
    g = load_golden("wavenet_cond")
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed).to(gpu)
            eps = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel)
            err = rel_err(eps, g[f"{name}/eps_bm{Bm}"])
            assert err < REL_TOL, f"{name} Bm={Bm}: {err:.3e}"
        eps = net((audio.to(gpu), steps.to(gpu)))
        assert rel_err(eps, g[f"{name}/eps_nomel"]) < REL_TOL


def test_shape_errors_are_loud(gpu):
    cfg = cases.WAVENET_CASES["wn_tiny"][0]
    net = cases.build_ours(cfg, 1).to(gpu)
    with pytest.raises(RuntimeError):
        net((torch.zeros(2, 1, 64, device=gpu), torch.zeros(3, 1, device=gpu)))
    bad = {k: v for k, v in net.state_dict().items()}
    bad["init_conv.0.conv.bias"] = torch.zeros(17)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)


@pytest.mark.parametrize("name", ["wn_c64", "wn_c128", "wn_h256_d36"])
def test_winograd_and_direct_layer_kernels_agree(gpu, name):
    """The default fused layer computes the dilated conv in Winograd F(2,3) form along the dilation stride
    (`csrc/wavenet_wino.hip`); `conv_algo=direct` is the three-tap form.  Same weights, same inputs: the two must agree
    far inside the 1e-3 bound -- at every staging variant of the Winograd kernel: 16-byte LDS-DMA (L % 4 == 0, d >= 4),
    the contiguous-row form (d <= 16), the dword form (L % 4 != 0), positions past L in the last pair block, d > L."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed + 9).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for L2, B2 in ((1, 1), (63, 2), (600, 2), (1001, 1), (4096, 1), (4100, 2)):
        audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
        with torch.no_grad():
            net.set_option("conv_algo", "winograd")
            w = net((audio.to(gpu), steps.to(gpu)))
            w2 = net((audio.to(gpu), steps.to(gpu)))
            net.set_option("conv_algo", "direct")
            d = net((audio.to(gpu), steps.to(gpu)))
        assert torch.equal(w, w2)                                  # deterministic (the skip atomics are one add per element)
        assert not torch.equal(w, d)                               # two different arithmetic paths
        assert rel_err(w, d) < 2e-5, (name, L2, B2, rel_err(w, d))
        if L2 <= 1001 and name != "wn_h256_d36":
            with torch.no_grad():
                ref = own.wavenet_forward(sd, cfg, audio, steps)
            assert rel_err(w, ref) < REL_TOL / 10, (name, L2, B2)
    with pytest.raises(RuntimeError):
        net.set_option("conv_algo", "fft")
