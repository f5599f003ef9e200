"""GPU parity: the HIP SaShiMi / S4 path (through the C ABI) against the reference's
golden outputs and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sashimi as osa
from tests import cases
from tests.conftest import REL_TOL, load_golden, rel_err
from tests.test_sashimi_oracle import _sd0, _sd1

pytestmark = pytest.mark.gpu


def _run(net, gpu, audio, steps, mel=None):
    with torch.no_grad():
        out = net((audio.to(gpu), steps.to(gpu)), mel_spec=None if mel is None else mel.to(gpu))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", list(cases.SASHIMI_CASES))
def test_sashimi_forward_matches_reference(gpu, name):
    cfg, B, wseed, iseed, store = cases.SASHIMI_CASES[name]
    g = load_golden("sashimi")
    net = cases.build_ours(cfg, wseed).to(gpu)
    if store:  # run on the reference's own post-warm-up parameters (C~, L = l_max)
        net.load_state_dict({k: v.to(gpu) for k, v in _sd1(g, name).items()})
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    eps = _run(net, gpu, audio, steps)
    assert eps.shape == (B, 1, cfg["L"])
    err = rel_err(eps, g[f"{name}/eps"])
    assert err < REL_TOL, f"{name}: rel err {err:.3e} vs reference"
    pre = net.read_tap("pre_final", (B, cfg["d_model"], cfg["L"]))
    dg = cases.summarize(pre.cpu(), stride=64)
    assert rel_err(dg["strided"], g[f"{name}/pre_final/strided"]) < REL_TOL
    assert torch.equal(eps, _run(net, gpu, audio, steps))            # deterministic
    assert torch.equal(eps, _run(net, gpu, audio, steps.long()))     # int64 steps (train.py:218)
    print(f"{name}: rel err vs reference {err:.3e}")


@pytest.mark.parametrize("name", ["ss_tiny", "ss_knobs"])
def test_s4_kernel_generator_matches_reference(gpu, name):
    """The convolution kernel built at weight-load time (Cauchy -> Woodbury -> bilinear factor ->
    irfft, s4.py:704-807) vs the reference's k."""
    g = load_golden("sashimi")
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES[name]
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.load_state_dict({k: v.to(gpu) for k, v in _sd1(g, name).items()})
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    _run(net, gpu, audio, steps)
    for key in g.files:
        if not key.startswith(f"{name}/k/"):
            continue
        prefix = key.split("/")[-1]
        k = torch.from_numpy(g[key])                       # (2, H, L)
        got = net.read_tap("k:" + prefix, tuple(k.shape)).cpu() / k.shape[-1]   # engine keeps L * k
        assert rel_err(got, k) < 1e-4, prefix


@pytest.mark.parametrize("L", [64, 250, 1000, 1024, 4000])
def test_fused_fft_convolution_matches_direct_convolution(gpu, L):
    """The single-kernel LDS FFT convolution (power-of-two size, re-placed anti-causal half) against
    the definition y[i] = sum_j k0[j] u[i-j] + sum_{m>=1} k1[m-1] u[i+m] evaluated in float64 through
    a 1-layer model; also exercised against the rocFFT (n = 2L) path of the same engine."""
    import os
    cfg = cases.ss_cfg(d_model=8, n_layers=1, L=L, pool=[], diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 77).to(gpu)
    audio, steps = cases.wavenet_inputs(3, L, 1, 78)
    got = _run(net, gpu, audio, steps)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = osa.sashimi_forward(sd, cfg, audio, steps)
    assert rel_err(got, ref) < REL_TOL
    os.environ["DWS_SASHIMI_ROCFFT"] = "1"
    try:
        net2 = cases.build_ours(cfg, 77).to(gpu)
        net2.load_state_dict(net.state_dict())
        got2 = _run(net2, gpu, audio, steps)
    finally:
        del os.environ["DWS_SASHIMI_ROCFFT"]
    assert rel_err(got, got2) < 1e-4


@pytest.mark.parametrize("name", list(cases.SASHIMI_COND_CASES))
def test_sashimi_conditional_matches_reference(gpu, name):
    cfg, B, Tmel, wseed, iseed, store = cases.SASHIMI_COND_CASES[name]
    g = load_golden("sashimi_cond")
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    for Bm in (1, B):
        mel = cases.mel_inputs(Bm, Tmel, iseed)
        err = rel_err(_run(net, gpu, audio, steps, mel), g[f"{name}/eps_bm{Bm}"])
        assert err < REL_TOL, f"{name} Bm={Bm}: {err:.3e}"
    assert rel_err(_run(net, gpu, audio, steps), g[f"{name}/eps_nomel"]) < REL_TOL


def test_config4_geometry_matches_reference(gpu):
    """BASELINE config 4's own geometry against the reference (tests/golden/sashimi_c4.npz): unet_d32_n6, L = 16000,
    mel [1 | B, 80, 63] whose 16128 upsampled frames are truncated to 16000 / 4000 / 1000 at the three stages."""
    cfg, B, Tmel, wseed, iseed = cases.SASHIMI_C4
    g = load_golden("sashimi_c4")
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    for Bm in (1, B):
        mel = cases.mel_inputs(Bm, Tmel, iseed)
        eps = _run(net, gpu, audio, steps, mel)
        err = rel_err(eps, g[f"eps_bm{Bm}"])
        assert err < REL_TOL, f"Bm={Bm}: {err:.3e}"
        pre = net.read_tap("pre_final", (B, cfg["d_model"], cfg["L"]))
        assert rel_err(cases.summarize(pre.cpu(), stride=64)["strided"], g[f"pre_final_bm{Bm}/strided"]) < REL_TOL
        print(f"config-4 geometry, mel batch {Bm}: rel err vs reference {err:.3e}")


def test_sashimi_matches_oracle_fresh_weights_and_first_forward_mutation(gpu):
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES["ss_knobs"]
    net = cases.build_ours(cfg, wseed + 9).to(gpu)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    audio, steps = cases.wavenet_inputs(4, cfg["L"], 1, iseed + 1)
    with torch.no_grad():
        ref = osa.sashimi_forward(sd0, cfg, audio, steps)
    got = _run(net, gpu, audio, steps)
    assert rel_err(got, ref) < REL_TOL
    sd1 = net.state_dict()
    k = "c_layers.0.layer.kernel.kernel"
    assert int(sd0[k + ".L"]) == 0 and int(sd1[k + ".L"]) == cfg["L"] // 2   # pool [2]
    assert not torch.equal(sd0[k + ".C"], sd1[k + ".C"].cpu())


def test_sashimi_sampler_matches_oracle(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    from oracle import diffusion as odiff
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES["ss_tiny"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    T = 5
    dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    g = torch.Generator().manual_seed(3)
    x_T = torch.randn(B, 1, cfg["L"], generator=g)
    noise = torch.randn(T, B, 1, cfg["L"], generator=g)
    x_graph = sampling(net, (B, 1, cfg["L"]), dh, x_T=x_T, noise=noise, use_graph=True)
    x_eager = sampling(net, (B, 1, cfg["L"]), dh, x_T=x_T, noise=noise, use_graph=False)
    assert torch.equal(x_graph, x_eager)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = odiff.sampling(osa.SashimiOracle(sd, cfg), (B, 1, cfg["L"]), odiff.calc_diffusion_hyperparams(T, 1e-4, 0.05),
                         x_T=x_T, noise=noise)
    assert rel_err(x_graph, ref) < REL_TOL


def test_sashimi_errors_are_loud(gpu):
    cfg = cases.SASHIMI_CASES["ss_tiny"][0]
    net = cases.build_ours(cfg, 1).to(gpu)
    with pytest.raises(RuntimeError):          # 1000 is not divisible by the pooling span 16
        net((torch.zeros(1, 1, 1000, device=gpu), torch.zeros(1, 1, device=gpu)))
    from diffwave_sashimi_amd.models import construct_model
    with pytest.raises(ValueError):
        construct_model(dict(cfg, diffusion_step_embed_dim_out=64))
    with pytest.raises(RuntimeError):
        construct_model(dict(cfg, L=1000, pool=[3, 7])).to(gpu)((torch.zeros(1, 1, 1000, device=gpu),
                                                                 torch.zeros(1, 1, device=gpu)))


def test_variable_length_calls_match_reference_sequence(gpu):
    """One module, four calls in a row at L, L/2, 2L, L (`s4.py:1387`): truncated kernels for the short input, l_max
    taps for the long one (fused LDS FFT at M = 1024 covers all three lengths here), parameters untouched."""
    from tests.test_sashimi_oracle import VARLEN_CFG
    g = load_golden("sashimi_varlen")
    net = cases.build_ours(VARLEN_CFG, 1).to(gpu)
    net.load_state_dict({k[len("sd0/"):]: torch.from_numpy(v).to(gpu) for k, v in g.items() if k.startswith("sd0/")})
    for i, L_in in enumerate([256, 128, 512, 256]):
        out = _run(net, gpu, torch.from_numpy(g[f"call{i}/audio"]), torch.from_numpy(g[f"call{i}/steps"]))
        assert out.shape == (2, 1, L_in)
        assert rel_err(out.cpu(), torch.from_numpy(g[f"call{i}/eps"])) < REL_TOL, (i, L_in)
        Ls = [int(v) for k, v in sorted(net.state_dict().items()) if k.endswith("kernel.kernel.L")]
        assert Ls == list(g[f"call{i}/L"])


@pytest.mark.parametrize("L_in", [4000, 20000, 40000, 212992])
def test_long_and_short_utterances_against_oracle(gpu, L_in):
    """Vocoder-style lengths around l_max = 16000 on the config-4 channel counts: 4000 (kernel truncated); 20000, 40000
    and 212992 = 832 mel frames, an LJSpeech utterance (`generate.py:156`): l_max taps, stages beyond the largest LDS
    transform (16384) run the segmented fused convolution (`fftconv_seg_kernel`); vs the CPU oracle."""
    cfg = cases.ss_cfg(d_model=32, n_layers=1, L=16000)
    net = cases.build_ours(cfg, 77).to(gpu)
    gen = torch.Generator().manual_seed(78)
    audio, steps = torch.randn(1, 1, L_in, generator=gen), torch.tensor([[37.0]])
    out = _run(net, gpu, audio, steps)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = osa.sashimi_forward(sd, cfg, audio, steps)
    assert rel_err(out.cpu(), ref) < REL_TOL


def test_segmented_long_rows_agree_with_the_rocfft_path(gpu):
    """Same engine, same weights, L = 40000 (top stage 40000 > 16384): the segmented fused path against the rocFFT
    n = 2L path (DWS_SASHIMI_ROCFFT=1), B = 3, two blocks per level."""
    import os
    cfg = cases.ss_cfg(d_model=32, n_layers=2, L=16000)
    net = cases.build_ours(cfg, 79).to(gpu)
    gen = torch.Generator().manual_seed(80)
    audio, steps = torch.randn(3, 1, 40000, generator=gen), torch.tensor([[3.0], [120.0], [199.0]])
    got = _run(net, gpu, audio, steps)
    os.environ["DWS_SASHIMI_ROCFFT"] = "1"
    try:
        net2 = cases.build_ours(cfg, 79).to(gpu)
        net2.load_state_dict(net.state_dict())
        ref = _run(net2, gpu, audio, steps)
    finally:
        del os.environ["DWS_SASHIMI_ROCFFT"]
    assert rel_err(got, ref) < 1e-4
    assert not torch.equal(got, ref)            # two different code paths really ran


@pytest.mark.parametrize("name", ["ss_d64_short", "ss_d128_short"])
def test_fused_next_block_layernorm_agrees_with_the_separate_pass(gpu, name):
    """Every LayerNorm of a forward comes out of its producer's epilogue: a block's tail kernel writes the next block's S4
    input (LN1 + step embedding, `sashimi.py:148-152`) when both sit on one stage, the init conv and the pooling GEMMs
    write the first block's of their stage, the last tail the network's final norm; DWS_SASHIMI_NO_LN_FUSION=1 runs the
    separate LayerNorm passes instead.  Same weights, both ways."""
    import os
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES[name]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    fused = _run(net, gpu, audio, steps)
    os.environ["DWS_SASHIMI_NO_LN_FUSION"] = "1"
    try:
        plain = _run(net, gpu, audio, steps)
    finally:
        del os.environ["DWS_SASHIMI_NO_LN_FUSION"]
    assert rel_err(fused, plain) < 1e-5
    assert torch.equal(fused, _run(net, gpu, audio, steps))


@pytest.mark.parametrize("name", ["ss_d64_short", "ss_d128_short", "ss_cond_d32"])
@pytest.mark.parametrize("switch", ["DWS_TAIL_CFG", "DWS_TAIL_NO_VEC", "DWS_TAIL_NO_CHAIN"])
def test_tail_kernel_variants_agree(gpu, name, switch):
    """The fused tail kernel (`sashimi.py:177-184`) in its other tile shapes (DWS_TAIL_CFG=1: 128-position tiles), with
    per-lane dword instead of 16-byte global traffic (DWS_TAIL_NO_VEC=1, the path L % 4 != 0 takes), and the LDS-tile
    kernel where the default is the register-chained one (DWS_TAIL_NO_CHAIN=1: H = 32 / 64 stages) -- same weights, same
    inputs (the conditional case with its mel term); only the order of the LayerNorm partial sums differs between shapes."""
    import os
    mel = None
    if name in cases.SASHIMI_COND_CASES:
        cfg, B, Tmel, wseed, iseed, _ = cases.SASHIMI_COND_CASES[name]
        mel = cases.mel_inputs(B, Tmel, iseed)
    else:
        cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES[name]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    base = _run(net, gpu, audio, steps, mel)
    os.environ[switch] = "1"
    try:
        other = _run(net, gpu, audio, steps, mel)
    finally:
        del os.environ[switch]
    assert rel_err(base, other) < 1e-5
    assert torch.equal(base, _run(net, gpu, audio, steps, mel))


def test_checkpoint_with_longer_kernels_than_configured_runs_long_inputs(gpu):
    """A checkpoint whose S4 kernels were set up for a longer l_max than the model is configured with (its `L` buffers
    say 32768, the config 16384): at a 65536-sample input the top stage is picked for the segmented convolution from the
    CONFIGURED length, but the kernels carry 32768 taps per direction -- more than a segment holds.  The engine must
    fall back to the rocFFT convolution for that stage (it used to abort with DWS_ERR_UNSUPPORTED), and agree with
    the CPU oracle."""
    long_cfg = cases.ss_cfg(d_model=8, n_layers=1, L=32768, diffusion_step_embed_dim_mid=64)
    donor = cases.build_ours(long_cfg, 91)
    donor._setup_C()                                   # C~ and L = 32768 / 8192 / 2048, as a checkpoint stores them
    sd = {k: v.detach().clone() for k, v in donor.state_dict().items()}
    short_cfg = dict(long_cfg, L=16384)
    net = cases.build_ours(short_cfg, 92)
    net.load_state_dict(sd)
    net = net.to(gpu)
    k = next(iter(net._blocks())).layer.kernel.kernel
    assert int(k.L) == 32768
    gen = torch.Generator().manual_seed(93)
    audio, steps = torch.randn(1, 1, 65536, generator=gen), torch.tensor([[11.0]])
    out = _run(net, gpu, audio, steps)
    with torch.no_grad():
        ref = osa.sashimi_forward({k_: v.cpu() for k_, v in net.state_dict().items()}, short_cfg, audio, steps)
    assert rel_err(out.cpu(), ref) < REL_TOL
    # same length, LARGER batch: the stage must stay on rocFFT with buffers and plans for the new batch (prepare() used
    # to re-pick the segmented path from the configured length; the layers then ran rocFFT on buffers sized for B = 1)
    audio3 = torch.cat([audio, torch.randn(2, 1, 65536, generator=gen)])
    steps3 = torch.tensor([[11.0], [3.0], [40.0]])
    out3 = _run(net, gpu, audio3, steps3)
    assert rel_err(out3[:1].cpu(), ref) < REL_TOL
    with torch.no_grad():
        ref3 = osa.sashimi_forward({k_: v.cpu() for k_, v in net.state_dict().items()}, short_cfg, audio3[2:], steps3[2:])
    assert rel_err(out3[2:].cpu(), ref3) < REL_TOL
    # shorter input afterwards: back on the fused paths, still right
    audio2 = torch.randn(1, 1, 16384, generator=gen)
    out2 = _run(net, gpu, audio2, steps)
    with torch.no_grad():
        ref2 = osa.sashimi_forward({k_: v.cpu() for k_, v in net.state_dict().items()}, short_cfg, audio2, steps)
    assert rel_err(out2.cpu(), ref2) < REL_TOL
