"""precision="f16x3": the WaveNet layer kernel of `csrc/wavenet_bx6.hip` instantiated with the 2-term fp16 split
(`csrc/bf16_split.h`: SplitF16x2): operands multiplied by a power of two (activations 2^4, gate 2^12, every weight matrix
by the power of two that brings its largest element into (1, 2]), split as h = fp16(x), l = fp16(x - h), three fp16 MFMA
products (h h, h l, l h) accumulated in fp32, the scales undone exactly on the accumulators.  22 bits per operand: NOT
fp32-faithful element by element, so the acceptance is the measured one of tests/test_bf16x6_gpu.py and nothing weaker:

  * GEMM level (`dws_gemm_f16x3`): |C - C64| <= 2^-21 sum |a||b| (the a-priori bound of the representation: 2^-23 per
    operand + the dropped l l product 2^-22), measured next to the fp32 sum in the same k-block order;
  * network level (`models/wavenet.py:82-121,149-165,202-210`): error against the FLOAT64 oracle <= 2 x the exact-f32
    MFMA path's on wn_c128, wn_h128_d30 and wn_h256_d36, max-rel and rms, eps and pre_final;
  * what the split does not cover is stated and tested: scaled operands beyond fp16's range overflow (activations beyond
    2^11 = 2048; the reference's activations are O(1): x' = (x + res) sqrt(.5) of unit-variance audio).
"""
import numpy as np
import pytest
import torch

from oracle import wavenet as own
from tests import cases
from tests.conftest import REL_TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _pow2_scale(t, target=2.0):
    """the packers' rule (csrc/wavenet_bx6.hip: weight_scale_kernel): max|t| * scale in (target/2, target]"""
    m = float(t.abs().max())
    return float(2.0 ** (np.log2(target) - np.ceil(np.log2(m)))) if m > 0 else 1.0


def _gemm_f16x3(A, B, sa, sb):
    from diffwave_sashimi_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.check(lib.dws_gemm_f16x3(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, sa, sb, _lib.current_stream()))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K,kind", [(32, 32, 16, "normal"), (64, 96, 256, "normal"), (128, 64, 1024, "normal"),
                                        (64, 64, 768, "weights"), (64, 64, 768, "small_activations"), (32, 32, 4096, "normal")])
def test_f16x3_gemm_against_float64(gpu, M, N, K, kind):
    """`weights`: A ~ N(0, 0.03) (weight-norm folded conv weights), B ~ N(0, 1) with a few 8-sigma outliers;
    `small_activations`: B ~ 1e-3 N(0, 1): the low fp16 terms are subnormal there and must not be flushed."""
    g = torch.Generator().manual_seed(2000 + M + N + K + len(kind))
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    if kind == "weights":
        A = A * 0.03
        B[torch.randint(0, K, (8,), generator=g), torch.randint(0, N, (8,), generator=g)] *= 8.0
    if kind == "small_activations":
        B = B * 1e-3
    sa, sb = _pow2_scale(A), 64.0
    C = _gemm_f16x3(A.to(gpu).contiguous(), B.to(gpu).contiguous(), sa, sb).cpu().double()
    C64 = A.double() @ B.double()
    scale = A.double().abs() @ B.double().abs()
    err = float(((C - C64).abs() / scale).max())
    seq = torch.zeros(M, N)
    for kb in range(0, K, 16):
        seq = seq + A[:, kb:kb + 16] @ B[kb:kb + 16]
    errseq = float(((seq.double() - C64).abs() / scale).max())
    print(f"M={M} N={N} K={K} {kind}: f16x3 {err:.3e} (2^{np.log2(max(err, 1e-300)):.1f}), fp32 in the same k-block order "
          f"{errseq:.3e}; scales {sa:g} {sb:g}")
    if kind == "small_activations":
        # |b| ~ 2^-10 * 2^6: the low terms sit at 2^-15..2^-16 = fp16 subnormals with 2^-24 absolute resolution, i.e.
        # 2^-20 of such a b: the representation bound is absolute here, 2^-25 / (2^-4 typical |b| scaled) ~ 2^-20
        assert err <= 2.0 ** -19, (err, errseq)
    else:
        assert err <= 2.0 ** -21, (err, errseq)


def test_f16x3_gemm_exact_where_the_terms_are(gpu):
    """Integers up to 2^11 times a power-of-two scale: h carries them exactly, l = 0, every product and sum is exact in
    fp32 -> the result is exact (a misplaced fragment element or a wrong scale shows up as an integer error); and 22-bit
    operands against the identity come back exactly (both terms of A used, the l h product placed right)."""
    g = torch.Generator().manual_seed(9)
    A = torch.randint(-64, 65, (64, 128), generator=g).float()
    B = torch.randint(-64, 65, (128, 96), generator=g).float()
    C = _gemm_f16x3(A.to(gpu), B.to(gpu), 2.0, 64.0).cpu()
    assert torch.equal(C, A @ B)
    A = torch.randint(-(1 << 21), 1 << 21, (32, 16), generator=g).float() * 2.0 ** -20    # 22 significant bits, |a| < 2
    B = torch.zeros(16, 32)
    B[torch.arange(16), torch.arange(16)] = 1.0
    C = _gemm_f16x3(A.to(gpu), B.to(gpu), 1.0, 64.0).cpu()
    assert torch.equal(C[:, :16], A) and torch.count_nonzero(C[:, 16:]) == 0


def test_f16x3_gemm_rejects_what_it_does_not_cover(gpu):
    z = lambda *s: torch.zeros(*s, device=gpu)
    with pytest.raises(NotImplementedError):
        _gemm_f16x3(z(33, 16), z(16, 32), 1.0, 1.0)
    with pytest.raises(RuntimeError):
        _gemm_f16x3(z(32, 16), z(16, 32), 3.0, 1.0)          # not a power of two: the scaling would round


def _f64_oracle(net, cfg, audio, steps):
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in net.state_dict().items()}
    with torch.no_grad():
        return own.wavenet_forward(sd64, cfg, audio.double(), steps, return_pre_final=True)


@pytest.mark.parametrize("name", ["wn_c128", "wn_h128_d30", "wn_h256_d36"])
def test_f16x3_error_against_float64_is_that_of_the_f32_path(gpu, name):
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    ref, ref_pre = cases.cached(("wavenet_f64", name), lambda: _f64_oracle(net, cfg, audio, steps))   # same seeds in both split test files
    out = {}
    with torch.no_grad():
        for prec in ("f32", "f16x3", "bf16x6"):
            net.set_option("precision", prec)
            eps = net((audio.to(gpu), steps.to(gpu)))
            pre = net.read_tap("pre_final", (B, cfg["skip_channels"], L))
            out[prec] = (eps.cpu(), pre.cpu())
        net.set_option("precision", "f32")
        again = net((audio.to(gpu), steps.to(gpu))).cpu()
    assert torch.equal(again, out["f32"][0])
    assert not torch.equal(out["f16x3"][0], out["f32"][0]) and not torch.equal(out["f16x3"][0], out["bf16x6"][0])
    e = {p: (rel_err(out[p][0], ref), rel_err(out[p][1], ref_pre)) for p in out}
    rms = {p: float(((out[p][1].double() - ref_pre) ** 2).mean().sqrt() / (ref_pre ** 2).mean().sqrt()) for p in out}
    print(f"{name}: max-rel error vs float64 (eps, pre_final) f32-MFMA {e['f32'][0]:.3e} {e['f32'][1]:.3e} | "
          f"f16x3 {e['f16x3'][0]:.3e} {e['f16x3'][1]:.3e} | bf16x6 {e['bf16x6'][0]:.3e} {e['bf16x6'][1]:.3e}; "
          f"rms-rel pre_final f32 {rms['f32']:.3e} f16x3 {rms['f16x3']:.3e} bf16x6 {rms['bf16x6']:.3e}")
    for k in (0, 1):
        assert e["f16x3"][k] <= 2.0 * e["f32"][k], (name, k, e)
    assert rms["f16x3"] <= 2.0 * rms["f32"], (name, rms)
    g = load_golden("wavenet")
    assert rel_err(out["f16x3"][0], g[f"{name}/eps"]) < REL_TOL / 100


@pytest.mark.parametrize("name", ["wn_c64", "wn_c128", "wn_h256_d36"])
def test_f16x3_agrees_with_the_f32_winograd_path_at_every_staging_variant(gpu, name):
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed + 9).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for L2, B2 in ((1, 1), (63, 2), (600, 2), (1001, 1), (4096, 1), (4100, 2)):
        audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
        with torch.no_grad():
            net.set_option("precision", "f32")
            w = net((audio.to(gpu), steps.to(gpu)))
            net.set_option("precision", "f16x3")
            s = net((audio.to(gpu), steps.to(gpu)))
            s2 = net((audio.to(gpu), steps.to(gpu)))
        assert torch.equal(s, s2)
        assert rel_err(s, w) < 1e-5, (name, L2, B2, rel_err(s, w))
        if L2 <= 1001 and name != "wn_h256_d36":
            with torch.no_grad():
                ref = own.wavenet_forward(sd, cfg, audio, steps)
            assert rel_err(s, ref) < REL_TOL / 100, (name, L2, B2)


def test_f16x3_conditional_matches_reference(gpu):
    name = "wn_cond_c64"
    cfg, B, L, Tmel, wseed, iseed, store = cases.WAVENET_COND_CASES[name]
    g = load_golden("wavenet_cond")
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.set_option("precision", "f16x3")
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed).to(gpu)
            eps = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel)
            err = rel_err(eps, g[f"{name}/eps_bm{Bm}"])
            assert err < REL_TOL / 100, f"{name} Bm={Bm}: {err:.3e}"
        eps = net((audio.to(gpu), steps.to(gpu)))
        assert rel_err(eps, g[f"{name}/eps_nomel"]) < REL_TOL / 100


def test_f16x3_follows_a_weight_update(gpu):
    """The per-matrix scales are recomputed at every commit: weights scaled by 2^5 after the first forward (another
    power of two per matrix) give the forward of the f32 path on the same weights."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.set_option("precision", "f16x3")
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        net((audio.to(gpu), steps.to(gpu)))
        for k, v in net.named_parameters():
            if k.endswith("res_conv.weight_g") or k.endswith("dilated_conv_layer.conv.weight_g"):
                v.mul_(0.03125 if "res_conv" in k else 3.0)
            if k.endswith("residual_blocks.1.dilated_conv_layer.conv.bias") or k.endswith("residual_blocks.2.skip_conv.bias"):
                v.fill_(40.0)        # a bias a thousand times the weights: enters the matrix scale, must not overflow fp16
        a = net((audio.to(gpu), steps.to(gpu)))
        net.set_option("precision", "f32")
        b = net((audio.to(gpu), steps.to(gpu)))
    assert rel_err(a, b) < 1e-5, rel_err(a, b)


def test_f16x3_range_is_what_the_header_says(gpu):
    """Activations of a few hundred (64 x unit-variance audio) are inside the fp16 range of the scaled split and match the
    f32 path as before; far beyond 2^11 the scaled high terms overflow to infinity and the output is NOT finite -- a range
    violation cannot pass as a plausible finite result."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        out = {}
        for prec in ("f32", "f16x3"):
            net.set_option("precision", prec)
            out[prec] = net(((64.0 * audio).to(gpu), steps.to(gpu))).cpu()
            if prec == "f32":
                xl = net.read_tap("x", (B, cfg["res_channels"], L)).abs().max().item()
                print(f"max |x| leaving the last layer at 64 x audio: {xl:.1f}")
        assert torch.isfinite(out["f16x3"]).all() and rel_err(out["f16x3"], out["f32"]) < 1e-5
        big = net(((3e5 * audio).to(gpu), steps.to(gpu))).cpu()
        assert not torch.isfinite(big).all()


def test_f16x3_sampler_graph_equals_the_per_step_loop(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.set_option("precision", "f16x3")
    T = 6
    dh = calc_diffusion_hyperparams(T, 1e-4, 0.05)
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(B, 1, L, generator=g)
    noise = torch.randn(T, B, 1, L, generator=g)
    a = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=True).cpu()
    b = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=False).cpu()
    assert torch.equal(a, b)
    net.set_option("precision", "f32")
    c = sampling(net, (B, 1, L), dh, x_T=x_T, noise=noise, use_graph=True).cpu()
    assert rel_err(a, c) < 1e-4


def test_f16x3_rejected_where_not_built_and_for_training(gpu):
    net = cases.build_ours(cases.WAVENET_CASES["wn_tiny"][0], 1).to(gpu)
    with pytest.raises(NotImplementedError):
        net.set_option("precision", "f16x3")
