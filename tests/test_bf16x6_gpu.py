"""precision="bf16x6": the WaveNet layer on the bf16 matrix cores at fp32-equivalent accuracy (3-term bf16 split, six
partial products, fp32 accumulate; `csrc/bf16_split.h`, `csrc/wavenet_bx6.hip`).

Acceptance is numerical and measured against FLOAT64, not against the fp32 reference (whose own rounding is as large as
what is being measured):
  * GEMM level: the split arithmetic alone (`dws_gemm_bf16x6`, the per-k-block sequence of the layer kernels) within
    2^-22 of the float64 product relative to sum |a||b| -- next to torch's fp32 matmul on the same operands;
  * network level (`models/wavenet.py:82-121,149-165,202-210`): error of the split path <= 2 x error of the exact-f32
    MFMA path, both against the float64 oracle, on wn_c128, wn_h128_d30 (BASELINE config 1 at L = 16000) and wn_h256_d36.
"""
import numpy as np
import pytest
import torch

from oracle import wavenet as own
from tests import cases
from tests.conftest import REL_TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _gemm_bx6(A, B):
    from diffwave_sashimi_amd import _lib
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    _lib.check(lib.dws_gemm_bf16x6(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, _lib.current_stream()))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K,spread", [(32, 32, 16, 0), (64, 96, 256, 0), (128, 64, 1024, 0), (64, 64, 768, 12),
                                          (32, 32, 4096, 6)])
def test_split_gemm_is_fp32_class_against_float64(gpu, M, N, K, spread):
    """|C - C64| <= 2^-22 * sum_k |a||b| elementwise.  `spread` scales every operand element by 2^U(-spread, spread):
    terms of very different magnitude in one dot product (what the dropped x1 w2 + x2 w1 + x2 w2 products would hurt);
    there the fp32 accumulation itself leaves 2^-22 (a few terms carry the sum, K roundings of the running sum), so the
    bound is the error of plain fp32 arithmetic in the kernel's own accumulation order on the same operands."""
    g = torch.Generator().manual_seed(1000 + M + N + K + spread)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    if spread:
        A = A * torch.exp2(torch.randint(-spread, spread + 1, A.shape, generator=g).float())
        B = B * torch.exp2(torch.randint(-spread, spread + 1, B.shape, generator=g).float())
    C = _gemm_bx6(A.to(gpu).contiguous(), B.to(gpu).contiguous()).cpu().double()
    C64 = A.double() @ B.double()
    scale = A.double().abs() @ B.double().abs()
    err = float(((C - C64).abs() / scale).max())
    err32 = float((((A.to(gpu) @ B.to(gpu)).cpu().double() - C64).abs() / scale).max())
    # the kernel's own accumulation order in plain fp32: k-blocks of 16 added to a running fp32 sum one after the other
    seq = torch.zeros(M, N)
    for kb in range(0, K, 16):
        seq = seq + A[:, kb:kb + 16] @ B[kb:kb + 16]
    errseq = float(((seq.double() - C64).abs() / scale).max())
    print(f"M={M} N={N} K={K} spread=2^+-{spread}: bf16x6 {err:.3e} (2^{np.log2(max(err, 1e-300)):.1f}), "
          f"fp32 in the same k-block order {errseq:.3e}, torch fp32 matmul (tree) {err32:.3e}")
    # (the matrix core's own fp32 adder is not an fmaf chain: on such operands it is measured at ~2-3x the sequential
    # fp32 sum's error, i.e. 2^-21 of sum |a||b| at K = 4096 -- four thousand times inside the a-priori bound K 2^-24)
    assert err <= (2.0 ** -22 if spread == 0 else min(2.0 ** -20, 4.0 * errseq)), (err, errseq, err32)


def test_split_gemm_reproduces_exactly_representable_products(gpu):
    """Small integers: every partial product and sum is exact in fp32, so the result must be exact too (a term dropped
    or a fragment element misplaced shows up as an integer error)."""
    g = torch.Generator().manual_seed(7)
    A = torch.randint(-64, 65, (64, 128), generator=g).float()
    B = torch.randint(-64, 65, (128, 96), generator=g).float()
    C = _gemm_bx6(A.to(gpu), B.to(gpu)).cpu()
    assert torch.equal(C, A @ B)
    # 24-bit mantissas: (2^12 + 1)^2-style operands need all three terms of both factors
    A = (torch.randint(-(1 << 23), 1 << 23, (32, 16), generator=g).float())
    B = torch.zeros(16, 32)
    B[torch.arange(16), torch.arange(16)] = 1.0        # identity in the first 16 columns: C[:, :16] = A exactly
    C = _gemm_bx6(A.to(gpu), B.to(gpu)).cpu()
    assert torch.equal(C[:, :16], A) and torch.count_nonzero(C[:, 16:]) == 0


def test_split_gemm_rejects_shapes_it_does_not_cover(gpu):
    with pytest.raises(NotImplementedError):
        _gemm_bx6(torch.zeros(33, 16, device=gpu), torch.zeros(16, 32, device=gpu))


def _f64_oracle(net, cfg, audio, steps):
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in net.state_dict().items()}
    with torch.no_grad():
        return own.wavenet_forward(sd64, cfg, audio.double(), steps, return_pre_final=True)


@pytest.mark.parametrize("name", ["wn_c128", "wn_h128_d30", "wn_h256_d36"])
def test_bf16x6_error_against_float64_is_that_of_the_f32_path(gpu, name):
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
    ref, ref_pre = cases.cached(("wavenet_f64", name), lambda: _f64_oracle(net, cfg, audio, steps))   # same seeds in both split test files
    out = {}
    with torch.no_grad():
        for prec in ("f32", "bf16x6", "bf16x3"):
            net.set_option("precision", prec)
            eps = net((audio.to(gpu), steps.to(gpu)))
            pre = net.read_tap("pre_final", (B, cfg["skip_channels"], L))
            out[prec] = (eps.cpu(), pre.cpu())
        net.set_option("precision", "f32")
        again = net((audio.to(gpu), steps.to(gpu))).cpu()
    assert torch.equal(again, out["f32"][0])                 # switching back restores the f32 path bit for bit
    assert not torch.equal(out["bf16x6"][0], out["f32"][0])  # and the split path really is another arithmetic
    e = {p: (rel_err(out[p][0], ref), rel_err(out[p][1], ref_pre)) for p in out}
    rms = {p: float(((out[p][1].double() - ref_pre) ** 2).mean().sqrt() / (ref_pre ** 2).mean().sqrt()) for p in out}
    print(f"{name}: max-rel error vs float64 (eps, pre_final) f32-MFMA {e['f32'][0]:.3e} {e['f32'][1]:.3e} | "
          f"bf16x6 {e['bf16x6'][0]:.3e} {e['bf16x6'][1]:.3e} | bf16x3 (2-term) {e['bf16x3'][0]:.3e} {e['bf16x3'][1]:.3e}; "
          f"rms-rel pre_final f32 {rms['f32']:.3e} bf16x6 {rms['bf16x6']:.3e} bf16x3 {rms['bf16x3']:.3e}")
    for k in (0, 1):
        assert e["bf16x6"][k] <= 2.0 * e["f32"][k], (name, k, e)
    assert rms["bf16x6"] <= 2.0 * rms["f32"], (name, rms)
    # and against the reference's own fp32 forward (the golden fixture): inside the 1e-3 bound with the f32 path's margin
    g = load_golden("wavenet")
    assert rel_err(out["bf16x6"][0], g[f"{name}/eps"]) < REL_TOL / 100


@pytest.mark.parametrize("name", ["wn_c64", "wn_c128", "wn_h256_d36"])
def test_bf16x6_agrees_with_the_f32_winograd_path_at_every_staging_variant(gpu, name):
    """Same tile geometry as the f32 Winograd kernel: 16-byte LDS-DMA (L % 4 == 0, d >= 4), the contiguous-row form
    (d <= 16), the dword form and the per-lane epilogue (L % 4 != 0), positions past L in the last pair block, d > L."""
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES[name]
    net = cases.build_ours(cfg, wseed + 9).to(gpu)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for L2, B2 in ((1, 1), (63, 2), (600, 2), (1001, 1), (4096, 1), (4100, 2)):
        audio, steps = cases.wavenet_inputs(B2, L2, 1, iseed + L2)
        with torch.no_grad():
            net.set_option("precision", "f32")
            w = net((audio.to(gpu), steps.to(gpu)))
            net.set_option("precision", "bf16x6")
            s = net((audio.to(gpu), steps.to(gpu)))
            s2 = net((audio.to(gpu), steps.to(gpu)))
        assert torch.equal(s, s2)                                  # deterministic: no atomics in this path
        assert rel_err(s, w) < 1e-5, (name, L2, B2, rel_err(s, w))
        if L2 <= 1001 and name != "wn_h256_d36":
            with torch.no_grad():
                ref = own.wavenet_forward(sd, cfg, audio, steps)
            assert rel_err(s, ref) < REL_TOL / 100, (name, L2, B2)


@pytest.mark.parametrize("name", ["wn_cond_c64"])
def test_bf16x6_conditional_matches_reference(gpu, name):
    cfg, B, L, Tmel, wseed, iseed, store = cases.WAVENET_COND_CASES[name]
    g = load_golden("wavenet_cond")
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.set_option("precision", "bf16x6")
    audio, steps = cases.wavenet_inputs(B, L, 1, iseed)
    with torch.no_grad():
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed).to(gpu)
            eps = net((audio.to(gpu), steps.to(gpu)), mel_spec=mel)
            err = rel_err(eps, g[f"{name}/eps_bm{Bm}"])
            assert err < REL_TOL / 100, f"{name} Bm={Bm}: {err:.3e}"
        eps = net((audio.to(gpu), steps.to(gpu)))
        assert rel_err(eps, g[f"{name}/eps_nomel"]) < REL_TOL / 100


def test_bf16x6_sampler_graph_equals_the_per_step_loop(gpu):
    """The step-table / hipGraph sampler (`generate.py:23-55`) with the split layer: graph replay == eager launches,
    bit for bit, and the trajectory stays within the f32 path's distance of the oracle."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_c64"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    net.set_option("precision", "bf16x6")
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


def test_bf16x6_rejected_where_not_built_and_for_training(gpu):
    net = cases.build_ours(cases.WAVENET_CASES["wn_tiny"][0], 1).to(gpu)
    with pytest.raises(NotImplementedError):
        net.set_option("precision", "bf16x6")
