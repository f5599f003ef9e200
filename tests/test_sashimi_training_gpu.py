"""GPU: SaShiMi training path (forward_train + backward through the C ABI): the gradient of every
parameter of the `train.py:198-222` loss -- including the S4 kernel parameters C, B, P, inv_w_real,
w_imag, log_dt behind the Cauchy / Woodbury / FFT chain (`s4.py:704-807`) -- against torch autograd
through the CPU oracle on the same inputs."""
import pytest
import torch
import torch.nn as nn

from oracle import sashimi as oss
from tests import cases
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu

TRAIN_CASES = {
    # H = 32/64/128, L = 1024/256/64 (all through the M = 1024 fused FFT), unet skips on every level
    "d32": (cases.ss_cfg(d_model=32, n_layers=1, L=1024, diffusion_step_embed_dim_mid=64), 2),
    # snet: no blocks on the way down, skips only after UpPool; pool 2 and expand 2 with ff 1
    "snet": (cases.ss_cfg(d_model=32, n_layers=1, L=512, pool=[2, 2], ff=1, unet=False, diffusion_step_embed_dim_mid=64), 3),
    # BASELINE config 5 channel counts (H = 128/256/512), two blocks per level, L = 2048 (M = 2048 at the top)
    "d128": (cases.ss_cfg(d_model=128, n_layers=2, L=2048), 1),
}


def _engine_and_oracle(cfg, B, gpu, wseed, aseed, gseed, mel=None, start=0, tries=3, precision="f32", also=()):
    """Engine gradients, the oracle's fp32 autograd and its FLOAT64 autograd (the rounding-noise yardstick of
    tests/gradcheck.py) on the same weights, audio, steps and noise."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    from tests import gradcheck
    L = cfg["L"]
    net = cases.build_ours(cfg, wseed)
    net._setup_C()     # the first-forward mutation (s4.py:531-551) on the host: this state_dict is what a checkpoint holds
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    # inputs away from every ReLU kink (tests/gradcheck.py: smooth_case), so that the plain 1e-3 bound applies
    # (the oracle side of a seeded case -- input search, float64 and fp32 autograd -- is shared by the tests that use it)
    okey = None if mel is not None else ("sashimi_train_oracle", repr(sorted(cfg.items())), B, wseed, aseed, gseed, start, tries)
    make = lambda: gradcheck.smooth_case(cfg, sd, dh, B, L, mel, aseed, gseed, start=start, tries=tries)
    audio, gseed, loss_of, truth, kink, tried = make() if okey is None else cases.cached(okey, make)
    print(f"inputs: try {tried} (audio seed {aseed + 1000 * tried}), largest kink noise {max(kink.values()):.1e}")
    net = net.to(gpu).train()

    def engine_grads(prec):
        net.set_option("precision", prec)
        net.zero_grad(set_to_none=True)
        loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, mel_spec=None if mel is None else mel.to(gpu),
                             generator=torch.Generator().manual_seed(gseed))
        loss.backward()
        return loss, {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}

    others = {p: engine_grads(p)[1] for p in also}     # the same weights / inputs under other precisions (see the callers)
    loss, got = engine_grads(precision)
    net.extra_grads = others
    make32 = lambda: gradcheck.oracle_grads(cfg, sd, loss_of, torch.float32)
    loss32, o32 = make32() if okey is None else cases.cached(okey + ("fp32",), make32)
    o32 = {k: o32[k] for k in got}
    truth = {k: truth[k] for k in got}
    for k, gk in got.items():
        assert torch.isfinite(gk).all(), k
    return net, got, o32, truth, float(loss), loss32, kink


@pytest.mark.parametrize("name", list(TRAIN_CASES))
def test_sashimi_parameter_gradients_match_autograd(gpu, name):
    """Per tensor: 1e-3 of the tensor's largest gradient, widened to 3x the measured fp32 rounding noise only for the
    cancelling sums where the oracle's own fp32 autograd is further than that from its float64 evaluation."""
    from tests import gradcheck
    cfg, B = TRAIN_CASES[name]
    # where the search for kink-free inputs starts / how long it may go on (found once with tests/gradcheck.smooth_case;
    # d128: 390 k ReLU inputs, none of six tries is kink-free, so the first one is taken)
    start, tries = {"d32": (2, 2), "d128": (0, 1)}.get(name, (0, 3))
    net, got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, gpu, 15, 19, 23, start=start, tries=tries)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    worst, worst_k = gradcheck.compare(got, o32, truth, label=name, kink=kink)
    e64 = gradcheck.errors(got, truth)
    k64 = max(e64, key=e64.get)
    print(f"{name}: worst parameter-gradient rel err vs oracle fp32 {worst:.3e} ({worst_k}); vs float64 {e64[k64]:.3e} ({k64})")


def test_sashimi_bf16x6_training_gradients_are_those_of_the_f32_path(gpu):
    """precision="bf16x6" in training: the pointwise GEMMs of forward_train / backward (`tapconv_mfma_kernel<.., SPLIT>`) and
    the weight gradients (`wgrad_dma4_kernel<1>`) on the bf16 matrix cores with the exact 3-term split.  BASELINE config 5's
    channel counts (H = 128 / 256 / 512).  Same rule as the f32 path (tests/gradcheck.py); and against FLOAT64: the worst tensor
    is no further than 2 x the f32 path's worst, the median tensor within 1.5 x, and a single tensor exceeds 2 x its f32 error
    only inside 30 % of its 1e-3 bound or inside the f32 path's own worst tensor (the cancelling sums -- log_dt, LayerNorm
    scalars -- land anywhere inside their rounding noise under ANY change of summation order: 1.8e-4 against 2.5e-5 on one
    log_dt, 4.1e-4 against 7.1e-4 on the worst one; with the LayerNorm epilogues of round 6 one log_dt of the split path moved to
    4.0e-4 while the f32 path's worst log_dt sits at 4.5e-4)."""
    from tests import gradcheck
    cfg, B = TRAIN_CASES["d128"]
    net, got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, gpu, 15, 19, 23, start=0, tries=1, precision="bf16x6",
                                                                    also=("f32",))
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    worst, worst_k = gradcheck.compare(got, o32, truth, label="d128 bf16x6", kink=kink)
    f32 = net.extra_grads["f32"]
    e6, e32 = gradcheck.errors(got, truth), gradcheck.errors(f32, truth)
    assert any(not torch.equal(got[k], f32[k]) for k in got)          # the split kernels really ran
    k6, k32 = max(e6, key=e6.get), max(e32, key=e32.get)
    bad = {k: (e6[k], e32[k]) for k in e6 if e6[k] > max(2.0 * e32[k], 0.3 * gradcheck.TOL, e32[k32])}
    med = sorted(e6[k] / max(e32[k], 1e-12) for k in e6)[len(e6) // 2]
    print(f"d128: worst gradient error vs float64: bf16x6 {e6[k6]:.3e} ({k6}) | f32 {e32[k32]:.3e} ({k32}); "
          f"median ratio bf16x6/f32 {med:.2f}")
    assert not bad, bad
    assert e6[k6] <= 2.0 * e32[k32] and med <= 1.5


def test_sashimi_training_rejects_the_fp16_split(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B = TRAIN_CASES["d32"]
    net = cases.build_ours(cfg, 16).to(gpu).train()
    net.set_option("precision", "f16x3")
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, cfg["L"], generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    with pytest.raises(NotImplementedError):
        training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))


def test_sashimi_training_step_reduces_the_loss(gpu):
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B = TRAIN_CASES["d32"]
    net = cases.build_ours(cfg, 16).to(gpu).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, cfg["L"], generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    losses = []
    for it in range(8):
        opt.zero_grad()
        loss = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0] * 0.9, losses
    net.eval()
    with torch.no_grad():
        out = net((audio, torch.zeros(B, 1, device=gpu)))
    assert torch.isfinite(out).all()


def test_conditional_sashimi_gradients_match_autograd(gpu):
    """Mel-conditional SaShiMi training: every block's conditioner (two weight-normed ConvTranspose2d upsamplers +
    `mel_conv`, pooled stages take the first L_stage upsampled frames, `sashimi.py:160-175`) gets gradients."""
    from tests import gradcheck
    cfg = cases.ss_cfg(unconditional=False, d_model=32, n_layers=1, L=1024, mel_upsample=[16, 16],
                       diffusion_step_embed_dim_mid=64)
    B, Tmel = 2, 4
    mel = torch.cat([cases.mel_inputs(1, Tmel, 41 + i) for i in range(B)])
    # (the LeakyReLU upsamplers see thousands of mel values: no kink-free inputs within reach -- first try, widened bound)
    net, got, o32, truth, loss, ref_loss, kink = _engine_and_oracle(cfg, B, gpu, 35, 39, 43, mel=mel, start=0, tries=1)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    worst, worst_k = gradcheck.compare(got, o32, truth, label="conditional", kink=kink)
    seen_cond = sum(("upsample_conv2d" in k or "mel_conv" in k) and float(v.abs().max()) > 0 for k, v in o32.items())
    assert seen_cond >= 9 * 5          # 5 blocks x (2 upsamplers x (bias, g, v) + mel_conv (bias, g, v))
    print(f"conditional: worst parameter-gradient rel err {worst:.3e} ({worst_k})")


def test_full_length_stage_gradients_match_autograd(gpu):
    """The training path at the transform size BASELINE config 5 runs at: a top stage of H = 128 channels and L = 16000
    samples (M = 16384: the persistent `fftconv_kernel<14>` and its adjoint, `fftcorr_kernel<14>` with two 16-point groups
    per thread and the bins of U parked in its output slab, the Cauchy / Woodbury chain over 8001 frequencies), one block
    per level, a batch of two clips (the batch sums of the weight gradients and the per-clip step embeddings are in
    play); every parameter gradient against the oracle's autograd (float32, with float64 as the yardstick; a few
    minutes of CPU)."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    from tests import gradcheck
    cfg = cases.ss_cfg(d_model=128, n_layers=1, L=16000)
    B, L = 2, 16000
    net = cases.build_ours(cfg, 15).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(19)) * 0.3
    loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, generator=torch.Generator().manual_seed(23))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss_of = gradcheck.mse_training_loss(audio, dh, None, generator=torch.Generator().manual_seed(23))
    loss32, o32 = gradcheck.oracle_grads(cfg, sd, loss_of, torch.float32)
    _, truth = gradcheck.oracle_grads(cfg, sd, loss_of, torch.float64)
    assert abs(float(loss) - loss32) < 1e-4 * max(1.0, abs(loss32))
    o32, truth = {k: o32[k] for k in got}, {k: truth[k] for k in got}
    worst, worst_k = gradcheck.compare(got, o32, truth, label="L16000")
    e64 = gradcheck.errors(got, truth)
    k64 = max(e64, key=e64.get)
    print(f"L = 16000 stage: worst vs oracle fp32 {worst:.3e} ({worst_k}); vs float64 {e64[k64]:.3e} ({k64})")


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_layernorm_fused_into_the_training_gemms_changes_nothing(gpu, precision):
    """forward_train writes LN2(x1) out of the epilogue that produces x1 (`sashimi.py:177-179`: the 2H x H GEMM + GLU + residual)
    and the NEXT block's LN1(out) + fc_t(e) (`sashimi.py:148-152`) out of the block's last GEMM whenever one workgroup holds
    every channel of a column (H = 128: both; H = 256: LN1) -- `tapconv_mfma_kernel`'s LayerNorm epilogue.  Same loss and the
    same gradients as with the separate LayerNorm passes (`set_option("train_ln_fusion", "0")`), and fewer LayerNorm launches."""
    import ctypes
    from diffwave_sashimi_amd import _lib
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B = TRAIN_CASES["d128"]
    net = cases.build_ours(cfg, 15).to(gpu).train()
    net.set_option("precision", precision)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.rand(2, 1, cfg["L"], generator=torch.Generator().manual_seed(3)) * 2 - 1) * 0.3
    lib = _lib.load()

    def run(fused):
        net.set_option("train_ln_fusion", "1" if fused else "0")
        net.zero_grad(set_to_none=True)
        _lib.check(lib.dws_profile_enable(b"ln_kernel"))
        loss = training_loss(net, nn.MSELoss(), audio.to(gpu), dh, generator=torch.Generator().manual_seed(5))
        torch.cuda.synchronize()
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.dws_profile_query(ctypes.byref(n), ctypes.byref(ms)))
        lib.dws_profile_disable()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, n.value

    l1, g1, n1 = run(True)
    l0, g0, n0 = run(False)
    # d128, n_layers = 2: 10 blocks at H = 128 / 256 (two LayerNorms each, 2 of the blocks at H = 512) + the final norm; unfused: 21 launches.  Fused: both
    # norms of the H = 128 blocks except the LN1 behind a pooling layer, LN1 of every second H = 256 block
    print(f"{precision}: LayerNorm launches in forward_train {n0} -> {n1}; loss {l0:.7f} / {l1:.7f}")
    assert n0 == 21 and n1 <= n0 - 8, (n0, n1)
    assert abs(l1 - l0) <= 2e-6 * abs(l0)
    gmax = max(float(v.abs().max()) for v in g0.values())
    worst = max((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * gmax), k) for k in g0)
    print(f"worst gradient difference fused vs separate: {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 2e-4, worst


_KGROUP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn
from tests import cases
from tests.test_sashimi_training_gpu import TRAIN_CASES
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
from diffwave_sashimi_amd.training import training_loss
cfg, _ = TRAIN_CASES["d128"]
net = cases.build_ours(cfg, 15).cuda().train()
dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
audio = (torch.rand(2, 1, cfg["L"], generator=torch.Generator().manual_seed(3)) * 2 - 1) * 0.3
import ctypes
from diffwave_sashimi_amd import _lib
lib = _lib.load()
losses = []
for step in range(3):      # later steps commit again on the same buffers (group bookkeeping is per commit / per backward)
    net.zero_grad(set_to_none=True)
    if step == 2:          # which chain ran: Cauchy launches (forward at the commit, adjoint in the backward) of the last step
        _lib.check(lib.dws_profile_enable(b"cauchy_sym"))
    loss = training_loss(net, nn.MSELoss(), audio.cuda(), dh, generator=torch.Generator().manual_seed(5))
    loss.backward()
    losses.append(float(loss.detach()))
    if step < 2:
        with torch.no_grad():      # what an optimizer step does to the engine: new tensor versions -> upload + commit (same values)
            for p in net.parameters():
                p.mul_(1.0)
torch.cuda.synchronize()
n, ms = ctypes.c_int64(), ctypes.c_double()
_lib.check(lib.dws_profile_query(ctypes.byref(n), ctypes.byref(ms)))
lib.dws_profile_disable()
torch.save({"losses": losses, "cauchy_launches": n.value,
            "grads": {k: p.grad.detach().cpu() for k, p in net.named_parameters()}}, sys.argv[1])
'''


def test_stacked_kernel_generation_is_the_per_block_chain(tmp_path, gpu):
    """Training commits generate the S4 kernels of all blocks of one shape in ONE chain over n H rows (`s4.py:704-807`; `KGroup` in
    sashimi_model.hip) and run the chain's adjoint once per group.  Same loss and gradients as one chain per block
    (`DWS_S4_KERNELS_PER_BLOCK=1`, read once per process: two fresh processes) up to the rounding of rocFFT's batched transforms,
    on d128 (H = 128 / 256 / 512, two blocks per level and direction: groups of 4 / 4 / 2), over three steps."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    script = tmp_path / "kgroup_worker.py"
    script.write_text(_KGROUP_WORKER)
    out = {}
    for tag, extra in (("stacked", {}), ("per_block", {"DWS_S4_KERNELS_PER_BLOCK": "1"})):
        env = dict(os.environ, DWS_ROOT=ROOT, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("DWS_S4_KERNELS_PER_BLOCK", None)
        env.update(extra)
        f = tmp_path / (tag + ".pt")
        r = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = torch.load(f)
    a, b = out["stacked"], out["per_block"]
    # the switch took: one Cauchy launch per block and direction against one per group (the first training commit of a model is
    # always per block -- the Cauchy products are kept from the first forward_train on -- hence the count on a later step)
    print(f"Cauchy launches of one step: stacked {a['cauchy_launches']}, per block {b['cauchy_launches']}")
    assert 0 < a["cauchy_launches"] < b["cauchy_launches"], (a["cauchy_launches"], b["cauchy_launches"])
    assert len(set(b["losses"])) == 1, b["losses"]                                     # deterministic re-commit
    assert all(abs(x - a["losses"][0]) <= 2e-6 * abs(a["losses"][0]) for x in a["losses"]), a["losses"]   # per block, then stacked twice
    assert abs(a["losses"][0] - b["losses"][0]) <= 2e-6 * abs(b["losses"][0]), (a["losses"], b["losses"])
    g1, g0 = a["grads"], b["grads"]
    gmax = max(float(v.abs().max()) for v in g0.values())
    worst = max((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * gmax), k) for k in g0)
    kern = max((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * gmax), k) for k in g0 if ".kernel.kernel." in k or k.endswith("layer.D"))
    print(f"stacked vs per-block: loss {a['losses'][0]:.7f} / {b['losses'][0]:.7f}; worst gradient difference {worst[0]:.2e} ({worst[1]}), S4 parameters {kern[0]:.2e} ({kern[1]})")
    assert worst[0] < 2e-4, worst
