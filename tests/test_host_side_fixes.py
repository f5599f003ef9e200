"""Host-side behaviours the reference has and a cache / default could silently break: YAML-1.2 floats in the config
composer (CPU); per-call sampling seeds, the conditioner cache, the sampler's schedule tables, forward/backward pairing
of the training path (GPU)."""
import os
import textwrap

import pytest
import torch
import torch.nn as nn

from tests import cases


def test_config_floats_without_a_dot_are_floats(tmp_path):
    """`configs/config.yaml` spells `learning_rate: 2e-4`; OmegaConf reads a float, PyYAML's YAML-1.1 resolver a str
    (-> `Adam(lr='2e-4')` TypeError).  Same for a command-line override."""
    from diffwave_sashimi_amd.generate import load_config
    (tmp_path / "config.yaml").write_text(textwrap.dedent("""\
        train:
          learning_rate: 2e-4
          n_iters: 1000
          name: null
        diffusion: {T: 200, beta_0: 1e-4, beta_T: 0.02}
        """))
    cfg = load_config(str(tmp_path))
    assert cfg["train"]["learning_rate"] == 2e-4 and isinstance(cfg["train"]["learning_rate"], float)
    assert cfg["diffusion"]["beta_0"] == 1e-4 and cfg["train"]["n_iters"] == 1000 and cfg["train"]["name"] is None
    cfg = load_config(str(tmp_path), ["train.learning_rate=3E-5", "train.name=run1e", "diffusion.T=50"])
    assert cfg["train"]["learning_rate"] == 3e-5 and cfg["train"]["name"] == "run1e" and cfg["diffusion"]["T"] == 50
    torch.optim.Adam([nn.Parameter(torch.zeros(1))], lr=cfg["train"]["learning_rate"])


@pytest.mark.gpu
def test_unseeded_sampling_calls_differ_and_manual_seed_governs(gpu):
    """The reference draws x_T / z from torch's generator, which advances (`generate.py:47,54`): successive batches of
    one `generate` run are different clips.  With `seed=None` each call takes a fresh Philox seed from that generator."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_tiny"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    dh = calc_diffusion_hyperparams(4, 1e-4, 0.05)
    torch.manual_seed(77)
    a, b = sampling(net, (B, 1, L), dh), sampling(net, (B, 1, L), dh)
    assert not torch.equal(a, b)
    torch.manual_seed(77)
    a2, b2 = sampling(net, (B, 1, L), dh), sampling(net, (B, 1, L), dh)
    assert torch.equal(a, a2) and torch.equal(b, b2)


@pytest.mark.gpu
def test_sampler_tables_follow_the_schedule_not_only_T(gpu):
    """Same T, different betas: `dws_sampler_steps` must not reuse the resident c1 / c2 / sigma tables."""
    import ctypes

    import numpy as np

    from diffwave_sashimi_amd import _lib
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    cfg, B, L, wseed, _, _ = cases.WAVENET_CASES["wn_tiny"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    lib = _lib.load()
    T = 6
    outs = {}
    for bT in (0.05, 0.2, 0.05):
        dh = calc_diffusion_hyperparams(T, 1e-4, bT)
        tabs = [np.ascontiguousarray(dh[k].numpy()) for k in ("Alpha", "Alpha_bar", "Sigma")]
        ptabs = [t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for t in tabs]
        x = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1)).to(gpu)
        net._sync_params()
        net._prepare(B, L)
        _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, T, 9, 1, _lib.current_stream()))
        torch.cuda.synchronize()
        ref = sampling(net, (B, 1, L), dh, x_T=torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1)), seed=9)
        assert torch.equal(x, ref), bT          # dws_sampler_run always uploads: the two entry points must agree
        outs.setdefault(bT, x.clone())
        assert torch.equal(outs[bT], x)
    assert not torch.equal(outs[0.05], outs[0.2])


@pytest.mark.gpu
def test_conditioner_cache_follows_the_mel_tensor(gpu):
    """A freed mel tensor's address used to be handed to the next one of the same shape (version counter 0 again) and
    matched the (pointer, version, shape) cache key: the engine then kept the previous utterance's conditioner.  The
    engine now remembers the tensor OBJECT (and thereby keeps it alive, so its address cannot be recycled).  Also:
    in-place edits of the same tensor, and `invalidate()` after a write the version counters cannot see."""
    cfg, B, Tmel, wseed, iseed, _ = cases.SASHIMI_COND_CASES["ss_cond_tiny"]
    net = cases.build_ours(cfg, wseed).to(gpu)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], 1, iseed)
    audio, steps = audio.to(gpu), steps.to(gpu)

    def run(mel):
        with torch.no_grad():
            return net((audio, steps), mel_spec=mel).clone()

    m1, m2 = cases.mel_inputs(B, Tmel, 1), cases.mel_inputs(B, Tmel, 2)
    want1, want2 = run(m1.to(gpu)), run(m2.to(gpu))
    assert not torch.equal(want1, want2)
    for _ in range(8):                      # alternate fresh same-shape tensors, dropping each after its call
        t = m1.to(gpu)
        p = t.data_ptr()
        assert torch.equal(run(t), want1)
        del t
        t = m2.to(gpu)
        assert t.data_ptr() != p            # the engine's reference keeps the previous mel's block out of the free list
        assert torch.equal(run(t), want2)
        del t
    t = m1.to(gpu)
    assert torch.equal(run(t), want1)
    t.copy_(m2.to(gpu))                     # in place: same object, version counter bumps
    assert torch.equal(run(t), want2)
    t.data.copy_(m1.to(gpu))                # through .data: invisible to the version counter ...
    net.invalidate()                        # ... so the caller says so
    assert torch.equal(run(t), want1)
    with torch.no_grad():                   # parameters: a .data write + invalidate() reaches the engine too
        w = net.state_dict()["final_conv.2.conv.bias"]
        w.data.add_(1.0)
    net.invalidate()
    assert torch.allclose(run(t), want1 + 1.0, atol=1e-6)


@pytest.mark.gpu
def test_backward_of_a_stale_training_forward_raises(gpu):
    """The engine keeps ONE forward's activations: a backward after another training forward ran must not silently use
    the wrong ones."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    cfg, B, L = cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=2, dilation_cycle=2), 2, 128
    net = cases.build_ours(cfg, 5).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    l1 = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))
    l2 = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(4))
    with pytest.raises(RuntimeError, match="ONE training forward"):
        l1.backward()
    l2.backward()                           # the latest forward is still valid
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_kernel_length_cache_follows_the_buffer():
    """`Sashimi._kernel_L` keeps a host copy of every S4 kernel's `L` buffer (reading it from the GPU is a blocking
    device-to-host copy, twice per block and call before).  The copy must follow in-place writes (`_setup_C`'s `fill_`,
    `load_state_dict`) and replaced buffers (`.to()`)."""
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES["ss_d64_short"]
    net = cases.build_ours(cfg, wseed)
    k = next(iter(net._blocks())).layer.kernel.kernel
    assert net._kernel_L(k) == int(k.L)
    k.L.fill_(123)
    assert net._kernel_L(k) == 123
    sd = net.state_dict()
    name = next(n for n in sd if n.endswith("kernel.kernel.L"))
    sd[name] = torch.full_like(sd[name], 77)
    net.load_state_dict(sd)
    assert net._kernel_L(k) == 77 == int(k.L)
    k.L = k.L.clone() + 1          # a new tensor object behind the same attribute
    assert net._kernel_L(k) == 78


def test_training_loss_staging_keeps_the_reference_rng_order():
    """`training_loss` draws steps and noise on the CPU generator (`train.py:198-222`); the pinned staging path used on
    the GPU must not change what is drawn.  On the CPU the stager is the identity: same seed -> same loss, and the
    generator ends in the same state as after the reference's two draws."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import _stage, training_loss
    dh = calc_diffusion_hyperparams(20, 1e-4, 0.05)
    audio = torch.rand(2, 1, 64) - 0.5
    net = lambda xs, mel_spec=None: xs[0] * 0.5
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    l1 = training_loss(net, nn.MSELoss(), audio, dh, generator=g1)
    steps = torch.randint(20, size=(2, 1, 1), generator=g2)
    z = torch.normal(0, 1, size=audio.shape, generator=g2)
    ab = dh["Alpha_bar"][steps]
    l2 = nn.MSELoss()((torch.sqrt(ab) * audio + torch.sqrt(1 - ab) * z) * 0.5, z)
    assert torch.equal(l1, l2) and torch.equal(g1.get_state(), g2.get_state())
    t = torch.arange(6.).view(2, 3)
    assert _stage(t, "cpu", "slot") is t or torch.equal(_stage(t, "cpu", "slot"), t)


@pytest.mark.gpu
@pytest.mark.parametrize("backbone", ["wavenet", "sashimi"])
@pytest.mark.parametrize("between", ["eval_forward", "sampling", "other_shape"])
def test_backward_after_an_interleaved_eval_forward_or_sampling_raises(gpu, backbone, between):
    """The engine's activations are shared by every kind of forward: an eval forward, a sampling run or a call at another
    shape between a training forward and its backward overwrites (or reallocates) them -- the backward must fail, on
    both sides of the C ABI, instead of returning gradients of the wrong activations."""
    import ctypes
    from diffwave_sashimi_amd import _lib
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams, sampling
    from diffwave_sashimi_amd.training import training_loss
    if backbone == "wavenet":
        cfg, B, L = cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=2, dilation_cycle=2), 2, 128
    else:
        cfg, B, L = cases.ss_cfg(d_model=8, n_layers=1, L=256, diffusion_step_embed_dim_mid=64), 2, 256
    net = cases.build_ours(cfg, 5).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = (torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1)) * 0.3).to(gpu)
    loss = training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3))
    if between == "eval_forward":
        with torch.no_grad():
            net((audio, torch.zeros(B, 1, device=gpu)))
    elif between == "sampling":
        sampling(net, (B, 1, L), calc_diffusion_hyperparams(2, 1e-4, 0.05), seed=1)
    else:
        with torch.no_grad():
            net((audio[:1], torch.zeros(1, 1, device=gpu)))
    with pytest.raises(RuntimeError, match="ONE training forward"):
        loss.backward()
    # and below the Python guard: the C ABI itself refuses (DWS_ERR_STATE)
    d = torch.zeros(B, 1, L, device=gpu)
    status = _lib.load().dws_model_backward(net._handle, d.data_ptr(), _lib.current_stream())
    assert status != _lib.DWS_OK and b"forward_train" in _lib.load().dws_last_error()
    # a fresh pair still works
    net.zero_grad()
    training_loss(net, nn.MSELoss(), audio, dh, generator=torch.Generator().manual_seed(3)).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_invalidate_forgets_the_cached_kernel_lengths():
    """`broadcast_state` writes `t.data` (int64 `L` buffers included) without touching the version counter;
    `invalidate()` is the documented escape hatch and must drop the host copies of `L` too."""
    cfg, B, wseed, iseed, _ = cases.SASHIMI_CASES["ss_tiny"]
    net = cases.build_ours(cfg, wseed)
    k = next(iter(net._blocks())).layer.kernel.kernel
    k.L.fill_(50)
    assert net._kernel_L(k) == 50
    k.L.data.copy_(torch.tensor(64))       # invisible to ._version
    assert net._kernel_L(k) == 50          # the stale copy: why invalidate() has to clear it
    net.invalidate()
    assert net._kernel_L(k) == 64


def test_build_notices_changed_compile_flags(monkeypatch):
    """`build.py` keeps the flags an object was compiled with in a sidecar: an experiment flag set through
    DWS_HIPCC_FLAGS_<stem> (or an edit of FILE_FLAGS) makes that object stale instead of silently reusing it."""
    from diffwave_sashimi_amd import build
    if not os.path.exists(build.LIB):
        pytest.skip("libdws.so not built")
    assert not build.needs_build()
    monkeypatch.setenv("DWS_HIPCC_FLAGS_common", "-DSOME_EXPERIMENT")
    assert build.needs_build()
    src = os.path.join(build.CSRC, "common.hip")
    assert build._cmd_changed("hipcc", src) and not build._cmd_changed("hipcc", os.path.join(build.CSRC, "api.hip"))


def test_build_staleness_follows_contents_not_mtimes(monkeypatch, tmp_path):
    """The built library travels to another machine inside a copied tree whose modification times say nothing: an object
    is stale when the CONTENTS of its source or of a header differ from what its sidecar recorded, and only then."""
    from diffwave_sashimi_amd import build
    if not os.path.exists(build.LIB):
        pytest.skip("libdws.so not built")
    assert not build.needs_build()
    src = os.path.join(build.CSRC, "common.hip")
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime, st.st_mtime + 10 ** 6))      # "newer" than the library: same bytes, nothing to do
        assert not build.needs_build()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    extra = tmp_path / "extra.h"
    extra.write_text("// a header that was not there when the objects were built\n")
    real = build._headers()
    monkeypatch.setattr(build, "_headers", lambda: real + [str(extra)])
    assert build.needs_build()
    assert all(build._cmd_changed("hipcc", s) for s in build.sources())


def test_s4_optimizer_hints_and_param_groups():
    """`OptimModule.register` (`models/s4.py:508-518`, called at `:634-638`; `sashimi.py:126` passes no lr) tags log_dt, B, P,
    inv_w_real, w_imag with `_optim = {"weight_decay": 0.0}`; C (`:631`) and everything else carry none.  The reference's
    `train.py:91` ignores the tags (one Adam group): that is `optim_param_groups`' default, so its optimizer state dicts load;
    `honour_hints=True` cuts one extra group per distinct hint, and under Adam without weight decay the step is the same."""
    from diffwave_sashimi_amd.train import optim_param_groups
    cfg = cases.ss_cfg(d_model=16, n_layers=1, L=256, diffusion_step_embed_dim_mid=32)
    net = cases.build_ours(cfg, 3)
    hinted = {k for k, p in net.named_parameters() if getattr(p, "_optim", None)}
    assert hinted and all(k.rsplit(".", 1)[1] in ("log_dt", "B", "P", "inv_w_real", "w_imag") and ".kernel.kernel." in k for k in hinted)
    n_blocks = sum(1 for k, _ in net.named_parameters() if k.endswith(".kernel.kernel.C"))
    assert len(hinted) == 5 * n_blocks
    assert all(p._optim == {"weight_decay": 0.0} for k, p in net.named_parameters() if k in hinted)
    plain = optim_param_groups(net)
    assert [id(p) for p in plain] == [id(p) for p in net.parameters()]
    groups = optim_param_groups(net, honour_hints=True)
    assert len(groups) == 2 and groups[1]["weight_decay"] == 0.0 and "lr" not in groups[1]
    ids = [id(p) for g in groups for p in g["params"]]
    assert sorted(ids) == sorted(id(p) for p in net.parameters()) and len(set(ids)) == len(ids)
    assert {id(p) for p in groups[1]["params"]} == {id(p) for k, p in net.named_parameters() if k in hinted}
    # one Adam step either way from the same gradients
    gen = torch.Generator().manual_seed(0)
    grads = [torch.randn(p.shape, generator=gen) for p in net.parameters()]
    start = [p.detach().clone() for p in net.parameters()]
    results = []
    for honour in (False, True):
        with torch.no_grad():
            for p, s in zip(net.parameters(), start):
                p.copy_(s)
        opt = torch.optim.Adam(optim_param_groups(net, honour), lr=2e-4)
        for p, g in zip(net.parameters(), grads):
            p.grad = g.clone()
        opt.step()
        results.append([p.detach().clone() for p in net.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*results))
    # a hint with its own learning rate becomes its own group and keeps that rate
    first = next(p for k, p in net.named_parameters() if k in hinted)
    first._optim = {"weight_decay": 0.0, "lr": 1e-3}
    groups = optim_param_groups(net, honour_hints=True)
    assert len(groups) == 3 and sum(1 for g in groups if g.get("lr") == 1e-3 and len(g["params"]) == 1) == 1
    opt = torch.optim.Adam(groups, lr=2e-4)
    assert sorted(g["lr"] for g in opt.param_groups) == [2e-4, 2e-4, 1e-3]
