"""Generate the committed golden vectors by importing the reference from
``/root/reference`` (build container only; see ``_refimport.py``).

    python tests/golden/make_golden.py [group ...]

Writes ``tests/golden/<group>.npz``.  Every group records inputs and the
reference's outputs (or the seeds that regenerate the inputs, see
``tests/cases.py``).  SaShiMi vectors are produced with the extension's
*symmetric* Cauchy semantics (``symmetric_cauchy = 1`` in the file)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import _refimport  # noqa: E402
from tests import cases  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


def sd_arrays(sd, prefix="sd/"):
    return {prefix + k: v for k, v in sd.items()}


def ref_model(cfg, state_dict):
    models, _, _, _ = _refimport.load()
    net = models.construct_model(dict(cfg)).eval()
    missing = net.load_state_dict(state_dict, strict=True)
    return net


# ---------------------------------------------------------------------------
def g_embedding():
    from models.utils import calc_diffusion_step_embedding as ref_emb
    t = torch.tensor([[0.], [1.], [7.], [49.], [199.]])
    save("embedding", t_float=t, emb_float=ref_emb(t, 128), t_int=t.long(), emb_int=ref_emb(t.long(), 128),
         emb_float_64=ref_emb(t, 64))


def g_schedule():
    _, _, utils, _ = _refimport.load()
    out = {}
    for tag, (T, b0, bT) in {"sc09": (200, 1e-4, 0.02), "ljspeech": (50, 1e-4, 0.05), "tiny": (6, 1e-4, 0.05)}.items():
        dh = utils.calc_diffusion_hyperparams(T, b0, bT, fast=True)
        for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
            out[f"{tag}/{k}"] = dh[k]
        out[f"{tag}/args"] = np.array([T, b0, bT], dtype=np.float64)
    # fast-sampling hook (`utils.py:136-138`)
    beta = [1e-4, 1e-3, 1e-2, 5e-2, 2e-1, 5e-1]
    dh = utils.calc_diffusion_hyperparams(200, 1e-4, 0.02, beta=beta, fast=True)
    out["fast/beta"] = np.array(beta, dtype=np.float64)
    for k in ("Beta", "Alpha", "Alpha_bar", "Sigma"):
        out[f"fast/{k}"] = dh[k]
    save("schedule", **out)


def g_wavenet():
    out = {}
    for name, (cfg, B, L, wseed, iseed, store) in cases.WAVENET_CASES.items():
        ours = cases.build_ours(cfg, wseed)
        sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
        net = ref_model(cfg, sd)   # strict load: proves state_dict key/shape compatibility
        audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
        with torch.no_grad():
            taps = {}
            h = net.final_conv[1].register_forward_hook(lambda m, i, o: taps.__setitem__("pre_final", o.detach()))
            eps = net((audio, steps))
            h.remove()
            eps_int = net((audio, steps.long()))
        assert torch.equal(eps, eps_int)
        out[f"{name}/eps"] = eps
        for k, v in cases.summarize(taps["pre_final"], stride=64).items():
            out[f"{name}/pre_final/{k}"] = v
        out[f"{name}/sd_digest"] = np.array([sum(float(v.double().sum()) for v in sd.values()),
                                             sum(float((v.double() ** 2).sum()) for v in sd.values())])
        if store:
            out.update(sd_arrays(sd, f"{name}/sd/"))
            out[f"{name}/audio"] = audio
            out[f"{name}/steps"] = steps
        print(name, "eps absmax", float(eps.abs().max()))
    save("wavenet", **out)


def g_wavenet_cond():
    out = {}
    for name, (cfg, B, L, Tmel, wseed, iseed, store) in cases.WAVENET_COND_CASES.items():
        ours = cases.build_ours(cfg, wseed)
        sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
        net = ref_model(cfg, sd)
        audio, steps = cases.wavenet_inputs(B, L, cfg["in_channels"], iseed)
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed)
            with torch.no_grad():
                eps = net((audio, steps), mel_spec=mel)
            out[f"{name}/eps_bm{Bm}"] = eps
            if store:
                out[f"{name}/mel_bm{Bm}"] = mel
        with torch.no_grad():
            out[f"{name}/eps_nomel"] = net((audio, steps))
        if store:
            out.update(sd_arrays(sd, f"{name}/sd/"))
            out[f"{name}/audio"] = audio
            out[f"{name}/steps"] = steps
    save("wavenet_cond", **out)


def g_sampler():
    """`generate.sampling` trajectories with recorded noise (T=6 and T=50 on the tiny net)."""
    _, generate, utils, _ = _refimport.load()
    cfg, B, L, wseed, iseed, _ = cases.WAVENET_CASES["wn_tiny"]
    ours = cases.build_ours(cfg, wseed)
    net = ref_model(cfg, {k: v.detach().clone() for k, v in ours.state_dict().items()})
    out = {}
    for tag, (T, b0, bT) in {"T6": (6, 1e-4, 0.05), "T50": (50, 1e-4, 0.05)}.items():
        dh = utils.calc_diffusion_hyperparams(T, b0, bT, fast=True)
        size = (B, 1, L)
        torch.manual_seed(1234)
        x_T = torch.normal(0, 1, size=size)
        noise = torch.zeros((T,) + size)
        for t in range(T - 1, 0, -1):  # the reference draws in this order (`generate.py:47,54`)
            noise[t] = torch.normal(0, 1, size=size)
        torch.manual_seed(1234)
        x0 = generate.sampling(net, size, dh)
        out[f"{tag}/x_T"] = x_T
        out[f"{tag}/noise"] = noise
        out[f"{tag}/x_0"] = x0
        out[f"{tag}/args"] = np.array([T, b0, bT], dtype=np.float64)
        print(tag, "x0 absmax", float(x0.abs().max()))
    save("sampler", **out)


def g_cauchy():
    """fp64 known answers from the reference's own formula + autograd
    (`extensions/cauchy/cauchy.py:19-26`, method of `test_cauchy.py:53-95`)."""
    import types
    stub = types.ModuleType("cauchy_mult")  # the compiled CUDA extension is not loadable here
    for n in ("cauchy_mult_fwd", "cauchy_mult_bwd", "cauchy_mult_sym_fwd", "cauchy_mult_sym_bwd"):
        setattr(stub, n, None)
    sys.modules["cauchy_mult"] = stub
    sys.path.insert(0, os.path.join(_refimport.REF, "extensions", "cauchy"))
    from cauchy import cauchy_mult_torch
    from oracle.cauchy import generate_data
    out = {}
    for N in (4, 16, 64):
        for L in (3, 17, 489, 1024):
            v_half, z, w_half = generate_data(4, N, L, symmetric=True, seed=2357)
            v = torch.cat([v_half, v_half.conj()], dim=-1).cdouble().requires_grad_(True)
            w = torch.cat([w_half, w_half.conj()], dim=-1).cdouble().requires_grad_(True)
            o = cauchy_mult_torch(v, z.cdouble(), w, symmetric=True)
            g = torch.Generator().manual_seed(99)
            dout = torch.randn(o.shape, dtype=torch.complex64, generator=g)
            dv, dw = torch.autograd.grad(o, (v, w), dout.cdouble())
            tag = f"sym/N{N}_L{L}"
            out[f"{tag}/v_half"], out[f"{tag}/z"], out[f"{tag}/w_half"], out[f"{tag}/dout"] = v_half, z, w_half, dout
            out[f"{tag}/out"], out[f"{tag}/dv"], out[f"{tag}/dw"] = o.detach(), dv[:, :N // 2], dw[:, :N // 2]
            # non-symmetric on the same draw (full vectors)
            vf, zf, wf = generate_data(2, N, L, symmetric=False, seed=2357)
            vf64 = vf.cdouble().requires_grad_(True)
            wf64 = wf.cdouble().requires_grad_(True)
            of = cauchy_mult_torch(vf64, zf.cdouble(), wf64, symmetric=False)
            doutf = torch.randn(of.shape, dtype=torch.complex64, generator=g)
            dvf, dwf = torch.autograd.grad(of, (vf64, wf64), doutf.cdouble())
            tag = f"nonsym/N{N}_L{L}"
            out[f"{tag}/v"], out[f"{tag}/z"], out[f"{tag}/w"], out[f"{tag}/dout"] = vf, zf, wf, doutf
            out[f"{tag}/out"], out[f"{tag}/dv"], out[f"{tag}/dw"] = of.detach(), dvf, dwf
    save("cauchy", **out)


def _ss_run(cfg, B, wseed, iseed, mel=None):
    ours = cases.build_ours(cfg, wseed)
    sd0 = {k: v.detach().clone() for k, v in ours.state_dict().items()}   # raw: L buffers 0, C not yet C~
    net = ref_model(cfg, sd0)
    audio, steps = cases.wavenet_inputs(B, cfg["L"], cfg["in_channels"], iseed)
    taps = {}
    h = net.final_conv[1].register_forward_hook(lambda m, i, o: taps.__setitem__("pre_final", o.detach()))
    with torch.no_grad():
        eps = net((audio, steps), mel_spec=mel)      # first forward: _setup_C mutates C / L (s4.py:686-687)
    h.remove()
    sd1 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net, sd0, sd1, audio, steps, eps, taps["pre_final"]


def g_sashimi():
    out = {"symmetric_cauchy": np.array(1)}
    for name, (cfg, B, wseed, iseed, store) in cases.SASHIMI_CASES.items():
        net, sd0, sd1, audio, steps, eps, pre = _ss_run(cfg, B, wseed, iseed)
        out[f"{name}/eps"] = eps
        for k, v in cases.summarize(pre, stride=64).items():
            out[f"{name}/pre_final/{k}"] = v
        if store:
            out.update(sd_arrays(sd0, f"{name}/sd0/"))
            # after the warm-up forward only C and L differ
            for k in sd1:
                if k.endswith("kernel.kernel.C") or k.endswith("kernel.kernel.L"):
                    out[f"{name}/sd1/{k}"] = sd1[k]
            out[f"{name}/audio"], out[f"{name}/steps"] = audio, steps
            with torch.no_grad():
                # S4 convolution kernel of the first block and of the centre block, and a block output
                first = "d_layers.0" if cfg["unet"] else "c_layers.0"
                for pre_, mod in ((first, dict(net.named_modules())[first]), ("c_layers.0", net.c_layers[0])):
                    Ls = mod.layer.L
                    k, _ = mod.layer.kernel(L=Ls, rate=1.0)
                    out[f"{name}/k/{pre_}"] = k
        print(name, "eps absmax", float(eps.abs().max()))
    save("sashimi", **out)


def g_sashimi_cond():
    out = {"symmetric_cauchy": np.array(1)}
    for name, (cfg, B, Tmel, wseed, iseed, store) in cases.SASHIMI_COND_CASES.items():
        for Bm in (1, B):
            mel = cases.mel_inputs(Bm, Tmel, iseed)
            net, sd0, sd1, audio, steps, eps, pre = _ss_run(cfg, B, wseed, iseed, mel=mel)
            out[f"{name}/eps_bm{Bm}"] = eps
            if store:
                out[f"{name}/mel_bm{Bm}"] = mel
        with torch.no_grad():
            out[f"{name}/eps_nomel"] = net((audio, steps))
        if store:
            out.update(sd_arrays(sd0, f"{name}/sd0/"))
            out[f"{name}/audio"], out[f"{name}/steps"] = audio, steps
    save("sashimi_cond", **out)


def g_sashimi_varlen():
    """Variable-length calls on ONE reference module, in sequence (`s4.py:517-522,698-702,805,1387`):
    L (warm-up: _setup_C), L/2 (kernel truncated), 2L (double_length mutates C and L), then L again (kernel now
    generated at 2L and truncated).  The state_dict after the doubling is recorded."""
    cfg = cases.ss_cfg(d_model=8, n_layers=1, L=256, diffusion_step_embed_dim_mid=64)
    ours = cases.build_ours(cfg, 211)
    sd0 = {k: v.detach().clone() for k, v in ours.state_dict().items()}
    net = ref_model(cfg, sd0)
    out = {"symmetric_cauchy": np.array(1)}
    out.update(sd_arrays(sd0, "sd0/"))
    g = torch.Generator().manual_seed(212)
    for i, L_in in enumerate([256, 128, 512, 256]):
        audio = torch.randn(2, 1, L_in, generator=g)
        steps = torch.randint(0, 200, (2, 1), generator=g).float()
        with torch.no_grad():
            eps = net((audio, steps))
        out[f"call{i}/audio"], out[f"call{i}/steps"], out[f"call{i}/eps"] = audio, steps, eps
        sd = net.state_dict()
        out[f"call{i}/L"] = np.array([int(sd[k]) for k in sorted(sd) if k.endswith("kernel.kernel.L")])
        if i == 2:
            for k in sd:
                if k.endswith("kernel.kernel.C"):
                    out[f"after_doubling/{k}"] = sd[k].detach().clone()
        print("varlen call", i, L_in, "eps absmax", float(eps.abs().max()), "L buffers", out[f"call{i}/L"])
    save("sashimi_varlen", **out)


def g_sashimi_c4():
    """BASELINE config 4 at its own geometry (cases.SASHIMI_C4): mel [1 | B, 80, 63] -> 16128 upsampled frames truncated
    to 16000 / 4000 / 1000 per stage."""
    cfg, B, Tmel, wseed, iseed = cases.SASHIMI_C4
    out = {"symmetric_cauchy": np.array(1)}
    for Bm in (1, B):
        mel = cases.mel_inputs(Bm, Tmel, iseed)
        net, sd0, sd1, audio, steps, eps, pre = _ss_run(cfg, B, wseed, iseed, mel=mel)
        out[f"eps_bm{Bm}"] = eps
        for k, v in cases.summarize(pre, stride=64).items():
            out[f"pre_final_bm{Bm}/{k}"] = v
        print("c4 Bm", Bm, "eps absmax", float(eps.abs().max()))
    save("sashimi_c4", **out)


from make_golden_cases import GRAD_CASES, GRAD_CASES_D32, grad_slice  # noqa: E402


def g_grads_d32():
    g_grads(GRAD_CASES_D32, "grads_d32", grad_slice)


def g_grads(case_table=None, fname="grads", keep=lambda t: t):
    """Reference gradients of the training loss (`train.py:198-222`, restated inline: train.py itself pulls in wandb and
    the data loaders): loss = MSE(net((x_t, t), mel), z) with t, z drawn from the global RNG after manual_seed, backward
    through the imported reference modules (SaShiMi: after the first-forward `_setup_C`, through the S4 kernel
    generation with the symmetric Cauchy semantics).  Stores the initial state_dict, inputs, loss and every gradient."""
    _, _, utils, _ = _refimport.load()
    out = {"symmetric_cauchy": np.array(1)}
    for name, (cfg, B, L, Tmel) in (case_table or GRAD_CASES).items():
        ours = cases.build_ours(cfg, 311)
        sd0 = {k: v.detach().clone() for k, v in ours.state_dict().items()}
        net = ref_model(cfg, sd0).train()
        dh = utils.calc_diffusion_hyperparams(T=50, beta_0=1e-4, beta_T=0.05, beta=None, fast=False)
        g = torch.Generator().manual_seed(312)
        audio = torch.randn(B, 1, L, generator=g) * 0.3
        mel = None if Tmel is None else torch.cat([cases.mel_inputs(1, Tmel, 313 + i) for i in range(B)])
        torch.manual_seed(314)
        T_, Alpha_bar = dh["T"], dh["Alpha_bar"]
        steps = torch.randint(T_, size=(B, 1, 1))
        z = torch.normal(0, 1, size=audio.shape)
        x_t = torch.sqrt(Alpha_bar[steps]) * audio + torch.sqrt(1 - Alpha_bar[steps]) * z
        loss = torch.nn.MSELoss()(net((x_t, steps.view(B, 1)), mel_spec=mel), z)
        loss.backward()
        # the weights are NOT stored: tests rebuild them with cases.build_ours(cfg, 311) (seeded); a digest guards that
        out[f"{name}/sd0_digest"] = torch.stack([v.double().sum() for v in sd0.values() if v.is_floating_point()]).sum().reshape(1)
        out[f"{name}/audio"], out[f"{name}/loss"] = audio, loss.detach().reshape(1)
        if mel is not None:
            out[f"{name}/mel"] = mel
        n = 0
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = keep(torch.zeros_like(p) if p.grad is None else p.grad.detach())
            n += 1
        print(name, "loss", float(loss), "params with grads", n)
    save(fname, **out)


def g_s4_parts():
    """Function-level vectors: TransposedLN, DownPool/UpPool index maps, FF, setup_C."""
    models, _, _, s4 = _refimport.load()
    from models import sashimi as rs
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(2, 6, 40, generator=g) * 3 + 1.5
    ln = rs.TransposedLN(6)
    with torch.no_grad():
        ln.m.fill_(0.3); ln.s.fill_(1.7)
        out["ln/x"], out["ln/y"], out["ln/ms"] = x, ln(x), np.array([0.3, 1.7], dtype=np.float32)
        # pure index maps (integer exact): feed an arange through einops' rearrange
        ar = torch.arange(2 * 3 * 20, dtype=torch.float32).reshape(2, 3, 20)
        from einops import rearrange
        out["pool/x"] = ar
        out["pool/down_p4"] = rearrange(ar, '... h (l s) -> ... (h s) l', s=4)
        ar2 = torch.arange(2 * 12 * 5, dtype=torch.float32).reshape(2, 12, 5)
        out["pool/y"] = ar2
        out["pool/up_p4"] = rearrange(ar2, '... (h s) l -> ... h (l s)', s=4)
    save("s4_parts", **out)


def g_mel():
    """`dataloaders/stft.py` (TacotronSTFT) imported from the reference.  `librosa` is absent from this image, so
    a stand-in module supplies the two functions stft.py takes from it: `librosa.util.pad_center` (zero padding,
    trivial) and `librosa.filters.mel`.  The filterbank comes from an implementation INDEPENDENT of this repo:
    `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` (installed in this image; Hugging
    Face's own test-suite checks it against `librosa.filters.mel`, whose defaults htk=False / norm='slaney' it
    reproduces), rounded to float32 as librosa returns it.  So these vectors pin the whole chain including the
    filterbank (`filterbank_pinned = 1`, `mel_basis` recorded)."""
    import types
    from transformers.audio_utils import mel_filter_bank
    lib = types.ModuleType("librosa")
    lib.util = types.ModuleType("librosa.util")
    lib.filters = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1):
        n = data.shape[axis]
        lpad = (size - n) // 2
        return np.pad(data, (lpad, size - n - lpad))
    lib.util.pad_center = pad_center
    lib.util.tiny = lambda x: np.finfo(np.float32).tiny
    lib.filters.mel = lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None: np.ascontiguousarray(mel_filter_bank(
        n_fft // 2 + 1, n_mels, fmin, sr / 2.0 if fmax is None else fmax, sr, norm="slaney", mel_scale="slaney").T
    ).astype(np.float32)
    sys.modules["librosa"], sys.modules["librosa.util"], sys.modules["librosa.filters"] = lib, lib.util, lib.filters
    import importlib.util                      # the package __init__ pulls in torchvision / torchaudio: load the file itself
    spec = importlib.util.spec_from_file_location("ref_stft", os.path.join(_refimport.REF, "dataloaders", "stft.py"))
    ref_stft = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_stft)
    TacotronSTFT = ref_stft.TacotronSTFT
    out = {"filterbank_pinned": np.ones(1)}
    g = torch.Generator().manual_seed(77)
    cases_ = {"lj": dict(filter_length=1024, hop_length=256, win_length=1024, sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0),
              "small": dict(filter_length=256, hop_length=64, win_length=200, sampling_rate=16000, mel_fmin=50.0, mel_fmax=7000.0)}
    for name, kw in cases_.items():
        st = TacotronSTFT(**kw)
        T = 16000 if name == "lj" else 3001
        t = torch.arange(T) / kw["sampling_rate"]
        y = 0.5 * torch.sin(2 * np.pi * 440.0 * t) * torch.linspace(0, 1, T) + 0.1 * torch.randn(2, T, generator=g)
        y = y.clamp(-1, 1)
        y[1, T // 2:] = 0.0                                   # silence: exercises the clamp at 1e-5
        mag, _ = st.stft_fn.transform(y)
        out[f"{name}/y"], out[f"{name}/mel"], out[f"{name}/mag"] = y, st.mel_spectrogram(y), mag
        out[f"{name}/mel_basis"] = st.mel_basis
        out[f"{name}/cfg"] = np.array([kw["filter_length"], kw["hop_length"], kw["win_length"], kw["sampling_rate"],
                                       kw["mel_fmin"], kw["mel_fmax"]], dtype=np.float64)
    save("mel", **out)


GROUPS = {"sashimi_c4": g_sashimi_c4, "grads_d32": g_grads_d32, "grads": g_grads, "mel": g_mel, "sashimi_varlen": g_sashimi_varlen, "sashimi": g_sashimi, "sashimi_cond": g_sashimi_cond, "s4_parts": g_s4_parts, "cauchy": g_cauchy, "embedding": g_embedding, "schedule": g_schedule, "wavenet": g_wavenet,
          "wavenet_cond": g_wavenet_cond, "sampler": g_sampler}

if __name__ == "__main__":
    _refimport.load()
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        GROUPS[g]()
