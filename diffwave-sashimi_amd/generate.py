"""``generate.py``-compatible driver on the HIP engine.

Same call surface and on-disk conventions as the reference (``generate.py:58-231``,
``utils.py:23-45,96-116``): Hydra-style config tree + ``key=value`` overrides,
``exp/<run>/checkpoint/<iter>.pkl`` holding ``{'model_state_dict': ...}``, run-directory
naming, ``<iter//1000>k_<n_samples*rank+i>.wav`` float32 files written with
``scipy.io.wavfile.write``, one process per GPU with no communication.

    python -m diffwave_sashimi_amd.generate --config-dir /path/to/configs experiment=sc09 model=wavenet \
        generate.n_samples=16 generate.ckpt_iter=max

hydra / omegaconf are not needed: ``load_config`` implements the subset the reference's
config tree uses (defaults lists, ``# @package _global_`` experiment files, ``${a.b}``
interpolation, dotted overrides).
"""
import argparse
import copy
import os
import re
import sys
import time

import numpy as np
import torch
import yaml


# --------------------------------------------------------------------------- config
class _Loader(yaml.SafeLoader):
    """SafeLoader with the YAML-1.2 float grammar OmegaConf/Hydra use: PyYAML (YAML 1.1) reads ``2e-4`` -- the
    reference's ``train.learning_rate`` (``configs/config.yaml``) -- as a string."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:\d[\d_]*\.[\d_]*(?:[eE][-+]?\d+)?|\.[\d_]+(?:[eE][-+]?\d+)?|\d[\d_]*[eE][-+]?\d+"
               r"|[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))


def _yaml(text):
    return yaml.load(text, Loader=_Loader)


def _deep_merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _load_yaml(path):
    with open(path) as f:
        text = f.read()
    return (_yaml(text) or {}), ("@package _global_" in text.split("\n", 1)[0])


def _set_dotted(cfg, key, value):
    parts = key.split(".")
    d = cfg
    for p in parts[:-1]:
        d = d.setdefault(p, {})
    d[parts[-1]] = value


def _get_dotted(cfg, key):
    d = cfg
    for p in key.split("."):
        d = d[p]
    return d


def _resolve(cfg, node=None):
    node = cfg if node is None else node
    items = node.items() if isinstance(node, dict) else enumerate(node)
    for k, v in list(items):
        if isinstance(v, (dict, list)):
            _resolve(cfg, v)
        elif isinstance(v, str):
            m = re.fullmatch(r"\$\{([^}]+)\}", v.strip())
            if m:
                node[k] = _get_dotted(cfg, m.group(1))
    return cfg


def load_config(config_dir, overrides=(), config_name="config"):
    """Compose ``<config_dir>/<config_name>.yaml`` the way ``@hydra.main`` does for the reference's
    tree (``configs/config.yaml:1-31``): group selections (``experiment=ljspeech``, ``model=wavenet``)
    pick files, ``/group: name`` defaults inside a ``# @package _global_`` experiment file mount under
    ``group`` BEFORE the file's own keys (so ``experiment/ljspeech.yaml``'s ``model.unconditional: false``
    lands on whichever model file was chosen), then dotted overrides, then ``${a.b}`` interpolation."""
    groups, values = {}, []
    for ov in overrides:
        k, _, v = ov.lstrip("+").partition("=")
        if "." not in k and os.path.isdir(os.path.join(config_dir, k)):
            groups[k] = v
        else:
            values.append((k, _yaml(v)))
    root, _ = _load_yaml(os.path.join(config_dir, config_name + ".yaml"))
    cfg = {}

    def mount(group, name):
        sub, is_global = _load_yaml(os.path.join(config_dir, group, str(name) + ".yaml"))
        for d in sub.pop("defaults", []):
            if isinstance(d, dict):
                for g, n in d.items():
                    g = g.lstrip("/")
                    mount(g, groups.get(g, n))
        if is_global:
            _deep_merge(cfg, sub)
        else:
            _deep_merge(cfg.setdefault(group, {}), sub)

    defaults = root.pop("defaults", [])
    if "_self_" not in defaults:
        defaults = list(defaults) + ["_self_"]
    for d in defaults:
        if d == "_self_":
            _deep_merge(cfg, root)
        elif isinstance(d, dict):
            for g, n in d.items():
                mount(g, groups.get(g, n))
    for k, v in values:
        _set_dotted(cfg, k, v)
    return _resolve(cfg)


# --------------------------------------------------------------------------- run directories / checkpoints
def find_max_epoch(path):
    """``utils.py:23-45``: largest ``<n>.pkl`` in ``path`` (-1 if none)."""
    epoch = -1
    for f in os.listdir(path):
        if len(f) > 4 and f.endswith(".pkl"):
            try:
                epoch = max(epoch, int(f[:-4]))
            except ValueError:
                continue
    return epoch


def local_path_name(name, model_cfg, diffusion_cfg, dataset_cfg):
    """Run-directory name of ``utils.py:96-108``, e.g. ``wnet_h128_d30_T200_betaT0.02_uncond``."""
    from .models import model_identifier
    model_name = model_identifier(model_cfg)
    diffusion_name = f"_T{diffusion_cfg['T']}_betaT{diffusion_cfg['beta_T']}"
    data_name = "" if model_cfg["unconditional"] else f"_L{dataset_cfg['segment_length']}_hop{dataset_cfg['hop_length']}"
    local_path = model_name + diffusion_name + data_name + f"_{'uncond' if model_cfg['unconditional'] else 'cond'}"
    if name:
        local_path = name + "_" + local_path
    return local_path


def local_directory(name, model_cfg, diffusion_cfg, dataset_cfg, output_directory, root="exp"):
    local_path = local_path_name(name, model_cfg, diffusion_cfg, dataset_cfg)
    output_directory = os.path.join(root, local_path, output_directory)
    os.makedirs(output_directory, mode=0o775, exist_ok=True)
    return local_path, output_directory


def smooth_ckpt(path, min_ckpt, max_ckpt):
    """``utils.py:47-74,154-166`` (experimental in the reference): running arithmetic mean of the
    ``model_state_dict`` of every checkpoint with ``min_ckpt < iteration <= max_ckpt``."""
    ckpts = []
    for f in os.listdir(path):
        if len(f) > 4 and f.endswith(".pkl"):
            try:
                it = int(f[:-4])
            except ValueError:
                continue
            if min_ckpt < it <= max_ckpt:
                ckpts.append(it)
    state_dict = None
    for n, it in enumerate(sorted(ckpts)):
        model_path = os.path.join(path, f"{it}.pkl")
        try:
            sd = torch.load(model_path, map_location="cpu")["model_state_dict"]
        except Exception:
            raise Exception(f"No valid model found at iteration {it}, path {model_path}")
        state_dict = sd if state_dict is None else {k: (state_dict[k] * n + sd[k]) / (n + 1) for k in sd}
    return state_dict


# --------------------------------------------------------------------------- generate
@torch.no_grad()
def generate(rank, diffusion_cfg, model_cfg, dataset_cfg, ckpt_iter="max", n_samples=1, name=None, batch_size=None,
             ckpt_smooth=None, mel_path=None, mel_name=None, dataloader=None, exp_root="exp", seed=None,
             written=None, precision=None):
    """``generate.py:58-200``.  ``ckpt_iter`` may additionally be ``"init"``: seeded random weights
    (no checkpoint), for smoke runs without trained weights.  ``precision`` (not in the reference; CLI:
    ``+engine.precision=bf16x6|f16x3``): the engine's opt-in matrix arithmetic, see ``include/dws.h``."""
    from .models import construct_model
    from .sampling import calc_diffusion_hyperparams, sampling
    from scipy.io.wavfile import write as wavwrite

    if rank is not None and torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    rank = rank or 0
    local_path, output_directory = local_directory(name, model_cfg, diffusion_cfg, dataset_cfg, "waveforms", exp_root)
    dh = calc_diffusion_hyperparams(**diffusion_cfg, fast=True)
    model_kwargs = {k: v for k, v in model_cfg.items()}
    net = construct_model(model_kwargs).cuda().eval()
    if precision not in (None, "f32"):
        net.set_option("precision", precision)     # NotImplementedError where the engine has no such kernels for this model

    ckpt_path = os.path.join(exp_root, local_path, "checkpoint")
    if ckpt_iter == "init":
        ckpt_iter = 0
    else:
        if ckpt_iter == "max":
            ckpt_iter = find_max_epoch(ckpt_path)
        ckpt_iter = int(ckpt_iter)
        if ckpt_smooth is None:
            model_file = os.path.join(ckpt_path, f"{ckpt_iter}.pkl")
            try:
                checkpoint = torch.load(model_file, map_location="cpu")
                net.load_state_dict(checkpoint["model_state_dict"])
            except Exception as e:  # the reference raises a bare 'No valid model found' (`generate.py:110-112`)
                raise Exception(f"No valid model found ({model_file}: {e})")
        else:                       # `generate.py:113-115`: average of the checkpoints in (ckpt_smooth, ckpt_iter]
            state_dict = smooth_ckpt(ckpt_path, int(ckpt_smooth), ckpt_iter)
            if state_dict is None:
                raise Exception(f"No checkpoints in ({ckpt_smooth}, {ckpt_iter}] under {ckpt_path}")
            net.load_state_dict(state_dict)
    output_directory = os.path.join(output_directory, str(ckpt_iter))
    os.makedirs(output_directory, mode=0o775, exist_ok=True)

    if batch_size is None:
        batch_size = n_samples
    assert n_samples % batch_size == 0
    if mel_name is not None:
        if mel_path is not None:      # pre-generated spectrogram (`generate.py:135-141`)
            try:
                mel = torch.load(os.path.join(mel_path, f"{mel_name}.wav.pt")).unsqueeze(0).cuda()
            except Exception:
                raise Exception("No ground truth mel spectrogram found")
        else:                         # from the waveform (`generate.py:142-153`)
            from .mel import Mel2Samp, load_wav_to_torch
            keys = ("filter_length", "hop_length", "win_length", "sampling_rate", "mel_fmin", "mel_fmax")
            _mel = Mel2Samp(**{k: dataset_cfg[k] for k in keys if k in dataset_cfg})
            audio, sr = load_wav_to_torch(os.path.join(str(dataset_cfg["data_path"]), f"{mel_name}.wav"))
            mel = _mel.get_mel(audio).unsqueeze(0)
        audio_length = mel.shape[-1] * dataset_cfg["hop_length"]
    else:
        audio_length, mel = dataset_cfg["segment_length"], None

    t0 = time.perf_counter()
    out = []
    for i in range(n_samples // batch_size):
        s = None if seed is None else seed + 1000 * rank + i
        out.append(sampling(net, (batch_size, 1, audio_length), dh, condition=mel, seed=s))
    generated_audio = torch.cat(out, dim=0)
    torch.cuda.synchronize()
    print(f"generated {n_samples} samples shape {tuple(generated_audio.shape)} at iteration {ckpt_iter} in "
          f"{time.perf_counter() - t0:.1f} seconds")
    for i in range(n_samples):
        outfile = "{}k_{}.wav".format(ckpt_iter // 1000, n_samples * rank + i)   # `generate.py:189`
        wavwrite(os.path.join(output_directory, outfile), dataset_cfg["sampling_rate"],
                 generated_audio[i].squeeze().cpu().numpy().astype(np.float32))
        if written is not None:
            written.append(os.path.join(output_directory, outfile))
    return generated_audio


def _worker(rank, cfg, exp_root):
    gen = dict(cfg.get("generate", {}))
    gen.setdefault("precision", (cfg.get("engine") or {}).get("precision"))
    generate(rank, dict(cfg["diffusion"]), dict(cfg["model"]), dict(cfg["dataset"]), exp_root=exp_root, **gen)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-dir", required=True, help="Hydra-style config tree (the reference's configs/)")
    ap.add_argument("--exp-root", default="exp")
    ap.add_argument("overrides", nargs="*", help="key=value overrides, e.g. experiment=sc09 model=wavenet")
    args = ap.parse_args(argv)
    cfg = load_config(args.config_dir, args.overrides)
    num_gpus = torch.cuda.device_count()
    if num_gpus <= 1:
        _worker(0, cfg, args.exp_root)
    else:  # one process per GPU, no communication (`generate.py:217-227`)
        import torch.multiprocessing as mp
        mp.spawn(_worker, args=(cfg, args.exp_root), nprocs=num_gpus, join=True)


if __name__ == "__main__":
    main(sys.argv[1:])
