"""``train.py``-compatible driver on the HIP engine.

Same call surface and on-disk conventions as the reference (``train.py:27-196,224-252``): Hydra-style
config tree + ``key=value`` overrides, ``exp/<run>/checkpoint/<iter>.pkl`` holding
``{'model_state_dict', 'optimizer_state_dict'}``, resume from ``train.ckpt_iter`` (``max`` = newest),
Adam at ``train.learning_rate``, one process per GPU with ``apply_gradient_allreduce`` (RCCL), rank-0
checkpoints followed by an in-loop ``generate`` call.  wandb is replaced by a JSON-lines log
(``exp/<run>/train_log.jsonl``, same keys: ``train/loss``, ``train/log_loss``, ``train/loss_epoch``).

    python -m diffwave_sashimi_amd.train --config-dir /path/to/configs experiment=sc09 model=sashimi \
        dataset.data_path=/data/sc09 train.batch_size_per_gpu=32

Datasets: ``sc09`` (a directory tree of 1 s / 16 kHz ``*_nohash_*.wav`` files, ``dataloaders/sc.py:25-64``),
``ljspeech`` (``dataloaders/mel2samp.py:59-113``: random ``segment_length`` crops of the wavs under ``data_path``;
the log-mel of each batch is computed on the GPU by ``dws_mel_spectrogram`` right before the step instead of per
item on the loader's CPU workers) and ``synthetic`` (uniform noise clips, for smoke runs).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

from .generate import find_max_epoch, generate, load_config, local_directory

HASH_DIVIDER, EXCEPT_FOLDER = "_nohash_", "_background_noise_"   # `dataloaders/sc.py:15-16`


def fix_length(t, length):
    """``dataloaders/sc.py:25-32``: crop or zero-pad a [1, n] waveform to ``length``."""
    assert t.dim() == 2 and t.shape[0] == 1
    if t.shape[1] > length:
        return t[:, :length]
    if t.shape[1] < length:
        return torch.cat([t, torch.zeros(1, length - t.shape[1])], dim=1)
    return t


class SpeechCommands(torch.utils.data.Dataset):
    """``dataloaders/sc.py:46-64``: items are ``(waveform[1,16000] in [-1,1), sample_rate, label)``."""

    def __init__(self, path, length=16000):
        self._path, self._length = path, length
        walker = sorted(str(p) for p in Path(path).glob("**/*.wav"))
        self._walker = [w for w in walker if HASH_DIVIDER in w and EXCEPT_FOLDER not in w]

    def __getitem__(self, n):
        from scipy.io import wavfile
        f = self._walker[n]
        label = os.path.split(os.path.relpath(f, self._path))[0]
        sr, x = wavfile.read(f)
        if x.dtype == np.int16:            # torchaudio.load(normalize=True) semantics
            x = x.astype(np.float32) / 32768.0
        elif x.dtype == np.int32:
            x = x.astype(np.float32) / 2147483648.0
        else:
            x = x.astype(np.float32)
        if x.ndim == 2:
            x = x[:, 0]
        return fix_length(torch.from_numpy(x).unsqueeze(0), self._length), sr, label

    def __len__(self):
        return len(self._walker)


class LJSegments(torch.utils.data.Dataset):
    """The audio half of ``Mel2Samp`` (``dataloaders/mel2samp.py:59-113``): wavs found under ``data_path``
    (`files_to_list`, shuffled with ``random.seed(1234)``), sampling-rate check, a random ``segment_length`` crop
    (zero-padded when shorter); items are the raw-scale float waveform ``[segment_length]``."""

    def __init__(self, data_path, segment_length, sampling_rate, valid=False, **_ignored):
        import random
        self.audio_files = sorted(str(p) for p in Path(data_path).glob("**/*.wav"))
        random.seed(1234)
        random.shuffle(self.audio_files)
        self.segment_length, self.sampling_rate, self.valid = segment_length, sampling_rate, valid

    def __getitem__(self, index):
        import random
        from .mel import load_wav_to_torch
        audio, sr = load_wav_to_torch(self.audio_files[index])
        if sr != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sr, self.sampling_rate))
        if not self.valid:
            if audio.size(0) >= self.segment_length:
                start = random.randint(0, audio.size(0) - self.segment_length)
                audio = audio[start:start + self.segment_length]
            else:
                audio = torch.nn.functional.pad(audio, (0, self.segment_length - audio.size(0)), "constant").data
        return audio

    def __len__(self):
        return len(self.audio_files)


class SyntheticClips(torch.utils.data.Dataset):
    """U(-0.3, 0.3) clips (SURVEY.md 8d's synthetic workload) with the sc09 item layout."""

    def __init__(self, n_items, length, sampling_rate=16000, seed=0):
        self.n, self.length, self.sr, self.seed = n_items, length, sampling_rate, seed

    def __getitem__(self, n):
        g = torch.Generator().manual_seed(self.seed * 1000003 + n)
        return (torch.rand(1, self.length, generator=g) * 2 - 1) * 0.3, self.sr, "synthetic"

    def __len__(self):
        return self.n


def dataloader(dataset_cfg, batch_size, num_gpus, unconditional=True, rank=0, num_workers=4):
    """``dataloaders/__init__.py:6-33``."""
    name = dataset_cfg.get("_name_", "sc09")
    if name == "sc09":
        assert unconditional
        dataset = SpeechCommands(dataset_cfg["data_path"], dataset_cfg.get("segment_length", 16000))
    elif name == "synthetic":
        dataset = SyntheticClips(dataset_cfg.get("n_items", 64), dataset_cfg.get("segment_length", 16000),
                                 dataset_cfg.get("sampling_rate", 16000))
    elif name == "ljspeech":
        assert not unconditional
        dataset = LJSegments(**{k: v for k, v in dataset_cfg.items() if k != "_name_"})
    else:
        raise NotImplementedError(f"dataset '{name}' is not built (sc09, ljspeech, synthetic are)")
    sampler = None
    if num_gpus > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=num_gpus, rank=rank)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, shuffle=False,
                                       num_workers=num_workers, pin_memory=False, drop_last=True)


def optim_param_groups(net, honour_hints=False):
    """What the optimizer is built from.  Default: `net.parameters()`, i.e. ONE group -- the reference's `train.py:91` (and its
    checkpoints' `optimizer_state_dict`, `:104-107,159`).  `honour_hints=True` (`+train.honour_optim_hints=true`): the S4 kernel
    parameters carry `_optim = {"weight_decay": 0.0[, "lr": ...]}` (`models/s4.py:508-518`), and every distinct hint becomes its
    own parameter group after the plain one, the way the S4 authors' own training harness consumes them.  Under the reference's
    Adam (no weight decay) and SaShiMi's lr=None construction the update is the same; the group layout differs, so such a run's
    optimizer state does not load into a one-group run and vice versa."""
    params = list(net.parameters())
    if not honour_hints:
        return params
    plain = [p for p in params if not getattr(p, "_optim", None)]
    groups = [{"params": plain}]
    keys = []
    for p in params:
        h = getattr(p, "_optim", None)
        if not h:
            continue
        k = tuple(sorted(h.items()))
        if k not in keys:
            keys.append(k)
            groups.append({"params": [], **dict(k)})
        groups[1 + keys.index(k)]["params"].append(p)
    return groups


def train(rank, num_gpus, diffusion_cfg, model_cfg, dataset_cfg, generate_cfg, ckpt_iter, n_iters, iters_per_ckpt,
          iters_per_logging, learning_rate, batch_size_per_gpu, name=None, exp_root="exp", num_workers=4, precision=None,
          honour_optim_hints=False):
    """``train.py:49-196``."""
    from .distributed_util import apply_gradient_allreduce, reduce_tensor
    from .models import construct_model
    from .sampling import calc_diffusion_hyperparams
    from .training import training_loss

    local_path, checkpoint_directory = local_directory(name, model_cfg, diffusion_cfg, dataset_cfg, "checkpoint", exp_root)
    log_path = os.path.join(exp_root, local_path, "train_log.jsonl")
    dh = calc_diffusion_hyperparams(**diffusion_cfg, fast=False)
    trainloader = dataloader(dataset_cfg, batch_size_per_gpu, num_gpus, unconditional=model_cfg["unconditional"],
                             rank=rank, num_workers=num_workers)
    if len(trainloader) == 0:
        raise RuntimeError("the dataset holds fewer clips than one batch")
    print("Data loaded")
    net = construct_model(dict(model_cfg)).cuda().train()
    if precision not in (None, "f32"):      # `+engine.precision=bf16x6`: SaShiMi's GEMMs and weight gradients on the bf16 matrix cores
        net.set_option("precision", precision)
    print(f"{type(net).__name__} parameters: {sum(p.numel() for p in net.parameters()) / 1e6:.6f}M")   # `utils.py:76-88`
    if num_gpus > 1:
        net = apply_gradient_allreduce(net)
    learning_rate = float(learning_rate)
    groups = optim_param_groups(net, str(honour_optim_hints).lower() in ("1", "true"))
    own_lr = [isinstance(g, dict) and "lr" in g for g in groups] if isinstance(groups[0], dict) else [False]
    optimizer = torch.optim.Adam(groups, lr=learning_rate)

    if ckpt_iter == "max":
        ckpt_iter = find_max_epoch(checkpoint_directory)
    ckpt_iter = int(ckpt_iter)
    if ckpt_iter >= 0:
        try:
            checkpoint = torch.load(os.path.join(checkpoint_directory, f"{ckpt_iter}.pkl"), map_location="cpu")
            net.load_state_dict(checkpoint["model_state_dict"])
            if "optimizer_state_dict" in checkpoint:
                optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
                for g, own in zip(optimizer.param_groups, own_lr):    # `train.py:111-112` (one group there); a hinted lr stays
                    if not own:
                        g["lr"] = learning_rate
            print(f"Successfully loaded model at iteration {ckpt_iter}")
        except Exception as e:   # the reference swallows the error the same way (`train.py:115-117`)
            print(f"Model checkpoint found at iteration {ckpt_iter}, but was not successfully loaded ({e}) - "
                  "training from scratch.")
            ckpt_iter = -1
    else:
        print("No valid checkpoint model found - training from scratch.")
        ckpt_iter = -1

    def log(record, step):
        if rank == 0:
            with open(log_path, "a") as f:
                f.write(json.dumps(dict(record, step=step)) + "\n")

    stft = None
    if not model_cfg["unconditional"]:
        from .mel import MAX_WAV_VALUE, TacotronSTFT
        keys = dict(filter_length="filter_length", hop_length="hop_length", win_length="win_length",
                    sampling_rate="sampling_rate", mel_fmin="mel_fmin", mel_fmax="mel_fmax")
        stft = TacotronSTFT(**{k: dataset_cfg[v] for k, v in keys.items() if v in dataset_cfg})
    loss_fn = nn.MSELoss()
    n_iter = ckpt_iter + 1
    epoch = 0
    while n_iter < n_iters + 1:
        epoch_loss, n_batches = 0.0, 0
        if getattr(trainloader, "sampler", None) is not None and hasattr(trainloader.sampler, "set_epoch"):
            trainloader.sampler.set_epoch(epoch)
        for data in trainloader:
            if stft is None:
                audio, mel = data[0].cuda(), None
            else:   # `mel2samp.py:76-82,107-111`: mel of audio / MAX_WAV_VALUE, audio [B, 1, L] in [-1, 1]
                audio = (data.cuda() / MAX_WAV_VALUE).unsqueeze(1)
                mel = stft.mel_spectrogram(audio[:, 0])
            optimizer.zero_grad()
            loss = training_loss(net, loss_fn, audio, dh, mel_spec=mel)
            reduced_loss = reduce_tensor(loss.data, num_gpus).item() if num_gpus > 1 else loss.item()
            loss.backward()
            optimizer.step()
            epoch_loss += reduced_loss
            n_batches += 1
            if n_iter % iters_per_logging == 0:
                log({"train/loss": reduced_loss, "train/log_loss": float(np.log(reduced_loss))}, n_iter)
            if n_iter % iters_per_ckpt == 0 and rank == 0:
                torch.save({"model_state_dict": net.state_dict(), "optimizer_state_dict": optimizer.state_dict()},
                           os.path.join(checkpoint_directory, f"{n_iter}.pkl"))
                print(f"model at iteration {n_iter} is saved")
                if generate_cfg and generate_cfg.get("n_samples", 0):
                    if not model_cfg["unconditional"]:
                        assert generate_cfg.get("mel_name") is not None     # `train.py:170`
                    gen = dict(generate_cfg, ckpt_iter=n_iter)
                    net.eval()
                    wavs = []
                    generate(rank, diffusion_cfg, model_cfg, dataset_cfg, name=name, exp_root=exp_root,
                             written=wavs, **gen)
                    net.train()
                    log({"val/audio": wavs}, n_iter)     # the reference logs the clips to wandb (`train.py:180-183`)
            n_iter += 1
            if n_iter >= n_iters + 1:
                break
        epoch += 1
        if n_batches:
            log({"train/loss_epoch": epoch_loss / n_batches, "train/log_loss_epoch": float(np.log(epoch_loss / n_batches))},
                n_iter)
    return net


def distributed_train(rank, num_gpus, group_name, cfg, exp_root="exp"):
    """``train.py:27-47``."""
    from .distributed_util import init_distributed
    dist_cfg = dict(cfg.get("distributed") or {})
    if num_gpus > 1:
        torch.cuda.set_device(rank % torch.cuda.device_count())
        init_distributed(rank, num_gpus, group_name, dist_cfg.get("dist_backend", "nccl"),
                         dist_cfg.get("dist_url", "tcp://127.0.0.1:54321"))
    tr = dict(cfg["train"])
    tr.setdefault("precision", (cfg.get("engine") or {}).get("precision"))
    train(rank, num_gpus, dict(cfg["diffusion"]), dict(cfg["model"]), dict(cfg["dataset"]), dict(cfg.get("generate") or {}),
          exp_root=exp_root, **tr)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-dir", required=True, help="Hydra-style config tree (the reference's configs/)")
    ap.add_argument("--exp-root", default="exp")
    ap.add_argument("overrides", nargs="*", help="key=value overrides, e.g. experiment=sc09 model=sashimi")
    args = ap.parse_args(argv)
    cfg = load_config(args.config_dir, args.overrides)
    cfg.pop("wandb", None)
    os.makedirs(args.exp_root, mode=0o775, exist_ok=True)
    group = time.strftime("%Y%m%d-%H%M%S")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:      # launched by torch.distributed.run
        distributed_train(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), group, cfg, args.exp_root)
        return
    num_gpus = torch.cuda.device_count()
    if num_gpus <= 1:
        distributed_train(0, 1, group, cfg, args.exp_root)
    else:  # one process per GPU (`train.py:236-249`)
        import torch.multiprocessing as mp
        mp.spawn(distributed_train, args=(num_gpus, group, cfg, args.exp_root), nprocs=num_gpus, join=True)


if __name__ == "__main__":
    main(sys.argv[1:])
