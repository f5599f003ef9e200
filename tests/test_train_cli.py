"""The train.py-compatible driver: dataset conventions (CPU) and an end-to-end run with checkpoint,
in-loop generation and resume through the HIP engine (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_generate_cli import _tree


def _write_wav(path, x, sr=16000):
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(path), exist_ok=True)
    wavfile.write(path, sr, x)


def test_speechcommands_dataset_conventions(tmp_path):
    """`dataloaders/sc.py:25-64`: only `*_nohash_*.wav` outside `_background_noise_`, sorted, [-1,1) floats,
    cropped / zero-padded to 16000, label = folder."""
    from diffwave_sashimi_amd.train import SpeechCommands, fix_length
    rng = np.random.default_rng(0)
    short = (rng.integers(-2000, 2000, 12000)).astype(np.int16)
    long_ = (rng.integers(-2000, 2000, 20000)).astype(np.int16)
    _write_wav(str(tmp_path / "zero" / "abc_nohash_0.wav"), short)
    _write_wav(str(tmp_path / "one" / "def_nohash_1.wav"), long_)
    _write_wav(str(tmp_path / "_background_noise_" / "x_nohash_0.wav"), short)
    _write_wav(str(tmp_path / "one" / "other.wav"), short)
    ds = SpeechCommands(str(tmp_path))
    assert len(ds) == 2
    x, sr, label = ds[0]          # sorted: one/def... first
    assert label == "one" and sr == 16000 and x.shape == (1, 16000)
    assert torch.equal(x[0], torch.from_numpy(long_[:16000].astype(np.float32) / 32768.0))
    x, sr, label = ds[1]
    assert label == "zero" and torch.equal(x[0, :12000], torch.from_numpy(short.astype(np.float32) / 32768.0))
    assert float(x[0, 12000:].abs().max()) == 0.0
    assert fix_length(torch.ones(1, 5), 3).shape == (1, 3)


def test_dataloader_shards_like_distributed_sampler():
    from diffwave_sashimi_amd.train import dataloader
    cfg = {"_name_": "synthetic", "n_items": 16, "segment_length": 64}
    seen = []
    for rank in range(2):
        dl = dataloader(cfg, batch_size=2, num_gpus=2, rank=rank, num_workers=0)
        dl.sampler.set_epoch(0)
        assert len(dl) == 4
        seen.append(torch.cat([b[0] for b in dl]))
    assert seen[0].shape == (8, 1, 64)
    # the two ranks see disjoint clips (each clip is seeded by its index)
    a = {tuple(np.round(x.flatten()[:4].numpy(), 6)) for x in seen[0]}
    b = {tuple(np.round(x.flatten()[:4].numpy(), 6)) for x in seen[1]}
    assert not (a & b) and len(a | b) == 16
    with pytest.raises(NotImplementedError):
        dataloader({"_name_": "ljspeech"}, 2, 1, unconditional=False)


@pytest.mark.gpu
def test_train_checkpoint_generate_resume(tmp_path, gpu):
    from diffwave_sashimi_amd.generate import load_config, local_path_name
    from diffwave_sashimi_amd.train import train
    d = _tree(tmp_path / "configs")
    cfg = load_config(d, ["model=wavenet", "model.res_channels=64", "model.skip_channels=64", "model.num_res_layers=4",
                          "model.dilation_cycle=4", "model.in_channels=1", "model.out_channels=1",
                          "model.diffusion_step_embed_dim_in=128", "model.diffusion_step_embed_dim_mid=512",
                          "model.diffusion_step_embed_dim_out=512",
                          "dataset._name_=synthetic", "dataset.segment_length=2048", "dataset.n_items=8",
                          "diffusion.T=20"])
    exp = str(tmp_path / "exp")
    common = dict(diffusion_cfg={k: v for k, v in cfg["diffusion"].items() if k != "beta"}, model_cfg=cfg["model"],
                  dataset_cfg=cfg["dataset"], iters_per_ckpt=3, iters_per_logging=1, learning_rate=2e-3,
                  batch_size_per_gpu=4, exp_root=exp, num_workers=0)
    torch.manual_seed(0)
    train(0, 1, generate_cfg={"n_samples": 2, "batch_size": 2}, ckpt_iter=-1, n_iters=4, **common)
    run = local_path_name(None, cfg["model"], cfg["diffusion"], cfg["dataset"])
    ck = os.path.join(exp, run, "checkpoint")
    assert sorted(os.listdir(ck)) == ["0.pkl", "3.pkl"]
    saved = torch.load(os.path.join(ck, "3.pkl"), map_location="cpu")
    assert set(saved) == {"model_state_dict", "optimizer_state_dict"}
    assert os.path.exists(os.path.join(exp, run, "waveforms", "3", "0k_1.wav"))      # in-loop generation at iteration 3
    log = [json.loads(l) for l in open(os.path.join(exp, run, "train_log.jsonl"))]
    steps = [r["step"] for r in log if "train/loss" in r]
    assert steps == [0, 1, 2, 3, 4]
    losses = [r["train/loss"] for r in log if "train/loss" in r]
    assert all(np.isfinite(losses))
    # resume: picks up 3.pkl (model + Adam state) and continues at iteration 4
    train(0, 1, generate_cfg={}, ckpt_iter="max", n_iters=6, **common)
    log = [json.loads(l) for l in open(os.path.join(exp, run, "train_log.jsonl"))]
    steps = [r["step"] for r in log if "train/loss" in r]
    assert steps == [0, 1, 2, 3, 4, 4, 5, 6]
    assert sorted(os.listdir(ck)) == ["0.pkl", "3.pkl", "6.pkl"]
    resumed = torch.load(os.path.join(ck, "6.pkl"), map_location="cpu")
    assert resumed["optimizer_state_dict"]["state"][0]["step"] > saved["optimizer_state_dict"]["state"][0]["step"]
