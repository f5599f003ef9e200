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
        dataloader({"_name_": "imagenet"}, 2, 1, unconditional=False)


def test_ljspeech_segments(tmp_path):
    """`dataloaders/mel2samp.py:84-111`: random crop of segment_length (zero pad when shorter), raw int16 scale."""
    from diffwave_sashimi_amd.train import dataloader
    rng = np.random.default_rng(1)
    _write_wav(str(tmp_path / "a.wav"), (rng.integers(-3000, 3000, 5000)).astype(np.int16), 22050)
    _write_wav(str(tmp_path / "b.wav"), (rng.integers(-3000, 3000, 900)).astype(np.int16), 22050)
    cfg = {"_name_": "ljspeech", "data_path": str(tmp_path), "segment_length": 2048, "sampling_rate": 22050,
           "filter_length": 1024, "hop_length": 256, "win_length": 1024, "mel_fmin": 0.0, "mel_fmax": 8000.0, "valid": False}
    dl = dataloader(cfg, batch_size=2, num_gpus=1, unconditional=False, num_workers=0)
    (batch,) = list(dl)
    assert batch.shape == (2, 2048) and batch.dtype == torch.float32
    assert float(batch.abs().max()) > 100          # raw sample values, scaled by 1/32768 only at the step
    short = batch[[float(r[900:].abs().max()) == 0 for r in batch].index(True)]
    assert float(short[:900].abs().max()) > 0
    bad = dict(cfg, sampling_rate=16000)
    with pytest.raises(ValueError):
        list(dataloader(bad, batch_size=2, num_gpus=1, unconditional=False, num_workers=0))


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


@pytest.mark.gpu
def test_train_conditional_from_wavs(tmp_path, gpu):
    """The ljspeech experiment end to end on tiny data: wav crops -> HIP mel front-end -> conditional training step
    -> checkpoint -> in-loop vocoding of `generate.mel_name` from its wav."""
    from diffwave_sashimi_amd.generate import local_path_name
    from diffwave_sashimi_amd.train import train
    rng = np.random.default_rng(2)
    data = tmp_path / "wavs"
    for i in range(4):
        _write_wav(str(data / f"LJ00{i}.wav"), (rng.standard_normal(3000) * 2500).astype(np.int16), 22050)
    model = dict(_name_="wavenet", unconditional=False, in_channels=1, out_channels=1, diffusion_step_embed_dim_in=128,
                 diffusion_step_embed_dim_mid=512, diffusion_step_embed_dim_out=512, res_channels=64, skip_channels=64,
                 num_res_layers=2, dilation_cycle=2, mel_upsample=[16, 16])
    ds = dict(_name_="ljspeech", data_path=str(data), segment_length=1024, sampling_rate=22050, filter_length=1024,
              hop_length=256, win_length=1024, mel_fmin=0.0, mel_fmax=8000.0, valid=False)
    diff = dict(T=10, beta_0=1e-4, beta_T=0.05)
    exp = str(tmp_path / "exp")
    train(0, 1, diff, model, ds, {"n_samples": 1, "mel_name": "LJ000"}, ckpt_iter=-1, n_iters=3, iters_per_ckpt=2,
          iters_per_logging=1, learning_rate=1e-3, batch_size_per_gpu=2, exp_root=exp, num_workers=0)
    run = local_path_name(None, model, diff, ds)
    assert sorted(os.listdir(os.path.join(exp, run, "checkpoint"))) == ["0.pkl", "2.pkl"]
    assert os.path.exists(os.path.join(exp, run, "waveforms", "2", "0k_0.wav"))
    log = [json.loads(l) for l in open(os.path.join(exp, run, "train_log.jsonl"))]
    assert all(np.isfinite(r["train/loss"]) for r in log if "train/loss" in r)
