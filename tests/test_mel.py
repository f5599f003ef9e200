"""Mel front-end (`dataloaders/stft.py:196-244`): oracle vs vectors captured from the reference (CPU), the
filterbank restatement's structural properties (CPU), and the HIP kernel vs both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import mel as omel
from tests.conftest import load_golden, rel_err


def _cfg(g, name):
    fl, hop, win, sr, fmin, fmax = g[f"{name}/cfg"]
    return dict(filter_length=int(fl), hop_length=int(hop), win_length=int(win), sampling_rate=int(sr),
                mel_fmin=float(fmin), mel_fmax=float(fmax))


@pytest.mark.parametrize("name", ["lj", "small"])
def test_oracle_matches_reference_stft(name):
    g = load_golden("mel")
    kw = _cfg(g, name)
    y = torch.from_numpy(g[f"{name}/y"])
    mag = omel.stft_magnitude(y, kw["filter_length"], kw["hop_length"], kw["win_length"])
    assert mag.shape == g[f"{name}/mag"].shape
    assert rel_err(mag, torch.from_numpy(g[f"{name}/mag"])) < 1e-6
    mel = omel.mel_spectrogram(y, torch.from_numpy(g[f"{name}/mel_basis"]), kw["filter_length"], kw["hop_length"],
                               kw["win_length"])
    assert float((mel - torch.from_numpy(g[f"{name}/mel"])).abs().max()) < 1e-5
    assert float(mel.min()) == pytest.approx(np.log(1e-5), abs=1e-6)       # the silent half hits the clamp


@pytest.mark.parametrize("name", ["lj", "small"])
def test_filterbank_matches_independent_implementation(name):
    """``mel_basis`` in the fixture is what the reference's ``TacotronSTFT`` held when the vectors were generated with
    ``librosa.filters.mel`` supplied by ``transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')``
    -- an implementation independent of this repo (checked against librosa by its own maintainers).  Our host-side
    restatement must reproduce it to one float32 ulp; plus the published constants of Slaney's scale (Auditory
    Toolbox): linear 200/3 Hz per mel below 1 kHz, then 27 steps per factor 6.4."""
    from diffwave_sashimi_amd.mel import _hz_to_mel, _mel_to_hz, mel_filterbank
    g = load_golden("mel")
    assert int(g["filterbank_pinned"][0]) == 1
    kw = _cfg(g, name)
    fb = mel_filterbank(kw["sampling_rate"], kw["filter_length"], 80, kw["mel_fmin"], kw["mel_fmax"])
    ref = g[f"{name}/mel_basis"]
    assert fb.shape == ref.shape and fb.dtype == ref.dtype == np.float32
    assert np.abs(fb.astype(np.float64) - ref.astype(np.float64)).max() <= 2e-9
    assert float(_hz_to_mel(1000.0)) == pytest.approx(15.0, abs=1e-12)
    assert float(_hz_to_mel(6400.0)) == pytest.approx(42.0, abs=1e-9)            # 15 + 27
    assert float(_mel_to_hz(3.0)) == pytest.approx(200.0, abs=1e-9)
    assert float(_mel_to_hz(42.0)) == pytest.approx(6400.0, rel=1e-12)


def test_filterbank_structure():
    """Structure of the Slaney filterbank (its numbers are pinned by the test above):
    triangles on a mel-spaced grid, unit area per filter in Hz, linear below 1 kHz."""
    from diffwave_sashimi_amd.mel import _hz_to_mel, _mel_to_hz, mel_filterbank, padded_hann
    sr, n_fft, n_mels = 22050, 1024, 80
    fb = mel_filterbank(sr, n_fft, n_mels, 0.0, 8000.0)
    assert fb.shape == (80, 513) and fb.dtype == np.float32 and (fb >= 0).all()
    freqs = np.linspace(0, sr / 2, 513)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), n_mels + 2))
    assert np.allclose(_hz_to_mel(_mel_to_hz(np.linspace(0, 40, 50))), np.linspace(0, 40, 50))
    assert np.allclose(np.diff(edges[edges < 1000.0]), 200.0 / 3 * (np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), n_mels + 2)[1]), rtol=1e-9)
    for m in (0, 10, 40, 79):
        nz = np.nonzero(fb[m])[0]
        assert freqs[nz[0]] > edges[m] - 1e-6 and freqs[nz[-1]] < edges[m + 2] + 1e-6      # support = (f_m, f_{m+2})
        peak = freqs[np.argmax(fb[m])]
        assert abs(peak - edges[m + 1]) <= (freqs[1] - freqs[0])                           # apex at the centre edge
    # slaney norm: each continuous triangle has unit area; wide filters approximate that on the FFT grid
    area = (fb[60:] * (freqs[1] - freqs[0])).sum(axis=1)
    assert np.allclose(area, 1.0, atol=0.03)
    assert fb[:, freqs > 8000.0 + (freqs[1] - freqs[0])].sum() == 0
    w = padded_hann(200, 256)
    assert w.shape == (256,) and w[:28].sum() == 0 and w[228:].sum() == 0 and w[28] == 0 and abs(w[128] - 1) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lj", "small"])
def test_hip_mel_matches_reference(gpu, name):
    from diffwave_sashimi_amd.mel import TacotronSTFT
    g = load_golden("mel")
    kw = _cfg(g, name)
    st = TacotronSTFT(**kw)
    assert np.allclose(st.mel_basis.numpy(), g[f"{name}/mel_basis"], rtol=0, atol=2e-9)   # one float32 ulp at 0.02
    y = torch.from_numpy(g[f"{name}/y"]).to(gpu)
    mel = st.mel_spectrogram(y).cpu()
    ref = torch.from_numpy(g[f"{name}/mel"])
    assert mel.shape == ref.shape
    # log-domain: 1e-3 relative on the mel energies = 1e-3 absolute on their logs
    assert float((mel - ref).abs().max()) < 1e-3
    with pytest.raises(RuntimeError):
        st.mel_spectrogram(y.cpu())


@pytest.mark.gpu
def test_hip_mel_edge_shapes(gpu):
    """Frame count T//hop + 1 and the reflect padding at both ends, against the oracle, for ragged lengths."""
    from diffwave_sashimi_amd.mel import TacotronSTFT
    st = TacotronSTFT(filter_length=256, hop_length=64, win_length=256, sampling_rate=16000, mel_fmin=0.0, mel_fmax=8000.0)
    gen = torch.Generator().manual_seed(5)
    for T in (129, 255, 256, 1000, 4097):
        y = (torch.rand(3, T, generator=gen) * 2 - 1) * 0.8
        ref = omel.mel_spectrogram(y, st.mel_basis, 256, 64, 256)
        got = st.mel_spectrogram(y.to(gpu)).cpu()
        assert got.shape == ref.shape == (3, 80, T // 64 + 1)
        assert float((got - ref).abs().max()) < 1e-3, T
    with pytest.raises(RuntimeError):
        st.mel_spectrogram(torch.zeros(1, 100, device=gpu))     # T <= filter_length/2: reflect padding undefined
