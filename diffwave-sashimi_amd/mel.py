"""Mel-spectrogram front-end of the vocoding path on the HIP engine.

Mirrors ``dataloaders/stft.py:196-244`` (``TacotronSTFT``) and ``dataloaders/mel2samp.py:45-82``
(``load_wav_to_torch``, ``Mel2Samp.get_mel``): reflect padding by ``filter_length/2``, Hann-windowed DFT
magnitudes at hop ``hop_length``, Slaney-normalised mel filterbank, ``log(clamp(., 1e-5))``.  The transform
runs in ``dws_mel_spectrogram`` (``csrc/mel_kernels.hip``); this module builds the two constant tables on
the host (window, filterbank) and keeps the reference's call surface.

The filterbank restates the published algorithm of ``librosa.filters.mel`` (``htk=False``,
``norm='slaney'``) -- librosa itself is not a dependency here; its numbers are pinned against an independent
implementation through ``tests/golden/mel.npz`` (``tests/test_mel.py``).
"""
import numpy as np
import torch

from . import _lib

MAX_WAV_VALUE = 32768.0     # `dataloaders/mel2samp.py:43`


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=80, fmin=0.0, fmax=None):
    """Slaney-style triangular mel filterbank, area-normalised: float32 ``[n_mels, n_fft//2 + 1]``."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def padded_hann(win_length, filter_length):
    """``scipy.signal.get_window('hann', win_length, fftbins=True)`` centre-padded to ``filter_length``
    (`stft.py:122-126`)."""
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    lpad = (filter_length - win_length) // 2
    return np.pad(w, (lpad, filter_length - win_length - lpad)).astype(np.float32)


class TacotronSTFT:
    """``dataloaders/stft.py:196-244``; ``mel_spectrogram(y)`` runs on the GPU ``y`` lives on."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        assert filter_length >= win_length
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax))
        self.window = torch.from_numpy(padded_hann(win_length, filter_length))
        self._dev = {}

    def mel_spectrogram(self, y):
        """``y`` [B, T] in [-1, 1] (cuda) -> log-mel [B, n_mel_channels, T // hop + 1]."""
        if y.device.type != "cuda":
            raise RuntimeError("libdws runs on the GPU only: move the waveform to cuda (there is no CPU fallback)")
        assert float(y.min()) >= -1 and float(y.max()) <= 1            # `stft.py:233-234`
        y = y.detach().to(torch.float32).contiguous()
        B, T = y.shape
        if y.device not in self._dev:
            self._dev[y.device] = (self.window.to(y.device), self.mel_basis.to(y.device))
        win, basis = self._dev[y.device]
        out = torch.empty(B, self.n_mel_channels, T // self.hop_length + 1, device=y.device, dtype=torch.float32)
        _lib.check(_lib.load().dws_mel_spectrogram(y.data_ptr(), B, T, win.data_ptr(), basis.data_ptr(),
                                                   self.filter_length, self.hop_length, self.n_mel_channels, 1e-5,
                                                   out.data_ptr(), _lib.current_stream()))
        return out


def load_wav_to_torch(full_path):
    """``dataloaders/mel2samp.py:51-56``: raw sample values as float32 (int16 range), and the rate."""
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.from_numpy(np.ascontiguousarray(data)).float(), sampling_rate


class Mel2Samp:
    """The ``get_mel`` half of ``dataloaders/mel2samp.py:59-82`` (the dataset half is not built)."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, sampling_rate=22050, mel_fmin=0.0,
                 mel_fmax=8000.0, **_ignored):
        self.stft = TacotronSTFT(filter_length=filter_length, hop_length=hop_length, win_length=win_length,
                                 sampling_rate=sampling_rate, mel_fmin=mel_fmin, mel_fmax=mel_fmax)
        self.sampling_rate = sampling_rate

    def get_mel(self, audio):
        audio_norm = (audio / MAX_WAV_VALUE).unsqueeze(0)
        return self.stft.mel_spectrogram(audio_norm.cuda()).squeeze(0)
