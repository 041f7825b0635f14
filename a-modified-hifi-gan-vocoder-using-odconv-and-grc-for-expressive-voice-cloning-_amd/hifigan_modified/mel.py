"""Mel front-end constants (host side): Slaney-scale triangular filterbank, as implied by the only mel parameters the
reference holds (speaker_embedding/ecapa_tdnn.py:163-170: librosa melspectrogram, n_fft 1024, hop 256, 80 mels).
The STFT / log-mel arithmetic itself is the HIP kernel mv_mel_loss (csrc/train.hip)."""
from __future__ import annotations

import math

import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    lin = f / f_sp
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0, device=None) -> torch.Tensor:
    """fp32 [n_mels, n_fft//2+1], Slaney area normalisation."""
    fmax = sr / 2 if fmax is None else fmax
    n_bins = n_fft // 2 + 1
    fft_f = np.linspace(0, sr / 2, n_bins)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(pts)
    ramps = pts[:, None] - fft_f[None, :]
    fb = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        fb[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    fb *= (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[:, None]
    t = torch.from_numpy(fb.astype(np.float32))
    return t.to(device) if device is not None else t
