"""Multi-period / multi-scale discriminators on MI355X.

Drop-in for the reference's ``hifigan_modified/discriminators.py``: MultiPeriodDiscriminator :12-28,
MultiScaleDiscriminator :30-46, Discriminator2D :48-84 (zero right-pad + ``view(B,C,period,T//period)``
fold, 5x Conv2d 3x3), Discriminator1D :86-117 (AvgPool1d + 5x Conv1d k15), HiFiGANDiscriminators :119-151.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import functional as Fn

_CH = (1, 32, 64, 128, 256, 1)


class Discriminator2D(nn.Module):
    def __init__(self, period: int):
        super().__init__()
        self.period = period
        layers = []
        for i in range(5):
            layers.append(nn.Conv2d(_CH[i], _CH[i + 1], (3, 3), padding=(1, 1)))
            if i < 4:
                layers.append(nn.LeakyReLU(0.1))
        self.conv_layers = nn.Sequential(*layers)  # convs at 0,2,4,6,8 like the reference

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.disc2d(x, self)


class Discriminator1D(nn.Module):
    def __init__(self, scale: int):
        super().__init__()
        self.scale = scale
        self.downsample = nn.AvgPool1d(scale, stride=scale)
        layers = []
        for i in range(5):
            layers.append(nn.Conv1d(_CH[i], _CH[i + 1], 15, padding=7))
            if i < 4:
                layers.append(nn.LeakyReLU(0.1))
        self.conv_layers = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.disc1d(x, self)


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, periods: List[int] = [2, 3, 5, 7, 11]):
        super().__init__()
        self.periods = periods
        self.discriminators = nn.ModuleList([Discriminator2D(p) for p in periods])

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        return [d(x) for d in self.discriminators]


class MultiScaleDiscriminator(nn.Module):
    def __init__(self, scales: List[int] = [1, 2, 4]):
        super().__init__()
        self.scales = scales
        self.discriminators = nn.ModuleList([Discriminator1D(s) for s in scales])

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        return [d(x) for d in self.discriminators]


class HiFiGANDiscriminators(nn.Module):
    def __init__(self):
        super().__init__()
        self.mpd = MultiPeriodDiscriminator()
        self.msd = MultiScaleDiscriminator()

    def forward(self, real_audio: torch.Tensor, fake_audio: torch.Tensor) -> dict:
        # discriminator step (no gradient flows into either waveform): real and fake ride through every sub-discriminator as
        # ONE batch of 2B - half the launches, twice the grid per launch - and the outputs are split again.  Every op is
        # per-sample, so the values are those of the two separate passes of discriminators.py:127-151.
        if (real_audio.shape == fake_audio.shape and real_audio.dtype == fake_audio.dtype and real_audio.is_cuda
                and torch.is_grad_enabled() and not real_audio.requires_grad and not fake_audio.requires_grad):
            both = Fn.batch_cat(real_audio, fake_audio)
            B = real_audio.shape[0]
            mpd, msd = [Fn.batch_split(o) for o in self.mpd(both)], [Fn.batch_split(o) for o in self.msd(both)]
            return {"mpd_real": [r for r, _ in mpd], "mpd_fake": [f for _, f in mpd],
                    "msd_real": [r for r, _ in msd], "msd_fake": [f for _, f in msd]}
        return {
            "mpd_real": self.mpd(real_audio),
            "mpd_fake": self.mpd(fake_audio),
            "msd_real": self.msd(real_audio),
            "msd_fake": self.msd(fake_audio),
        }
