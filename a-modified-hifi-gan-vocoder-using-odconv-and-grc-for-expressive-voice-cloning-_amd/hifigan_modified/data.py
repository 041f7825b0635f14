"""Wave -> mel front end and clip sampler (the step before the path).

The reference ships no vocoder dataset code; its only mel parameters are ``speaker_embedding/ecapa_tdnn.py:163-170``
(n_fft 1024, hop 256, win 1024, 80 mels) and ``configs/eval_config.yaml:23-29`` (fmin 0, fmax 8000).  The front end here
is the same HIP framing + DFT + mel kernel the mel/STFT loss uses (``mv_mel_loss`` with ``want_mel``), so training targets
and the loss agree bit for bit, and no CPU librosa stage sits in front of the GPU.
"""
from __future__ import annotations

import torch

from . import functional as Fn
from .mel import mel_filterbank


class MelFrontEnd:
    """log-mel [B, n_mels, T // hop] (fp32) of waveforms [B, 1, T] on the GPU."""

    def __init__(self, sample_rate=22050, n_fft=1024, hop=256, n_mels=80, fmin=0.0, fmax=8000.0, clamp=1e-5, device="cuda"):
        self.n_fft, self.hop, self.clamp = n_fft, hop, clamp
        self.fb = mel_filterbank(sample_rate, n_fft, n_mels, fmin, fmax, device=device)

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        if wave.dim() == 2:
            wave = wave.unsqueeze(1)
        return Fn.mel_spectrogram(wave, self.fb, self.n_fft, self.hop, self.clamp)


class ClipSampler:
    """Random fixed-length training clips from a list of utterances (1-D waveforms of any length, host or device memory):
    ``sample(B)`` -> (wave [B,1,clip], mel [B,n_mels,clip//hop]).  Each rank seeds its own generator
    (seed + rank: data-parallel ranks draw different clips); short utterances are zero-padded on the right."""

    def __init__(self, utterances, clip_samples=8192, front_end: MelFrontEnd | None = None, seed=0, rank=0, device="cuda",
                 dtype=torch.float32):
        if not utterances:
            raise ValueError("ClipSampler needs at least one utterance")
        self.utts = [u.reshape(-1) for u in utterances]
        self.clip = int(clip_samples)
        self.fe = front_end
        self.device, self.dtype = device, dtype
        self.gen = torch.Generator().manual_seed(int(seed) + int(rank))

    def draw(self, batch):
        """host-side index draw: [(utterance index, start sample)] - deterministic per (seed, rank)."""
        picks = []
        for _ in range(batch):
            i = int(torch.randint(len(self.utts), (1,), generator=self.gen))
            n = self.utts[i].numel()
            s = int(torch.randint(max(n - self.clip, 0) + 1, (1,), generator=self.gen))
            picks.append((i, s))
        return picks

    def sample(self, batch):
        wave = torch.zeros(batch, 1, self.clip, device=self.device, dtype=torch.float32)
        for r, (i, s) in enumerate(self.draw(batch)):
            seg = self.utts[i][s:s + self.clip]
            wave[r, 0, :seg.numel()] = seg.to(self.device, torch.float32)
        mel = self.fe(wave) if self.fe is not None else None
        return wave.to(self.dtype), mel
