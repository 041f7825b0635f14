"""Plain HiFi-GAN generator, V3 shape (BASELINE configs[0]: "HiFi-GAN V3 generator (no ODConv), 80-mel -> 22.05 kHz").

The reference reaches this network only through ``fairseq.models.text_to_speech.hifigan.Generator``
(``agent/tts/codehifigan.py:6``, ``agent/tts/vocoder.py:24``; fairseq==0.12.2 pinned in ``asr_bleu/requirements.txt:1``),
which is neither installed nor vendored: **parity unpinned** - the architecture below restates the published V3
configuration of Kong et al. 2020 (ResBlock2; upsample rates [8,8,4], kernels [16,16,8], 256 initial channels, resblock
kernels [3,5,7], dilations [[1,2],[2,6],[3,12]], LeakyReLU 0.1, default slope 0.01 before conv_post, k7 pre/post convs,
tanh, branch outputs averaged; weight norm folded as ``agent/tts/vocoder.py:45`` does at inference) and is checked against
this build's own CPU restatement (``oracle.plain_hifigan_forward``) only.  ``state_dict`` keys follow the published
implementation (``conv_pre``, ``ups.{i}``, ``resblocks.{j}.convs.{k}``, ``conv_post``).  Inference only.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops, _native as N
from . import functional as Fn


class _ResBlock2(nn.Module):
    def __init__(self, channels, kernel_size, dilations):
        super().__init__()
        self.kernel_size, self.dilations = kernel_size, list(dilations)
        self.convs = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, dilation=d, padding=(kernel_size - 1) * d // 2)
                                    for d in dilations])

    def forward(self, x):
        for c, d in zip(self.convs, self.dilations):
            a = ops.act(x, N.ACT_LRELU, 0.1)
            x = _conv_same(a, c, d, res=x)           # x + conv(lrelu(x))
        return x


def _conv_same(x, conv, dilation=1, res=None):
    """'same' Conv1d of an nn.Conv1d's parameters (+ residual): MFMA channels-last kernels for 16-bit storage,
    the generic direct kernel otherwise."""
    w, b = Fn._w(conv.weight, x), Fn._w(conv.bias, x)
    ks = w.shape[2]
    pad = (ks - 1) * dilation // 2
    if ops.mfma_conv1d_ok(x, w, 1, pad, dilation, 1):
        y = Fn.conv1d(x, conv.weight, conv.bias, padding=pad, dilation=dilation)
        return y if res is None else ops.act(y, N.ACT_NONE, 0.0, res=res)
    return ops.conv1d(x, w, b, None, 1, pad, dilation, 1, res=res)


class PlainHiFiGANGenerator(nn.Module):
    def __init__(self, n_mels=80, upsample_rates=(8, 8, 4), upsample_kernel_sizes=(16, 16, 8), upsample_initial_channel=256,
                 resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12))):
        super().__init__()
        self.upsample_rates = list(upsample_rates)
        self.upsample_factors = self.upsample_rates          # ChunkedVocoder / GraphedVocoder read this name
        self.num_kernels = len(resblock_kernel_sizes)
        ch = upsample_initial_channel
        self.conv_pre = nn.Conv1d(n_mels, ch, 7, padding=3)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(nn.ConvTranspose1d(ch // (2 ** i), ch // (2 ** (i + 1)), k, u, padding=(k - u) // 2))
            for ks, ds in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(_ResBlock2(ch // (2 ** (i + 1)), ks, ds))
        self.conv_post = nn.Conv1d(ch // (2 ** len(self.upsample_rates)), 1, 7, padding=3)

    @torch.no_grad()
    def forward(self, mel, speaker_emb=None, emotion_emb=None):
        """mel [B, n_mels, T] -> waveform [B, 1, T * prod(upsample_rates)].  The embeddings are accepted and ignored
        (an unconditioned vocoder), so the callers of the conditioned generator can swap this one in."""
        x = _conv_same(mel if mel.is_contiguous() else mel.contiguous(), self.conv_pre)
        for i, up in enumerate(self.ups):
            a = ops.act(x, N.ACT_LRELU, 0.1)
            k, u = up.kernel_size[0], up.stride[0]
            x = ops.conv_transpose1d(a, Fn._w(up.weight, a), Fn._w(up.bias, a), None, u, (k - u) // 2, 0, 1)
            xs = None
            for j in range(self.num_kernels):
                r = self.resblocks[i * self.num_kernels + j](x)
                xs = r if xs is None else ops.act(r, N.ACT_NONE, 0.0, res=xs)
            x = ops.scale_(xs, 1.0 / self.num_kernels)
        a = ops.act(x, N.ACT_LRELU, 0.01)
        cp = self.conv_post
        return ops.conv1d(a, Fn._w(cp.weight, a), Fn._w(cp.bias, a), None, 1, 3, 1, 1, N.ACT_TANH)
