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


class _ConvFused:
    """nn.Conv1d ('same', stride 1, any dilation) or nn.ConvTranspose1d through the fused channels-last MFMA kernel
    (mv_odconv_cl_fwd with ONE bank and alpha = 1; fp32 storage runs it with split bf16 operands, ~2^-16): the generic direct
    kernels took 1.4 ms per transposed conv and ~90 us per resblock conv at B = 1 x 344 frames - 6.2 ms per eager forward."""

    def __init__(self, conv):
        self.conv = conv
        self.transposed = isinstance(conv, nn.ConvTranspose1d)
        self._packed = {}
        self._ones = {}

    def geometry(self):
        c = self.conv
        return c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]

    def supported(self):
        c = self.conv
        cin, cout, ks, s, pad, dil = self.geometry()
        if cin % 8 or cout % 8 or c.groups != 1:
            return False
        if self.transposed:
            return (s * cout) % 16 == 0 and ks % s == 0 and dil == 1 and c.output_padding[0] == 0
        return cout % 16 == 0 and s == 1 and 2 * pad == dil * (ks - 1)

    def packed(self, dtype, device):
        from ctypes import c_void_p
        w = self.conv.weight
        ver = (w._version, w.data_ptr(), ops.param_epoch())
        hit = self._packed.get(dtype)
        if hit is not None and hit[0] == ver and hit[1].device == device:
            return hit[1]
        cin, cout, ks, s, pad, dil = self.geometry()
        tr = int(self.transposed)
        nbytes = N.lib().mv_odconv_cl_packed_bytes(cin, cout, ks, s, tr, 1, ops._DT[dtype])
        if nbytes == 0:
            raise RuntimeError("plain HiFi-GAN conv: unsupported geometry")
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        wd = w.detach().contiguous()                      # [Cout, Cin, ks] / [Cin, Cout, ks] = one bank
        N.call("mv_odconv_cl_pack", c_void_p(wd.data_ptr()), ops._DT[wd.dtype], c_void_p(buf.data_ptr()), cin, cout, ks, s, tr, 1,
               ops._DT[dtype], ops._stream())
        self._packed[dtype] = (ver, buf)
        return buf

    def __call__(self, x_cl, act=N.ACT_NONE, slope=0.1):
        """x_cl [B, T, Cin] channels-last -> act(conv(x)) [B, T', Cout] channels-last."""
        from ctypes import c_void_p
        cin, cout, ks, s, pad, dil = self.geometry()
        B, Tin, _ = x_cl.shape
        Tout = (Tin - 1) * s - 2 * pad + ks if self.transposed else Tin
        key = (B, x_cl.device)
        ones = self._ones.get(key)
        if ones is None:
            ones = self._ones[key] = torch.ones(B, 1, device=x_cl.device, dtype=torch.float32)
        y_cl = torch.empty(B, Tout, cout, device=x_cl.device, dtype=x_cl.dtype)
        P = lambda t: None if t is None else c_void_p(t.data_ptr())
        N.call("mv_odconv_cl_fwd", P(x_cl), P(self.packed(x_cl.dtype, x_cl.device)), P(Fn._w(self.conv.bias, x_cl)), P(ones), None, 0,
               None, None, None, 0, P(y_cl), None, B, cin, Tin, cout, Tout, ks, s, pad, dil, int(self.transposed), 1, int(act),
               float(slope), ops._dt(x_cl), ops._stream())
        return y_cl


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

    def _fused(self, conv):
        """The _ConvFused of one of this generator's conv modules (None: geometry outside the fused kernel's envelope)."""
        cache = self.__dict__.setdefault("_mv_conv_fused", {})
        f = cache.get(id(conv), 0)
        if f == 0:
            f = _ConvFused(conv)
            f = cache[id(conv)] = f if f.supported() else None
        return f

    def _up_fused(self, i):
        return self._fused(self.ups[i])

    def fused_supported(self):
        mods = [self.conv_pre] + list(self.ups) + [c for rb in self.resblocks for c in rb.convs]
        return all(self._fused(m) is not None for m in mods)

    def graphed(self, mel):
        """One captured forward for a fixed input shape: returns a callable that replays it on the current contents of the
        static buffer (`.mel`) and returns the static waveform.  ~80 short launches are host-bound when issued eagerly."""
        from .graphs import GraphedVocoder
        gv = GraphedVocoder(self, mel)

        def replay():
            return gv.replay()
        replay.mel = gv.mel
        return replay

    def _forward_cl(self, mel):
        """Channels-last pipeline: every convolution on the fused MFMA kernel, activations as the producing conv's epilogue where
        the raw value is not needed again, the residual adds / averages as elementwise launches (layout-agnostic)."""
        x = self._fused(self.conv_pre)(ops.nct_to_ntc(mel), N.ACT_LRELU, 0.1)      # lrelu(conv_pre(mel)): only the activated value is used
        a = x
        for i, up in enumerate(self.ups):
            x = self._fused(up)(a)                                                  # raw: the resblocks add it back
            a0 = ops.act(x, N.ACT_LRELU, 0.1)                                       # shared by the first conv of the three resblocks
            xs = None
            for j in range(self.num_kernels):
                rb = self.resblocks[i * self.num_kernels + j]
                r, ar = x, a0
                for q, c in enumerate(rb.convs):
                    y = self._fused(c)(ar)
                    r = ops.act(y, N.ACT_NONE, 0.0, res=r)                          # r + conv(lrelu(r))
                    if q + 1 < len(rb.convs):
                        ar = ops.act(r, N.ACT_LRELU, 0.1)
                xs = r if xs is None else ops.act(r, N.ACT_NONE, 0.0, res=xs)
            ops.scale_(xs, 1.0 / self.num_kernels)
            last = i + 1 == len(self.ups)
            a = ops.act(xs, N.ACT_LRELU, 0.01 if last else 0.1)
        cp = self.conv_post
        a = ops.ntc_to_nct(a)                                                       # conv_post has ONE output channel: generic direct kernel
        return ops.conv1d(a, Fn._w(cp.weight, a), Fn._w(cp.bias, a), None, 1, 3, 1, 1, N.ACT_TANH)

    @torch.no_grad()
    def forward(self, mel, speaker_emb=None, emotion_emb=None, force_generic=False):
        """mel [B, n_mels, T] -> waveform [B, 1, T * prod(upsample_rates)].  The embeddings are accepted and ignored
        (an unconditioned vocoder), so the callers of the conditioned generator can swap this one in."""
        mel = mel if mel.is_contiguous() else mel.contiguous()
        if not force_generic and self.fused_supported():
            return self._forward_cl(mel)
        x = _conv_same(mel, self.conv_pre)
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
