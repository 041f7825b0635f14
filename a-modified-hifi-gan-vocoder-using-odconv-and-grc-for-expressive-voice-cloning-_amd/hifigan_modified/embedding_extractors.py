"""Conditioning producers on MI355X: ECAPA-TDNN speaker encoder and Emotion2Vec emotion encoder.

Drop-in for the reference's ``embedding_extractors.py`` (:13-100 ``ECAPA_TDNN``, :102-150 ``SE_Res2Block``, :152-170
``SE_Module``, :172-257 ``Emotion2Vec``, :259-284 ``EmbeddingExtractor``): same class names, constructor arguments, return
tuples and ``state_dict`` keys (the parameters live in the same torch container modules, which are never called).

How it runs (inference; activations channels-last [B][T][C] in the input's dtype):
  * every Conv1d / Linear over the sequence is one MFMA implicit-GEMM launch (``mv_dconv_cl_fwd``) with the following eval-mode
    BatchNorm folded into the packed weights and ReLU / tanh in the epilogue (fp32 storage: the generic fp32 kernels);
  * the pieces that are not convolutions are the kernels of ``csrc/embed.hip``: flash multi-head attention on MFMA,
    add+LayerNorm, SE squeeze / excitation / scale+residual, the Res2Net chain glue, attentive statistics pooling, L2 norm.

Deviations from the reference, on purpose:
  * ``ECAPA_TDNN.final_proj`` is ``Linear(6*hidden, embedding_dim)``: the reference's ``Linear(3*hidden, .)`` (:48) cannot take
    ``cat(mean, std)`` (:84-87) and its forward raises; 6*hidden is what the forward needs.
  * The producers are frozen feature extractors here: BatchNorm always uses its running statistics, dropout is the identity and
    nothing is differentiable (training the extractors is outside the vocoder hot path).  In ``.train()`` mode the extra logits of
    the reference's return tuples are still returned (classifier heads evaluated on the frozen embedding).
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _native as N
from . import ops

_RELU = (N.ACT_LRELU, 0.0)
_NONE = (N.ACT_NONE, 0.0)
_TANH = (N.ACT_TANH, 0.0)
_UNFUSED_RES2 = False   # tests: force the conv-by-conv Res2Net chain
_SKINNY_ROWS = int(__import__("os").environ.get("MV_SKINNY_ROWS", "4096"))     # up to this many positions the split-K GEMM beats the tiled conv kernel (tools/bench_embed.py)


def _sig(mod: nn.Module):
    return (ops.param_epoch(),) + tuple(t._version for t in mod.state_dict(keep_vars=True).values())


def _fold_bn(w: torch.Tensor, b: Optional[torch.Tensor], bn: Optional[nn.BatchNorm1d]):
    """Weights/bias of `bn(conv(x))` in eval mode as one affine map (fp32)."""
    w = w.detach().float()
    b = torch.zeros(w.shape[0], device=w.device) if b is None else b.detach().float()
    if bn is None:
        return w, b
    s = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    return w * s.view(-1, *([1] * (w.dim() - 1))), (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()


class _Affine:
    """One Conv1d(k, dilation, 'same') or Linear over a channels-last sequence, prepared once per weight version."""

    def __init__(self, w, b, bn, dtype, dilation=1, pad_in=False):
        w, b = _fold_bn(w, b, bn)
        if w.dim() == 2:
            w = w.unsqueeze(-1)
        self.cout, self.cin, self.ks = w.shape
        self.dil = dilation
        self.dtype = dtype
        self.mfma = (dtype in (torch.bfloat16, torch.float16) and self.ks in ops._MFMA_KS and self.cout % 32 == 0
                     and (self.ks - 1) * dilation <= 128 and (pad_in or self.cin % 32 == 0))
        if self.mfma:
            self.cinp = ops._up32(self.cin)
            self.packed = ops.dconv_pack(w.unsqueeze(2).contiguous(), dtype, 0, self.cout, self.cinp)
            self.bias = b.to(dtype)
        else:
            self.w, self.bias = w.to(dtype).contiguous(), b.to(dtype).contiguous()

    def __call__(self, x_cl: torch.Tensor, act=_NONE) -> torch.Tensor:
        kind, slope = act
        if self.mfma and x_cl.shape[-1] == self.cinp:
            if self.ks == 1:        # rows are independent: one long sequence fills the 128-position tiles whatever T is
                B, T, C = x_cl.shape
                if B * T <= _SKINNY_ROWS and self.cout % 64 == 0:
                    y = torch.empty(B, T, self.cout, device=x_cl.device, dtype=x_cl.dtype)
                    rc = N.lib().mv_gemm_cl_skinny(ops._p(x_cl), ops._p(self.packed), ops._p(self.bias), ops._p(y), B * T, C, self.cout,
                                                   kind, float(slope), ops._dt(x_cl), ops._stream())
                    if rc == 0:
                        return y
                    if rc != -3:
                        N.check(rc, "mv_gemm_cl_skinny")
                return ops.dconv_cl(x_cl.view(1, B * T, C), self.packed, self.bias, self.cout, 1, 1, 1, kind, slope).view(B, T, self.cout)
            return ops.dconv_cl(x_cl, self.packed, self.bias, self.cout, 1, self.ks, self.dil, kind, slope)
        if self.mfma:
            raise RuntimeError(f"channels-last input has {x_cl.shape[-1]} channels, packed weights expect {self.cinp}")
        B, T, C = x_cl.shape
        if self.ks == 1:
            y = ops.linear(x_cl.reshape(B * T, C), self.w.view(self.cout, self.cin), self.bias).view(B, T, self.cout)
            return y if kind == N.ACT_NONE else ops.act(y, kind, slope)
        y = ops.conv1d(ops.ntc_to_nct(x_cl, self.cin), self.w, self.bias, padding=self.dil * (self.ks - 1) // 2,
                       dilation=self.dil, act=kind, slope=slope)
        return ops.nct_to_ntc(y)


class _Plan:
    """Folded / packed weights of one module for one dtype; rebuilt when any parameter or buffer changes."""

    def __init__(self):
        self.sig, self.dtype, self.items = None, None, None

    def get(self, mod, dtype, build):
        sig = _sig(mod)
        if self.items is None or self.sig != sig or self.dtype != dtype:
            self.items, self.sig, self.dtype = build(dtype), sig, dtype
        return self.items


def _f32(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)


def _mean_t(x_cl):
    B, T, C = x_cl.shape
    out = _f32(B, C, device=x_cl.device)
    N.call("mv_mean_t_cl", ops._p(x_cl), ops._p(out), B, T, C, ops._dt(x_cl), ops._stream())
    return out


def _l2norm(x32, dtype):
    B, C = x32.shape
    y = torch.empty(B, C, device=x32.device, dtype=dtype)
    N.call("mv_l2norm_rows", ops._p(x32), ops._p(y), B, C, 1e-12, ops._DT[dtype], ops._stream())
    return y


def _mel_to_cl(x: torch.Tensor, cinp: int) -> torch.Tensor:
    if x.dim() != 3:
        raise RuntimeError(f"expected a mel-spectrogram [B, C, T], got {tuple(x.shape)}")
    ops._need_gpu(x)
    return ops.nct_to_ntc(x, cinp)


class SE_Module(nn.Module):
    """embedding_extractors.py:152-170."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(nn.Linear(channels, channels // reduction), nn.ReLU(),
                                nn.Linear(channels // reduction, channels), nn.Sigmoid())

    def gate(self, x_cl: torch.Tensor) -> torch.Tensor:
        B, T, C = x_cl.shape
        m = _mean_t(x_cl)
        g = _f32(B, C, device=x_cl.device)
        f0, f2 = self.fc[0], self.fc[2]
        w1, b1, w2, b2 = (t.detach().float().contiguous() for t in (f0.weight, f0.bias, f2.weight, f2.bias))   # no-ops for fp32 masters
        N.call("mv_se_gate", ops._p(m), ops._p(w1), ops._p(b1), ops._p(w2), ops._p(b2), ops._p(g), B, C, w1.shape[0], ops._stream())
        return g

    def scale_add_cl(self, x_cl, res_cl):
        B, T, C = x_cl.shape
        y = torch.empty_like(x_cl)
        N.call("mv_scale_add_cl", ops._p(x_cl), ops._p(self.gate(x_cl)), ops._p(res_cl), ops._p(y), B, T, C, ops._dt(x_cl), ops._stream())
        return y

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x_cl = _mel_to_cl(x, None)
        zero = torch.zeros_like(x_cl)
        return ops.ntc_to_nct(self.scale_add_cl(x_cl, zero))


class SE_Res2Block(nn.Module):
    """embedding_extractors.py:102-150."""

    def __init__(self, channels: int, dilation: int = 1, scale: int = 8):
        super().__init__()
        self.channels, self.scale, self.dilation = channels, scale, dilation
        self.conv1 = nn.Conv1d(channels, channels, kernel_size=1)
        self.bn1 = nn.BatchNorm1d(channels)
        self.scale_convs = nn.ModuleList([nn.Conv1d(channels // scale, channels // scale, kernel_size=3, padding=dilation,
                                                    dilation=dilation) for _ in range(scale)])
        self.conv2 = nn.Conv1d(channels, channels, kernel_size=1)
        self.bn2 = nn.BatchNorm1d(channels)
        self.se = SE_Module(channels)
        self._plan = _Plan()

    def _build(self, dtype):
        convs = list(self.scale_convs)[1:]
        plan = {"c1": _Affine(self.conv1.weight, self.conv1.bias, self.bn1, dtype),
                "sc": [None] + [_Affine(c.weight, c.bias, None, dtype, self.dilation) for c in convs],
                "c2": _Affine(self.conv2.weight, self.conv2.bias, self.bn2, dtype), "chain": None}
        cs = self.channels // self.scale
        if dtype != torch.float32 and self.scale == 8 and cs in (32, 64) and self.dilation <= 4:      # one launch for the whole chain
            plan["chain"] = (torch.cat([ops.dconv_pack(c.weight.detach().unsqueeze(2), dtype, 0) for c in convs]),
                             torch.stack([c.bias.detach() for c in convs]).to(dtype).contiguous())
        return plan

    def forward_cl(self, x_cl: torch.Tensor) -> torch.Tensor:
        p = self._plan.get(self, x_cl.dtype, self._build)
        B, T, C = x_cl.shape
        cs, rows, dt = C // self.scale, B * T, ops._dt(x_cl)
        if cs % 8:
            raise RuntimeError(f"SE_Res2Block: channels/scale = {cs} must be a multiple of 8")
        u = p["c1"](x_cl, _RELU)
        cat = torch.empty_like(u)
        if p["chain"] is not None and not _UNFUSED_RES2:
            N.call("mv_res2_chain", ops._p(u), ops._p(p["chain"][0]), ops._p(p["chain"][1]), ops._p(cat), B, T, C, cs, self.dilation, dt,
                   ops._stream())
            return self.se.scale_add_cl(p["c2"](cat, _RELU), x_cl)
        nxt = torch.empty(B, T, cs, device=u.device, dtype=u.dtype)
        # ys[0] = xs[0]; input of conv 1 = xs[1] + ys[0]  (:137-141)
        N.call("mv_res2_glue", ops._p(u), C, ops._p(u), ops._p(cat), ops._p(nxt), rows, C, cs, 0, cs, dt, ops._stream())
        for i in range(1, self.scale):
            y = p["sc"][i](nxt, _NONE)
            last = i == self.scale - 1
            N.call("mv_res2_glue", ops._p(y), cs, ops._p(u), ops._p(cat), None if last else ops._p(nxt), rows, C, cs, i * cs,
                   0 if last else (i + 1) * cs, dt, ops._stream())
        v = p["c2"](cat, _RELU)
        return self.se.scale_add_cl(v, x_cl)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.ntc_to_nct(self.forward_cl(_mel_to_cl(x, None)))


class ECAPA_TDNN(nn.Module):
    """embedding_extractors.py:13-100.  forward(mel [B, input_dim, T]) -> (L2-normalised embedding [B, embedding_dim], logits|None)."""

    def __init__(self, input_dim: int = 80, hidden_dim: int = 512, embedding_dim: int = 192, num_speakers: int = 1000):
        super().__init__()
        self.input_dim, self.hidden_dim, self.embedding_dim = input_dim, hidden_dim, embedding_dim
        self.input_conv = nn.Conv1d(input_dim, hidden_dim, kernel_size=5, dilation=1)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.se_res2_blocks = nn.ModuleList([SE_Res2Block(hidden_dim, dilation=2), SE_Res2Block(hidden_dim, dilation=3),
                                             SE_Res2Block(hidden_dim, dilation=4)])
        self.channel_expansion = nn.Conv1d(hidden_dim, 3 * hidden_dim, kernel_size=1)
        self.bn2 = nn.BatchNorm1d(3 * hidden_dim)
        self.attention = nn.Sequential(nn.Conv1d(3 * hidden_dim, hidden_dim, kernel_size=1), nn.Tanh(),
                                       nn.Conv1d(hidden_dim, 3 * hidden_dim, kernel_size=1), nn.Softmax(dim=1))
        self.final_proj = nn.Linear(6 * hidden_dim, embedding_dim)      # reference: 3*hidden (does not run), see module docstring
        self.bn3 = nn.BatchNorm1d(embedding_dim)
        self.speaker_classifier = nn.Linear(embedding_dim, num_speakers)
        self._plan = _Plan()

    def _build(self, dtype):
        fw, fb = _fold_bn(self.final_proj.weight, self.final_proj.bias, self.bn3)
        return {"in": _Affine(self.input_conv.weight, self.input_conv.bias, self.bn1, dtype, pad_in=True),
                "exp": _Affine(self.channel_expansion.weight, self.channel_expansion.bias, self.bn2, dtype),
                "a0": _Affine(self.attention[0].weight, self.attention[0].bias, None, dtype),
                "a2": _Affine(self.attention[2].weight, self.attention[2].bias, None, dtype),
                "final": (fw.contiguous(), fb.contiguous())}

    def pooled_cl(self, x: torch.Tensor) -> torch.Tensor:
        """mel [B, C, T] -> attentive statistics [B, 6*hidden] (fp32)."""
        if x.dim() != 3 or x.shape[1] != self.input_dim:
            raise RuntimeError(f"ECAPA_TDNN expects a mel-spectrogram [B, {self.input_dim}, T], got {tuple(x.shape)}")
        p = self._plan.get(self, x.dtype, self._build)
        B, _, T = x.shape
        if T < 6:
            raise RuntimeError(f"ECAPA_TDNN needs at least 6 frames (valid k=5 conv, unbiased std over time), got {T}")
        a = p["in"]
        h = a(_mel_to_cl(x, a.cinp if a.mfma else None), _RELU)                # 'same' conv; the reference's is valid (:25): crop
        hc = torch.empty(B, T - 4, h.shape[2], device=h.device, dtype=h.dtype)
        ops.copy_rows(h[:, 2:T - 2], hc)
        h = hc
        for blk in self.se_res2_blocks:
            h = blk.forward_cl(h)
        e = p["exp"](h, _RELU)
        lg = p["a2"](p["a0"](e, _TANH), _NONE)
        Tt, C3 = e.shape[1], e.shape[2]
        ws = torch.empty(N.lib().mv_asp_workspace_bytes(B, Tt), device=e.device, dtype=torch.uint8)
        pooled = _f32(B, 2 * C3, device=e.device)
        N.call("mv_asp_pool", ops._p(e), ops._p(lg), ops._p(ws), ops._p(pooled), B, Tt, C3, ops._dt(e), ops._stream())
        return pooled

    @torch.no_grad()
    def embed(self, x: torch.Tensor) -> torch.Tensor:
        pooled = self.pooled_cl(x)
        fw, fb = self._plan.items["final"]
        return _l2norm(ops.linear(pooled, fw, fb), x.dtype)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        emb = self.embed(x)
        if self.training:
            c = self.speaker_classifier
            return emb, ops.linear(emb, ops.cast(c.weight.detach(), emb.dtype), ops.cast(c.bias.detach(), emb.dtype))
        return emb, None


class Emotion2Vec(nn.Module):
    """embedding_extractors.py:172-257.  forward(mel) -> (frame embeddings [B,T,E], utterance embedding [B,E], logits|None).
    In `.train()` mode the reference feeds the [B, embedding_dim] utterance embedding to `Linear(hidden_dim, .)` (:250 vs :209): a shape
    error unless the two sizes agree.  Reproduced as a RuntimeError."""

    def __init__(self, input_dim: int = 80, hidden_dim: int = 512, embedding_dim: int = 256, num_emotions: int = 8):
        super().__init__()
        self.input_dim, self.hidden_dim, self.embedding_dim = input_dim, hidden_dim, embedding_dim
        self.feature_extractor = nn.Sequential(
            nn.Conv1d(input_dim, hidden_dim, kernel_size=7, padding=3), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
            nn.Conv1d(hidden_dim, hidden_dim, kernel_size=5, padding=2), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
            nn.Conv1d(hidden_dim, hidden_dim, kernel_size=3, padding=1), nn.BatchNorm1d(hidden_dim), nn.ReLU())
        layer = nn.TransformerEncoderLayer(d_model=hidden_dim, nhead=8, dim_feedforward=hidden_dim * 4, dropout=0.1,
                                           activation="relu", batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=6)       # parameter container only: never called
        self.emotion_classifier = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(), nn.Dropout(0.1),
                                                nn.Linear(hidden_dim // 2, num_emotions))
        self.frame_projection = nn.Linear(hidden_dim, embedding_dim)
        self.utterance_projection = nn.Linear(hidden_dim, embedding_dim)
        self.nhead = 8
        self._plan = _Plan()

    def _build(self, dtype):
        fe = self.feature_extractor
        layers = []
        for l in self.transformer.layers:
            if l.norm_first:
                raise RuntimeError("Emotion2Vec: pre-norm encoder layers are not what the reference builds")
            layers.append({"qkv": _Affine(l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, None, dtype),
                           "out": _Affine(l.self_attn.out_proj.weight, l.self_attn.out_proj.bias, None, dtype),
                           "ff1": _Affine(l.linear1.weight, l.linear1.bias, None, dtype),
                           "ff2": _Affine(l.linear2.weight, l.linear2.bias, None, dtype),
                           "n1": (l.norm1.weight.detach().float().contiguous(), l.norm1.bias.detach().float().contiguous(), l.norm1.eps),
                           "n2": (l.norm2.weight.detach().float().contiguous(), l.norm2.bias.detach().float().contiguous(), l.norm2.eps)})
        return {"convs": [_Affine(fe[0].weight, fe[0].bias, fe[1], dtype, pad_in=True), _Affine(fe[3].weight, fe[3].bias, fe[4], dtype),
                          _Affine(fe[6].weight, fe[6].bias, fe[7], dtype)],
                "layers": layers,
                "frame": _Affine(self.frame_projection.weight, self.frame_projection.bias, None, dtype)}

    @staticmethod
    def _add_ln(x, res, n):
        B, T, C = x.shape
        y = torch.empty_like(x)
        N.call("mv_add_layernorm", ops._p(x), ops._p(res), ops._p(n[0]), ops._p(n[1]), ops._p(y), B * T, C, float(n[2]), ops._dt(x),
               ops._stream())
        return y

    def encode_cl(self, x: torch.Tensor) -> torch.Tensor:
        """mel [B, C, T] -> transformer output [B, T, hidden] (channels-last)."""
        if x.dim() != 3 or x.shape[1] != self.input_dim:
            raise RuntimeError(f"Emotion2Vec expects a mel-spectrogram [B, {self.input_dim}, T], got {tuple(x.shape)}")
        p = self._plan.get(self, x.dtype, self._build)
        c0 = p["convs"][0]
        f = _mel_to_cl(x, c0.cinp if c0.mfma else None)
        for c in p["convs"]:
            f = c(f, _RELU)
        B, T, H = f.shape
        hd = H // self.nhead
        for L in p["layers"]:
            qkv = L["qkv"](f, _NONE)
            att = torch.empty_like(f)
            N.call("mv_mha_fwd", ops._p(qkv), ops._p(att), B, T, self.nhead, hd, ops._dt(f), ops._stream())
            f = self._add_ln(L["out"](att, _NONE), f, L["n1"])
            f = self._add_ln(L["ff2"](L["ff1"](f, _RELU), _NONE), f, L["n2"])
        return f

    def _utterance(self, f, dtype):
        up = self.utterance_projection
        return _l2norm(ops.linear(_mean_t(f), up.weight.detach().float(), up.bias.detach().float()), dtype)

    @torch.no_grad()
    def embed(self, x: torch.Tensor) -> torch.Tensor:
        """Utterance-level embedding only (what EmbeddingExtractor consumes): no frame projection, no classifier head."""
        return self._utterance(self.encode_cl(x), x.dtype)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        f = self.encode_cl(x)
        frame = self._plan.items["frame"](f, _NONE)
        utt = self._utterance(f, x.dtype)
        if self.training:
            c0, c3 = self.emotion_classifier[0], self.emotion_classifier[3]
            cast = lambda t: ops.cast(t.detach(), utt.dtype)
            h = ops.act(ops.linear(utt, cast(c0.weight), cast(c0.bias)), N.ACT_LRELU, 0.0)
            return frame, utt, ops.linear(h, cast(c3.weight), cast(c3.bias))
        return frame, utt, None


_TWO_STREAMS = os.environ.get("MV_EMBED_STREAMS", "2") != "1"
_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=key)
    return st


class EmbeddingExtractor(nn.Module):
    """embedding_extractors.py:259-284: mel [B, 80, T] -> (speaker embedding [B, 192], emotion embedding [B, 256])."""

    def __init__(self, speaker_embedding_dim: int = 192, emotion_embedding_dim: int = 256):
        super().__init__()
        self.speaker_extractor = ECAPA_TDNN(embedding_dim=speaker_embedding_dim)
        self.emotion_extractor = Emotion2Vec(embedding_dim=emotion_embedding_dim)

    @torch.no_grad()
    def forward(self, mel_spectrogram: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        # the reference discards the classifier logits and the frame embeddings here (:277-282); they are not computed at all, which
        # also keeps `.train()` mode usable (Emotion2Vec.forward raises there, as the reference's does: see the class docstring)
        if not (mel_spectrogram.is_cuda and _TWO_STREAMS):
            return self.speaker_extractor.embed(mel_spectrogram), self.emotion_extractor.embed(mel_spectrogram)
        # The two encoders are independent chains of ~40 short, latency-bound launches each: the emotion encoder runs on a second HIP
        # stream (fork / join by events - also what a graph capture records, so the replayed graph has two parallel branches) and the
        # GPU overlaps them (0.53 -> ~0.3 ms per B = 32 x 32-frame batch).  MV_EMBED_STREAMS=1 keeps everything on one stream.
        main = torch.cuda.current_stream(mel_spectrogram.device)
        side = _side_stream(mel_spectrogram.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            emo = self.emotion_extractor.embed(mel_spectrogram)
        spk = self.speaker_extractor.embed(mel_spectrogram)
        main.wait_stream(side)
        if not torch.cuda.is_current_stream_capturing():
            emo.record_stream(main)            # allocated on the side stream, consumed on the caller's
        return spk, emo
