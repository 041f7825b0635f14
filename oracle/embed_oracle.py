"""CPU oracle for the conditioning producers: ECAPA-TDNN speaker and Emotion2Vec emotion encoders.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may.  Plain functional math on CPU tensors (eval mode: BatchNorm uses its running
statistics, dropout is the identity), parameters as a flat ``dict[str, Tensor]`` keyed like the reference's ``state_dict``.
Pinned against golden vectors generated from the reference's own classes (``tests/golden/make_goldens_embed.py``; checked in
``tests/test_oracle_vs_golden.py``).

Reference lines restated (paths relative to the reference root):
  se_module            embedding_extractors.py:152-170
  se_res2_block        embedding_extractors.py:102-150
  ecapa_tdnn           embedding_extractors.py:13-100   (final_proj takes cat(mean, std) = 6*hidden features: the reference's
                                                        Linear(3*hidden, .) at :48 cannot consume :84-87; fixture made with that
                                                        one layer resized, see make_goldens_embed.py)
  emotion2vec          embedding_extractors.py:172-257  (nn.TransformerEncoderLayer defaults: post-norm, ReLU, eps 1e-5,
                                                        nhead 8, batch_first; restated here as explicit matmuls)
  embedding_extractor  embedding_extractors.py:259-284
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def batch_norm_eval(x: Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> Tensor:
    shape = (1, -1) + (1,) * (x.dim() - 2)
    inv = torch.rsqrt(sd[prefix + "running_var"].to(x.dtype) + eps) * sd[prefix + "weight"].to(x.dtype)
    return (x - sd[prefix + "running_mean"].to(x.dtype).view(shape)) * inv.view(shape) + sd[prefix + "bias"].to(x.dtype).view(shape)


def _conv(x: Tensor, sd: SD, prefix: str, **kw) -> Tensor:
    return F.conv1d(x, sd[prefix + "weight"].to(x.dtype), sd[prefix + "bias"].to(x.dtype), **kw)


def _linear(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return x @ sd[prefix + "weight"].to(x.dtype).t() + sd[prefix + "bias"].to(x.dtype)


def se_module(x: Tensor, sd: SD, prefix: str) -> Tensor:
    y = x.mean(dim=2)
    y = torch.sigmoid(_linear(torch.relu(_linear(y, sd, prefix + "fc.0.")), sd, prefix + "fc.2."))
    return x * y.unsqueeze(-1)


def se_res2_block(x: Tensor, sd: SD, prefix: str, dilation: int, scale: int = 8) -> Tensor:
    h = torch.relu(batch_norm_eval(_conv(x, sd, prefix + "conv1."), sd, prefix + "bn1."))
    xs = torch.chunk(h, scale, dim=1)
    ys = [xs[0]]
    for i in range(1, scale):
        ys.append(_conv(xs[i] + ys[-1], sd, prefix + f"scale_convs.{i}.", padding=dilation, dilation=dilation))
    h = torch.cat(ys, dim=1)
    h = torch.relu(batch_norm_eval(_conv(h, sd, prefix + "conv2."), sd, prefix + "bn2."))
    return se_module(h, sd, prefix + "se.") + x


def ecapa_tdnn(mel: Tensor, sd: SD, prefix: str = "", dilations=(2, 3, 4), want_taps: bool = False):
    taps = {}
    x = torch.relu(batch_norm_eval(_conv(mel, sd, prefix + "input_conv."), sd, prefix + "bn1."))
    for i, d in enumerate(dilations):
        x = se_res2_block(x, sd, prefix + f"se_res2_blocks.{i}.", d)
        taps[f"block{i}"] = x
    x = torch.relu(batch_norm_eval(_conv(x, sd, prefix + "channel_expansion."), sd, prefix + "bn2."))
    a = torch.tanh(_conv(x, sd, prefix + "attention.0."))
    a = torch.softmax(_conv(a, sd, prefix + "attention.2."), dim=1)          # over channels, as the reference has it (:44)
    taps["attention"] = a
    att = x * a
    pooled = torch.cat([att.mean(dim=2), att.std(dim=2)], dim=1)               # torch.std: unbiased
    taps["pooled"] = pooled
    e = batch_norm_eval(_linear(pooled, sd, prefix + "final_proj."), sd, prefix + "bn3.")
    e = F.normalize(e, p=2, dim=1)
    return (e, taps) if want_taps else e


def layer_norm(x: Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * sd[prefix + "weight"].to(x.dtype) + sd[prefix + "bias"].to(x.dtype)


def transformer_encoder_layer(x: Tensor, sd: SD, prefix: str, nhead: int = 8) -> Tensor:
    """x [B,T,H]; post-norm layer: x = LN1(x + MHA(x)); x = LN2(x + W2 relu(W1 x))."""
    B, T, H = x.shape
    hd = H // nhead
    qkv = x @ sd[prefix + "self_attn.in_proj_weight"].to(x.dtype).t() + sd[prefix + "self_attn.in_proj_bias"].to(x.dtype)
    q, k, v = (t.view(B, T, nhead, hd).transpose(1, 2) for t in qkv.split(H, dim=-1))
    p = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1)
    a = (p @ v).transpose(1, 2).reshape(B, T, H)
    x = layer_norm(x + _linear(a, sd, prefix + "self_attn.out_proj."), sd, prefix + "norm1.")
    ff = _linear(torch.relu(_linear(x, sd, prefix + "linear1.")), sd, prefix + "linear2.")
    return layer_norm(x + ff, sd, prefix + "norm2.")


def emotion2vec(mel: Tensor, sd: SD, prefix: str = "", num_layers: int = 6, want_taps: bool = False):
    taps = {}
    x = mel
    for ci, bi, pad in ((0, 1, 3), (3, 4, 2), (6, 7, 1)):
        x = torch.relu(batch_norm_eval(_conv(x, sd, prefix + f"feature_extractor.{ci}.", padding=pad), sd,
                                       prefix + f"feature_extractor.{bi}."))
    taps["features"] = x
    x = x.transpose(1, 2)
    for l in range(num_layers):
        x = transformer_encoder_layer(x, sd, prefix + f"transformer.layers.{l}.")
        if l == 0:
            taps["layer0"] = x
    frame = _linear(x, sd, prefix + "frame_projection.")
    utt = F.normalize(_linear(x.mean(dim=1), sd, prefix + "utterance_projection."), p=2, dim=1)
    return (frame, utt, taps) if want_taps else (frame, utt)


def embedding_extractor(mel: Tensor, sd: SD, prefix: str = "") -> Tuple[Tensor, Tensor]:
    spk = ecapa_tdnn(mel, sd, prefix + "speaker_extractor.")
    _, emo = emotion2vec(mel, sd, prefix + "emotion_extractor.")
    return spk, emo
