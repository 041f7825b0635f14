"""GRC+LoRA residual block, FiLM and the multi-receptive-field block on MI355X.

Drop-in for the reference's ``hifigan_modified/grc_lora.py`` (GRC_LoRA_Block :5-68, FiLMLayer :70-129,
MultiReceptiveFieldBlock :131-163).  Differences that are deliberate and documented in DESIGN.md:
``residual_proj`` is created in the constructor (the reference creates it lazily inside forward,
grc_lora.py:62-66, on the CPU and outside every optimizer) - it is drawn from a forked RNG so the
main random stream, and therefore every other parameter, matches the reference for a given seed.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn


def _norm_groups(channels):
    return min(8, channels // 4) if channels >= 4 else 1


class GRC_LoRA_Block(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, dilation, r=4):
        super().__init__()
        groups = min(in_channels, out_channels, 4)
        in_channels, out_channels = max(in_channels, groups), max(out_channels, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.dilation, self.groups = kernel_size, dilation, groups
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding=(kernel_size - 1) * dilation // 2,
                              dilation=dilation, groups=groups)
        self.lora_A = nn.Parameter(torch.randn(in_channels, r))
        self.lora_B = nn.Parameter(torch.randn(r, out_channels))
        self.lora_scaling = nn.Parameter(torch.ones(1))
        self.output_projection = nn.Conv1d(out_channels, out_channels, 1)
        self.norm_groups = _norm_groups(out_channels)
        self.norm = nn.GroupNorm(self.norm_groups, out_channels)
        self.activation = nn.SiLU()
        if in_channels != out_channels:
            with torch.random.fork_rng(devices=[]):
                self.residual_proj = nn.Conv1d(in_channels, out_channels, 1)
            self.register_load_state_dict_post_hook(self._tolerate_missing_residual_proj)

    @staticmethod
    def _tolerate_missing_residual_proj(module, incompatible_keys):
        # a reference checkpoint written before its first forward has no residual_proj.* keys
        incompatible_keys.missing_keys[:] = [k for k in incompatible_keys.missing_keys
                                             if ".residual_proj." not in "." + k]

    def forward(self, x, out=None, out_channel_offset=0):
        return Fn.grc_lora_block(x, self, out=out, out_channel_offset=out_channel_offset)


class FiLMLayer(nn.Module):
    def __init__(self, feature_dim, condition_dim):
        super().__init__()
        self.feature_dim, self.condition_dim = feature_dim, condition_dim
        self.condition_projection = nn.Linear(condition_dim, feature_dim * 2)

    def condition(self, speaker_emb=None, emotion_emb=None):
        """cat -> zero-pad / truncate to condition_dim (grc_lora.py:82-105).  Host-side glue on [B, Cd]."""
        if speaker_emb is not None and emotion_emb is not None:
            cond = torch.cat([speaker_emb, emotion_emb], dim=1)
        elif speaker_emb is not None:
            cond = speaker_emb
        elif emotion_emb is not None:
            cond = emotion_emb
        else:
            return None
        want = self.condition_projection.in_features
        if cond.size(1) < want:
            cond = torch.cat([cond, cond.new_zeros(cond.size(0), want - cond.size(1))], dim=1)
        elif cond.size(1) > want:
            cond = cond[:, :want]
        return cond

    def forward(self, features, speaker_emb=None, emotion_emb=None):
        cond = self.condition(speaker_emb, emotion_emb)
        if cond is None:
            return features
        return Fn.film(features, cond, self.condition_projection.weight, self.condition_projection.bias,
                       self.feature_dim)


class MultiReceptiveFieldBlock(nn.Module):
    def __init__(self, in_channels, out_channels, dilations=[1, 3, 5], groups=4, r=16, dropout=0.1):
        super().__init__()
        cpd = out_channels // len(dilations)
        cpd = (cpd // groups) * groups
        if cpd < groups:
            cpd = groups
        self.in_channels, self.out_channels = in_channels, out_channels
        self.dilations, self.channels_per_dilation = list(dilations), cpd
        self.conv_layers = nn.ModuleList([GRC_LoRA_Block(in_channels, cpd, 3, d, r) for d in dilations])
        self.fusion = nn.Conv1d(cpd * len(dilations), out_channels, 1)
        self.dropout = nn.Dropout(dropout)
        self.norm_groups = _norm_groups(out_channels)
        self.norm = nn.GroupNorm(self.norm_groups, out_channels)

    def forward(self, x, speaker_emb=None, emotion_emb=None):
        # the embeddings are accepted and ignored, exactly like grc_lora.py:157
        return Fn.mrf_block(x, self)
