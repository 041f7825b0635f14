"""Generators of the modified HiFi-GAN on MI355X.

``ModifiedHiFiGANGenerator`` / ``HiFiGANGenerator`` implement the generator that the reference's
``conditioned_hifigan.py:4,57-67`` is written against (its source was deleted upstream; the
specification is SURVEY.md Appendix A): ODConv1d input projection -> FiLM -> 4x (ODConvTranspose1d +
LeakyReLU 0.1) -> 3x MultiReceptiveFieldBlock -> Conv1d(k=11) -> tanh.  ``GroupedResidualConv1D`` and
``FeatureWiseLinearModulation`` are the two working blocks of the current ``generator.py``
(:109-172, :174-199).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from .discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator
from .grc_lora import FiLMLayer, MultiReceptiveFieldBlock
from .odconv import ODConv1d, ODConvTranspose1d


class GroupedResidualConv1D(nn.Module):
    def __init__(self, channels: int, kernel_size: int = 3, dilation: int = 1, groups: int = 4, lora_rank: int = 8):
        super().__init__()
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, dilation
        self.groups, self.lora_rank = groups, lora_rank
        self.grouped_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) * dilation // 2,
                                      dilation=dilation, groups=groups)
        self.lora_A = nn.Parameter(torch.randn(lora_rank, channels // groups))
        self.lora_B = nn.Parameter(torch.randn(channels // groups, lora_rank))
        self.lora_alpha = nn.Parameter(torch.ones(1))
        self.channel_mixer = nn.Conv1d(channels, channels, 1)
        self.activation = nn.LeakyReLU(0.1)
        self.norm = nn.GroupNorm(groups, channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.grouped_residual_conv1d(x, self)


class FeatureWiseLinearModulation(nn.Module):
    def __init__(self, embedding_dim: int, feature_dim: int):
        super().__init__()
        self.embedding_dim, self.feature_dim = embedding_dim, feature_dim
        self.scale_proj = nn.Linear(embedding_dim, feature_dim)
        self.shift_proj = nn.Linear(embedding_dim, feature_dim)

    def forward(self, x, speaker_embedding, emotion_embedding):
        return Fn.film2(x, speaker_embedding, emotion_embedding, self.scale_proj.weight, self.scale_proj.bias,
                        self.shift_proj.weight, self.shift_proj.bias)


class ModifiedHiFiGANGenerator(nn.Module):
    """mel [B, mel_channels, T] -> waveform [B, 1, T * prod(upsample_factors)]."""

    def __init__(self, mel_channels=80, hidden_channels=512, kernel_size=7, upsample_factors=[8, 8, 2, 2],
                 resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 groups=4, lora_rank=16, dropout=0.1, input_channels=None, speaker_embedding_dim=192,
                 emotion_embedding_dim=256):
        super().__init__()
        if input_channels is not None:  # keyword of the current-source class (generator.py:204)
            mel_channels = input_channels
        self.mel_channels, self.hidden_channels, self.kernel_size = mel_channels, hidden_channels, kernel_size
        self.upsample_factors = list(upsample_factors)
        self.resblock_kernel_sizes = list(resblock_kernel_sizes)
        self.resblock_dilation_sizes = [list(d) for d in resblock_dilation_sizes]
        self.groups, self.lora_rank, self.dropout = groups, lora_rank, dropout
        self.speaker_embedding_dim, self.emotion_embedding_dim = speaker_embedding_dim, emotion_embedding_dim

        self.input_proj = ODConv1d(mel_channels, hidden_channels, kernel_size, padding=kernel_size // 2,
                                   K=4, reduction_factor=4)
        self.upsample_layers = nn.ModuleList()
        cur, n_up = hidden_channels, len(self.upsample_factors)
        for i, f in enumerate(self.upsample_factors):
            out = max(cur // 2, groups * 2) if i < n_up - 1 else max(cur, groups * 2)
            out = max(out // groups * groups, 64)
            self.upsample_layers.append(nn.Sequential(
                ODConvTranspose1d(cur, out, kernel_size=f * 2, stride=f, padding=f // 2, output_padding=f % 2,
                                  K=4, reduction_factor=4),
                nn.LeakyReLU(0.1)))
            cur = out
        self.mrf_blocks = nn.ModuleList()
        last_resblock_kernel = kernel_size
        for last_resblock_kernel, dilations in zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes):
            ch = max(cur, groups * len(dilations) * 2) // groups * groups
            self.mrf_blocks.append(MultiReceptiveFieldBlock(cur, ch, dilations=list(dilations),
                                                            groups=min(groups, ch // 4), r=lora_rank,
                                                            dropout=dropout))
            cur = ch
        # the original constructor re-used its MRF loop variable here, so the output conv has the LAST
        # resblock kernel size (11 by default), not `kernel_size`
        self.output_proj = nn.Conv1d(cur, 1, last_resblock_kernel, padding=last_resblock_kernel // 2)
        self.final_film = FiLMLayer(cur, cur)
        self._initialize_weights()

    def _initialize_weights(self):
        nn.init.kaiming_normal_(self.output_proj.weight, mode="fan_out", nonlinearity="leaky_relu")
        nn.init.zeros_(self.output_proj.bias)
        for layer in self.upsample_layers:
            layer[0]._initialize_weights()

    def unused_parameters(self):
        for m in [self.input_proj] + [l[0] for l in self.upsample_layers]:
            yield from m.unused_parameters()

    def set_mixed_precision(self, through="up1", dtype=torch.float16, mrf_weights=None):
        """Inference-only storage mix for an fp32-storage model (not in the reference): the stages up to and including `through`
        ("input_proj", "up0", "up1", ...; None switches the mix off) run in `dtype` storage, everything behind - the last upsamplers,
        the three MultiReceptiveFieldBlocks, the output conv - in fp32 storage with split bf16 MFMA operands.  The early stages feed
        a 1e3x gain chain, but their rounding enters once; tools/error_budget.py puts fp16 through up1 at 5.9e-4 waveform rel-L2 on the
        22 kHz generator (north_star: 1e-3) and above 1e-3 on the 48 kHz one - DESIGN.md section 5.
        mrf_weights="fp16": the three MultiReceptiveFieldBlocks keep fp32 storage and hi + lo (f16) activation operands but use their
        weights as single f16 values - two MFMA products per MAC instead of three (MV_F32_W16; +1.1e-4 in quadrature at 22 kHz,
        out of tolerance at 48 kHz).  Returns self."""
        if mrf_weights not in (None, "fp16"):
            raise ValueError("mrf_weights must be None or 'fp16'")
        object.__setattr__(self, "_mv_mrf_w16", mrf_weights == "fp16" and through is not None)
        if through is None:
            mixed = None
        else:
            names = ["input_proj"] + [f"up{i}" for i in range(len(self.upsample_layers))]
            if through not in names or dtype not in (torch.float16, torch.bfloat16):
                raise ValueError(f"through must be one of {names} (or None), dtype fp16 / bf16")
            mixed = (names.index(through), dtype)
            if (self.mel_channels, self.upsample_factors) != (80, [8, 8, 2, 2]):
                import warnings
                warnings.warn("set_mixed_precision: the error budget behind this mix (DESIGN.md section 5) was measured on the 80-mel "
                              f"[8, 8, 2, 2] generator only; on {self.mel_channels}-mel {self.upsample_factors} (the 48 kHz geometry: every "
                              "sub-fp32 mix simulated above 1e-3 waveform rel-L2) check the output against fp32 storage before using it",
                              stacklevel=2)
        object.__setattr__(self, "_mv_mixed", mixed)
        return self

    @property
    def mixed_precision(self):
        m = getattr(self, "_mv_mixed", None)
        return None if m is None else ((["input_proj"] + [f"up{i}" for i in range(len(self.upsample_layers))])[m[0]], m[1])

    def forward(self, mel, speaker_emb=None, emotion_emb=None, return_stages=False, force_generic=False):
        if not force_generic and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
                                      ) and not (self.training and self.dropout > 0):
            from .fused import generator_fused_for
            fz = generator_fused_for(self)
            if fz is not None:
                # inference fast path: channels-last MFMA pipeline (csrc/odconv_fused.hip, mrf_fused.hip, conv_out.hip)
                if return_stages or Fn.DISPATCH != "torch_ops":
                    return fz.forward(mel, speaker_emb, emotion_emb, cache=Fn._cache, return_stages=return_stages)
                t = Fn._t()      # one operator for the whole captured pipeline: torch.ops.mi355x_vocoder.generator_forward
                return t.OPS.generator_forward(mel, speaker_emb, emotion_emb, t.object_handle(fz))
        st = {}
        x = self.input_proj(mel)
        st["input_proj"] = x
        if speaker_emb is not None or emotion_emb is not None:
            x = self.final_film(x, speaker_emb, emotion_emb)
            st["film"] = x
        for i, layer in enumerate(self.upsample_layers):
            x = layer[0](x, act="lrelu", slope=layer[1].negative_slope)   # LeakyReLU fused into the ODConvT launch
            st[f"up{i}"] = x
        for i, blk in enumerate(self.mrf_blocks):
            x = blk(x, speaker_emb, emotion_emb)
            st[f"mrf{i}"] = x
        k = self.output_proj.kernel_size[0]
        x = Fn.conv1d(x, self.output_proj.weight, self.output_proj.bias, padding=k // 2, act="tanh")
        st["wave"] = x
        return st if return_stages else x


class HiFiGANGenerator(nn.Module):
    """Generator + MPD + MSD container ('HiFiGANGenerator')."""

    def __init__(self, mel_channels=80, hidden_channels=512, kernel_size=7, upsample_factors=[8, 8, 2, 2],
                 resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                 groups=4, lora_rank=16, dropout=0.1):
        super().__init__()
        self.generator = ModifiedHiFiGANGenerator(
            mel_channels=mel_channels, hidden_channels=hidden_channels, kernel_size=kernel_size,
            upsample_factors=upsample_factors, resblock_kernel_sizes=resblock_kernel_sizes,
            resblock_dilation_sizes=resblock_dilation_sizes, groups=groups, lora_rank=lora_rank, dropout=dropout)
        self.mpd = MultiPeriodDiscriminator()
        self.msd = MultiScaleDiscriminator()

    def forward(self, mel, speaker_emb=None, emotion_emb=None):
        return self.generator(mel, speaker_emb, emotion_emb)

    def get_discriminator_outputs(self, real_audio, fake_audio):
        return {
            "mpd_real": self.mpd(real_audio),
            "mpd_fake": self.mpd(fake_audio),
            "msd_real": self.msd(real_audio),
            "msd_fake": self.msd(fake_audio),
        }
