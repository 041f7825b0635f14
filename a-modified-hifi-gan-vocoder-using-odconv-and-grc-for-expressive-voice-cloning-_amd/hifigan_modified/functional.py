"""Differentiable building blocks of the vocoder path, expressed over the HIP ops in ``ops.py``.

Each public function corresponds to one reference module's forward (cited below).  Parameters are
taken as they live in the nn.Module (fp32 masters or already low precision) and are cast to the
activation dtype through a version-keyed cache.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import ops

_cache = ops._ParamCache()
_ACT = {None: N.ACT_NONE, "none": N.ACT_NONE, "lrelu": N.ACT_LRELU, "tanh": N.ACT_TANH, "silu": N.ACT_SILU}


def _w(p, like):
    return _cache.get(p, like.dtype)


class _Pending(torch.autograd.Function):
    """Marks an output as differentiable-in-principle: the forward is the HIP path; asking for a
    gradient through an op whose backward kernels are not built yet fails loudly instead of silently
    returning zeros."""

    @staticmethod
    def forward(ctx, name, y, *deps):
        ctx.name = name
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(f"mi355x vocoder: backward of `{ctx.name}` is not available in this build")


def _track(name, y, *deps):
    if torch.is_grad_enabled() and any(d is not None and d.requires_grad for d in deps):
        return _Pending.apply(name, y, *deps)
    return y


# ----------------------------------------------------------------------------------------------- ODConv
def odconv_attention(x, att_w, att_b):
    """odconv.py:36-40,85."""
    K, C = att_w.shape[0], att_w.shape[1]
    with torch.no_grad():
        alpha = ops.odconv_attn(x, _w(att_w, x).view(K, C), _w(att_b, x))
    return _track("odconv_attention", alpha, x, att_w, att_b)


def odconv1d(x, kernels, bias, att_w, att_b, stride=1, padding=0, dilation=1, act=None, slope=0.1):
    """odconv.py:73-108: attention, then ONE launch doing aggregate + conv (+ act)."""
    with torch.no_grad():
        K, C = att_w.shape[0], att_w.shape[1]
        alpha = ops.odconv_attn(x, _w(att_w, x).view(K, C), _w(att_b, x))
        y = ops.conv1d(x, _w(kernels, x), _w(bias, x), alpha, stride, padding, dilation, 1, _ACT[act], slope)
    return _track("odconv1d", y, x, kernels, bias, att_w, att_b)


def odconv_transpose1d(x, kernels, bias, att_w, att_b, stride=1, padding=0, output_padding=0, dilation=1,
                       act=None, slope=0.1):
    """odconv.py:172-205."""
    with torch.no_grad():
        K, C = att_w.shape[0], att_w.shape[1]
        alpha = ops.odconv_attn(x, _w(att_w, x).view(K, C), _w(att_b, x))
        y = ops.conv_transpose1d(x, _w(kernels, x), _w(bias, x), alpha, stride, padding, output_padding, dilation,
                                 _ACT[act], slope)
    return _track("odconv_transpose1d", y, x, kernels, bias, att_w, att_b)


# ----------------------------------------------------------------------------------------------- plain layers
def conv1d(x, weight, bias, stride=1, padding=0, dilation=1, groups=1, act=None, slope=0.1):
    with torch.no_grad():
        y = ops.conv1d(x, _w(weight, x), _w(bias, x), None, stride, padding, dilation, groups, _ACT[act], slope)
    return _track("conv1d", y, x, weight, bias)


def film(x, cond, proj_w, proj_b, feature_dim):
    """grc_lora.py:108-129 (the condition is already cat/pad/truncated by FiLMLayer.condition)."""
    with torch.no_grad():
        proj = ops.linear(ops.cast(cond, x.dtype), _w(proj_w, x), _w(proj_b, x))
        y = ops.film(x, proj, feature_dim)
    return _track("film", y, x, cond, proj_w, proj_b)


def film2(x, spk, emo, scale_w, scale_b, shift_w, shift_b):
    """generator.py:187-199: (W_s e + b_s) * x + (W_h e + b_h), e = spk + emo."""
    with torch.no_grad():
        e = ops.act(ops.cast(spk, x.dtype), N.ACT_NONE, res=ops.cast(emo, x.dtype))
        scale = ops.linear(e, _w(scale_w, x), _w(scale_b, x))
        shift = ops.linear(e, _w(shift_w, x), _w(shift_b, x))
        y = ops.scale_shift(x, scale, shift)
    return _track("film2", y, x, spk, emo, scale_w, scale_b, shift_w, shift_b)


# ----------------------------------------------------------------------------------------------- GRC + LoRA / MRF (generic shapes)
def _grc_forward_into(x, blk, out, off):
    """grc_lora.py:32-68 with the parameter algebra folded: conv_g + LoRA + 1x1 -> one dense dilated conv."""
    k, d = blk.kernel_size, blk.dilation
    if k % 2 == 0:
        raise RuntimeError("GRC_LoRA_Block: even kernel sizes make base/LoRA lengths differ (as in the reference)")
    w_eff, b_eff = ops.grc_fold_weights(_w(blk.conv.weight, x), _w(blk.conv.bias, x), _w(blk.lora_A, x),
                                        _w(blk.lora_B, x), _w(blk.lora_scaling, x),
                                        _w(blk.output_projection.weight, x), _w(blk.output_projection.bias, x),
                                        blk.groups)
    v = ops.conv1d(x, w_eff, b_eff, None, 1, (k - 1) * d // 2, d, 1)
    mean, rstd = ops.groupnorm_stats(v, blk.norm_groups, blk.norm.eps)
    C = blk.out_channels
    ysl = out[:, off:off + C]
    if blk.in_channels != C:
        ops.conv1d(x, _w(blk.residual_proj.weight, x), _w(blk.residual_proj.bias, x), out=out, out_channel_offset=off)
        res = ysl
    else:
        res = x
    ops.groupnorm_apply(v, mean, rstd, _w(blk.norm.weight, x), _w(blk.norm.bias, x), blk.norm_groups,
                        act=N.ACT_SILU, res=res, out=ysl)


def grc_lora_block(x, blk, out=None, out_channel_offset=0):
    with torch.no_grad():
        x = x if x.stride(2) == 1 else x.contiguous()
        if out is None:
            out = torch.empty(x.shape[0], blk.out_channels, x.shape[2], device=x.device, dtype=x.dtype)
            out_channel_offset = 0
        _grc_forward_into(x, blk, out, out_channel_offset)
    return _track("grc_lora_block", out, x, *blk.parameters())


def mrf_block(x, blk, force_generic=False):
    """grc_lora.py:157-163.  64-channel blocks of the generator's shape run the fused MFMA kernel
    (csrc/mrf_fused.hip) in channels-last layout; any other shape runs the generic kernels, where the
    three branches write straight into the channel slices of the concat buffer (no torch.cat copy)."""
    from .fused import mrf_fused_for
    fz = None if force_generic else mrf_fused_for(blk)
    if fz is not None and not (blk.training and blk.dropout.p > 0):
        with torch.no_grad():
            y = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x)))
        return _track("mrf_block", y, x, *blk.parameters())
    with torch.no_grad():
        x = x if x.is_contiguous() else x.contiguous()
        B, C, T = x.shape
        cpd, n = blk.channels_per_dilation, len(blk.dilations)
        cat = torch.empty(B, cpd * n, T, device=x.device, dtype=x.dtype)
        for i, g in enumerate(blk.conv_layers):
            _grc_forward_into(x, g, cat, i * cpd)
        f = ops.conv1d(cat, _w(blk.fusion.weight, x), _w(blk.fusion.bias, x))
        mean, rstd = ops.groupnorm_stats(f, blk.norm_groups, blk.norm.eps)
        mask, scale = None, 1.0
        p = blk.dropout.p
        if blk.training and p > 0:
            mask = (torch.rand(f.shape, device=x.device) >= p).to(torch.uint8)  # TODO(philox kernel)
            scale = 1.0 / (1.0 - p)
        y = ops.groupnorm_apply(f, mean, rstd, _w(blk.norm.weight, x), _w(blk.norm.bias, x), blk.norm_groups,
                                res=x, mask=mask, mask_scale=scale)
    return _track("mrf_block", y, x, *blk.parameters())


def grouped_residual_conv1d(x, blk):
    """generator.py:141-172: LeakyReLU(GN_G(Conv1x1(conv_g(x) + alpha*LoRA_g(x)) + x))."""
    with torch.no_grad():
        x = x if x.is_contiguous() else x.contiguous()
        G, C = blk.groups, blk.channels
        # LoRA_g is the same [C/G x C/G] map M = B A on every group: a grouped 1x1 conv with weight M per group
        M = ops.linear(_w(blk.lora_B, x), _w(blk.lora_A, x).t().contiguous())         # [C/G, C/G] = B @ A
        wl = M.repeat(G, 1).unsqueeze(-1).contiguous()                                  # [C, C/G, 1]
        k, d = blk.kernel_size, blk.dilation
        h = ops.conv1d(x, _w(blk.grouped_conv.weight, x), _w(blk.grouped_conv.bias, x), None, 1, (k - 1) * d // 2, d, G)
        # u = h + alpha * lora: scale the (tiny) LoRA weight by alpha, accumulate onto h through `res`
        wl = ops.scale_shift(wl.view(1, 1, -1), ops.cast(blk.lora_alpha.detach(), x.dtype).view(1, 1),
                             torch.zeros(1, 1, device=x.device, dtype=x.dtype)).view(C, C // G, 1)
        u = ops.conv1d(x, wl, None, None, 1, 0, 1, G, res=h)
        m = ops.conv1d(u, _w(blk.channel_mixer.weight, x), _w(blk.channel_mixer.bias, x), res=x)
        mean, rstd = ops.groupnorm_stats(m, G, blk.norm.eps)
        y = ops.groupnorm_apply(m, mean, rstd, _w(blk.norm.weight, x), _w(blk.norm.bias, x), G, act=N.ACT_LRELU, slope=0.1)
    return _track("grouped_residual_conv1d", y, x, *blk.parameters())


# ----------------------------------------------------------------------------------------------- discriminators
def _convs(seq):
    return [seq[i] for i in (0, 2, 4, 6, 8)]


def disc2d(x, blk):
    """discriminators.py:68-84."""
    with torch.no_grad():
        h = ops.mpd_fold(x, blk.period)
        for li, conv in enumerate(_convs(blk.conv_layers)):
            h = ops.conv2d(h, _w(conv.weight, x), _w(conv.bias, x), (1, 1), N.ACT_LRELU if li < 4 else N.ACT_NONE, 0.1)
    return _track("disc2d", h, x, *blk.parameters())


def disc1d(x, blk):
    """discriminators.py:109-117."""
    with torch.no_grad():
        h = ops.avgpool1d(x, blk.scale) if blk.scale > 1 else x
        for li, conv in enumerate(_convs(blk.conv_layers)):
            h = ops.conv1d(h, _w(conv.weight, x), _w(conv.bias, x), None, 1, 7, 1, 1,
                           N.ACT_LRELU if li < 4 else N.ACT_NONE, 0.1)
    return _track("disc1d", h, x, *blk.parameters())
