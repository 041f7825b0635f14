"""Differentiable building blocks of the vocoder path, expressed over the HIP ops in ``ops.py``.

Each public function corresponds to one reference module's forward (cited below) and is a
``torch.autograd.Function`` whose forward AND backward are HIP kernels (include/mi355x_vocoder.h);
torch only threads the graph together.  Parameters are taken as they live in the nn.Module (fp32
masters or already low precision) and are cast to the activation dtype through a version-keyed cache;
parameter gradients are produced in fp32 and cast to the parameter's dtype.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _native as N
from . import ops
from .ops import _zeros, _new_zeros

_cache = ops._ParamCache()
_ACT = {None: N.ACT_NONE, "none": N.ACT_NONE, "lrelu": N.ACT_LRELU, "tanh": N.ACT_TANH, "silu": N.ACT_SILU}

# The public functions below dispatch through the registered PyTorch operators (torch.ops.mi355x_vocoder.*, torch_ops.py), whose
# forward / backward are the static methods of the Function classes in this file.  DISPATCH = "direct" calls Function.apply
# instead (same kernels, no dispatcher round trip) - a debugging switch, the default is the operator path.
import os as _os
DISPATCH = _os.environ.get("MV_DISPATCH", "torch_ops")
_T = None


def _t():
    global _T
    if _T is None:
        from . import torch_ops
        _T = torch_ops
    return _T


def _w(p, like):
    return _cache.get(p, like.dtype)


def _to(g, p):
    """fp32 gradient -> the parameter's dtype/shape."""
    if g is None:
        return None
    g = g.view(p.shape)
    return g if g.dtype == p.dtype else ops.cast(g, p.dtype)


class _Pending(Function):
    """Forward-only ops (second-design blocks): asking for a gradient fails loudly instead of returning zeros."""

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
class _ODConv(Function):
    """odconv.py:73-108 / :172-205.  cfg = (transposed, stride, padding, output_padding, dilation, act, slope)."""

    @staticmethod
    def forward(ctx, x, kernels, bias, att_w, att_b, cfg, fused=None):
        transposed, stride, padding, out_pad, dilation, act, slope = cfg
        x = x if x.is_contiguous() else x.contiguous()
        K, C = att_w.shape[0], att_w.shape[1]
        wa, wk, wb = _w(att_w, x).view(K, C), _w(kernels, x), _w(bias, x)
        alpha, pooled = ops.odconv_attn(x, wa, _w(att_b, x), want_pooled=True)
        y = None
        if fused is not None and x.dtype != torch.float32:
            # 16-bit storage: the forward runs on the fused channels-last MFMA kernel with the alpha computed above
            y = ops.ntc_to_nct(fused.forward_cl(ops.nct_to_ntc(x), _cache, alpha=alpha, act=act, slope=slope))
        elif transposed:
            y = ops.conv_transpose1d(x, wk, wb, alpha, stride, padding, out_pad, dilation, act, slope)
        else:
            y = ops.conv1d(x, wk, wb, alpha, stride, padding, dilation, 1, act, slope)
        ctx.cfg = cfg
        ctx.fused = fused
        ctx.save_for_backward(x, kernels, bias, att_w, att_b, alpha, pooled, y if act != N.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, kernels, bias, att_w, att_b, alpha, pooled, y = ctx.saved_tensors
        transposed, stride, padding, out_pad, dilation, act, slope = ctx.cfg
        gy = gy if gy.is_contiguous() else gy.contiguous()
        g = ops.act_bwd(gy, y, act, slope) if act != N.ACT_NONE else gy
        K, C = att_w.shape[0], att_w.shape[1]
        wk = _w(kernels, x)
        ks = kernels.shape[3]
        B, Cin, Tin = x.shape
        Tout = g.shape[2]
        # data gradient: the adjoint convolution with the same (alpha-aggregated) kernels (skipped for a leaf input
        # such as the mel batch, whose gradient nobody consumes)
        gx = None
        mfma = transposed and ctx.fused is not None and x.dtype != torch.float32 and ctx.fused.dgrad_supported()
        gp = ctx.fused.pad_grad(g, Tin) if mfma else None      # time-padded channels-last gradient shared by both adjoint ops
        if ctx.needs_input_grad[0]:
            if mfma:
                gx = ctx.fused.dgrad(gp, alpha, Tin)     # the adjoint 2-tap ODConv on the fused MFMA kernel (None: tile too wide)
            if gx is not None:
                pass
            elif transposed:
                gx = ops.conv1d(g, wk, None, alpha, stride, padding, dilation, 1)
            else:
                opad = Tin - ((Tout - 1) * stride - 2 * padding + dilation * (ks - 1) + 1)
                gx = ops.conv_transpose1d(g, wk, None, alpha, stride, padding, opad, dilation)
        # kernel-bank gradient + d alpha = <per-sample wgrad, W_k>
        res = ctx.fused.wgrad(ops.nct_to_ntc(x), gp, wk, alpha) if mfma else None
        if res is not None:
            gw, galpha = res
        elif transposed:   # same kernel with the roles of input and output-gradient swapped
            gw, galpha = ops.conv1d_wgrad(g, x, wk, alpha, ks, stride, padding, dilation)
        else:
            gw, galpha = ops.conv1d_wgrad(x, g, wk, alpha, ks, stride, padding, dilation)
        gb = ops.bias_grad(g, alpha, _w(bias, x), galpha)
        gwa, gba, gm = ops.odconv_attn_bwd(alpha, galpha, pooled, _w(att_w, x).view(K, C), Tin)
        if gx is not None:
            ops.add_rowconst_(gx, gm)      # the pooling path: (1/T) Wa^T gz added to every time step
        return gx, _to(gw, kernels), _to(gb, bias), _to(gwa, att_w), _to(gba, att_b), None, None


def odconv_attention(x, att_w, att_b):
    """odconv.py:36-40,85 (inspection helper; not differentiable on its own)."""
    K, C = att_w.shape[0], att_w.shape[1]
    with torch.no_grad():
        return ops.odconv_attn(x, _w(att_w, x).view(K, C), _w(att_b, x))


def odconv1d(x, kernels, bias, att_w, att_b, stride=1, padding=0, dilation=1, act=None, slope=0.1, fused=None):
    if DISPATCH == "torch_ops":
        t = _t()
        return t.OPS.odconv1d(x, kernels, bias, att_w, att_b, stride, padding, 0, dilation, _ACT[act], slope, t.object_handle(fused))
    return _ODConv.apply(x, kernels, bias, att_w, att_b, (False, stride, padding, 0, dilation, _ACT[act], slope), fused)


def odconv_transpose1d(x, kernels, bias, att_w, att_b, stride=1, padding=0, output_padding=0, dilation=1,
                       act=None, slope=0.1, fused=None):
    if DISPATCH == "torch_ops":
        t = _t()
        return t.OPS.odconv_transpose1d(x, kernels, bias, att_w, att_b, stride, padding, output_padding, dilation, _ACT[act], slope,
                                        t.object_handle(fused))
    return _ODConv.apply(x, kernels, bias, att_w, att_b, (True, stride, padding, output_padding, dilation, _ACT[act], slope),
                         fused)


# ----------------------------------------------------------------------------------------------- plain convolutions
class _Conv1d(Function):
    """nn.Conv1d (+ fused activation).  cfg = (stride, padding, dilation, groups, act, slope).
    16-bit dense stride-1 'same' convs (the GRC / fusion convs of grc_lora.py:36-41,148) run forward, data gradient and
    weight gradient on the channels-last MFMA kernels; everything else on the generic NCT kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        stride, padding, dilation, groups, act, slope = cfg
        x = x if x.stride(2) == 1 else x.contiguous()
        ctx.cfg = cfg
        ctx.mfma = ops.mfma_conv1d_ok(x, weight, stride, padding, dilation, groups)
        if ctx.mfma:
            Cout, Cin, ks = weight.shape
            cop, cip = ops._up32(Cout), ops._up32(Cin)
            x_cl = ops.nct_to_ntc(x, cip)
            bp = None
            if bias is not None:
                bp = _w(bias, x)
                if cop != Cout:                       # parameter-sized glue: bias zero-padded to the MFMA granule
                    bp = _zeros(cop, device=x.device, dtype=x.dtype)
                    bp[:Cout] = _w(bias, x)
            y_cl = ops.dconv_cl(x_cl, ops.dconv_pack(weight.reshape(Cout, Cin, 1, ks), x.dtype, 0, cop, cip), bp, cop, 1, ks, dilation)
            y = ops.ntc_to_nct(y_cl, Cout)
            if act != N.ACT_NONE:                     # activation as its own (tiny) launch on the NCT result
                y = ops.act(y, act, slope)
            ctx.save_for_backward(x_cl, weight, bias, y if act != N.ACT_NONE else None)
            return y
        y = ops.conv1d(x, _w(weight, x), _w(bias, x), None, stride, padding, dilation, groups, act, slope)
        ctx.save_for_backward(x, weight, bias, y if act != N.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, y = ctx.saved_tensors
        stride, padding, dilation, groups, act, slope = ctx.cfg
        if ctx.mfma:
            Cout, Cin, ks = weight.shape
            cop, cip = ops._up32(Cout), ops._up32(Cin)
            gy = gy if gy.is_contiguous() else gy.contiguous()
            if act != N.ACT_NONE:
                gy = ops.act_bwd(gy, y, act, slope)
            g_cl = ops.nct_to_ntc(gy, cop)
            gx = gw = gb = None
            if ctx.needs_input_grad[0]:
                gx_cl = ops.dconv_cl(g_cl, ops.dconv_pack(weight.reshape(Cout, Cin, 1, ks), gy.dtype, 1, cop, cip), None, cip, 1, ks,
                                     dilation)
                gx = ops.ntc_to_nct(gx_cl, Cin)
            need_b = bias is not None and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                gwp, gbp = ops.dconv_wgrad_cl(x, g_cl, 1, ks, dilation, want_bias=True)
                gwp = gwp.view(cop, cip, ks)
                if (cop, cip) != (Cout, Cin):
                    gwc = torch.empty(Cout, Cin, ks, device=gwp.device, dtype=gwp.dtype)
                    ops.copy_rows(gwp.view(1, cop, cip * ks)[:, :Cout], gwc.view(1, Cout, Cin * ks))
                    gwp = gwc
                gw = _to(gwp, weight)
                if need_b:
                    gb = _to(gbp[:Cout], bias)
            elif need_b:
                gb = _to(ops.bias_grad(gy), bias)
            return gx, gw, gb, None
        if groups != 1:
            raise NotImplementedError("mi355x vocoder: backward of grouped conv1d is not available in this build")
        gy = gy if gy.is_contiguous() else gy.contiguous()
        g = ops.act_bwd(gy, y, act, slope) if act != N.ACT_NONE else gy
        ks = weight.shape[2]
        Tin, Tout = x.shape[2], g.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            opad = Tin - ((Tout - 1) * stride - 2 * padding + dilation * (ks - 1) + 1)
            gx = ops.conv_transpose1d(g, _w(weight, x), None, None, stride, padding, opad, dilation)
        if ctx.needs_input_grad[1]:
            gw, _ = ops.conv1d_wgrad(x, g, None, None, ks, stride, padding, dilation)
            gw = _to(gw, weight)
        if bias is not None and ctx.needs_input_grad[2]:
            gb = _to(ops.bias_grad(g), bias)
        return gx, gw, gb, None


def conv1d(x, weight, bias, stride=1, padding=0, dilation=1, groups=1, act=None, slope=0.1):
    if DISPATCH == "torch_ops":
        return _t().OPS.conv1d(x, weight, bias, stride, padding, dilation, groups, _ACT[act], slope)
    return _Conv1d.apply(x, weight, bias, (stride, padding, dilation, groups, _ACT[act], slope))


class _Conv2d(Function):
    """nn.Conv2d stride 1 (+ LeakyReLU) - discriminators.py:57-65."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        padding, act, slope = cfg
        y = ops.conv2d(x, _w(weight, x), _w(bias, x), padding, act, slope)
        ctx.cfg = cfg
        ctx.save_for_backward(x, weight, bias, y if act != N.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, y = ctx.saved_tensors
        (ph, pw), act, slope = ctx.cfg
        gy = gy if gy.is_contiguous() else gy.contiguous()
        g = ops.act_bwd(gy, y, act, slope) if act != N.ACT_NONE else gy
        kh, kw = weight.shape[2], weight.shape[3]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.conv2d(g, ops.conv2d_flip_weights(_w(weight, x)), None, (kh - 1 - ph, kw - 1 - pw))
        if ctx.needs_input_grad[1]:
            gw = _to(ops.conv2d_wgrad(x, g, kh, kw, ph, pw), weight)
        if bias is not None and ctx.needs_input_grad[2]:
            B, C, H, W = g.shape
            gb = _to(ops.bias_grad(g.view(B, C, H * W)), bias)
        return gx, gw, gb, None


def conv2d(x, weight, bias, padding=(1, 1), act=None, slope=0.1):
    if DISPATCH == "torch_ops":
        return _t().OPS.conv2d(x, weight, bias, padding[0], padding[1], _ACT[act], slope)
    return _Conv2d.apply(x, weight, bias, (tuple(padding), _ACT[act], slope))


# ----------------------------------------------------------------------------------------------- GroupNorm (+act, +dropout, +residual)
class _GroupNorm(Function):
    """y = act(GN(x)) * mask*scale + res   (grc_lora.py:58-59,68 and :161-163).  cfg = (G, eps, act, slope, mask_scale)."""

    @staticmethod
    def forward(ctx, x, gw, gb, res, mask, cfg):
        G, eps, act, slope, mask_scale = cfg
        x = x if x.stride(2) == 1 else x.contiguous()
        mean, rstd = ops.groupnorm_stats(x, G, eps)
        if res is not None and res.stride(2) != 1:
            res = res.contiguous()
        y = ops.groupnorm_apply(x, mean, rstd, _w(gw, x), _w(gb, x), G, act, slope, res, mask, mask_scale)
        ctx.cfg = cfg
        ctx.has_res = res is not None
        ctx.save_for_backward(x, gw, gb, mean, rstd, mask)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gw, gb, mean, rstd, mask = ctx.saved_tensors
        G, eps, act, slope, mask_scale = ctx.cfg
        gy = gy if gy.stride(2) == 1 else gy.contiguous()
        gx, dgw, dgb = ops.groupnorm_bwd(x, gy, mean, rstd, _w(gw, x), _w(gb, x), G, act, slope, mask, mask_scale)
        return gx, _to(dgw, gw), _to(dgb, gb), (gy if ctx.has_res else None), None, None


def group_norm(x, gw, gb, G, eps=1e-5, act=None, slope=0.1, res=None, mask=None, mask_scale=1.0):
    if DISPATCH == "torch_ops":
        return _t().OPS.group_norm(x, gw, gb, res, mask, G, eps, _ACT[act], slope, mask_scale)
    return _GroupNorm.apply(x, gw, gb, res, mask, (G, eps, _ACT[act], slope, mask_scale))


# ----------------------------------------------------------------------------------------------- FiLM / second-design FiLM
class _Film(Function):
    """grc_lora.py:108-129 (the condition is already cat/pad/truncated by FiLMLayer.condition)."""

    @staticmethod
    def forward(ctx, x, cond, proj_w, proj_b, feature_dim):
        cond_t = ops.cast(cond, x.dtype)
        proj = ops.linear(cond_t, _w(proj_w, x), _w(proj_b, x))
        y = ops.film(x, proj, feature_dim)
        ctx.F = feature_dim
        ctx.save_for_backward(x, cond_t, proj, proj_w, proj_b, cond)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, cond_t, proj, proj_w, proj_b, cond = ctx.saved_tensors
        gx, gproj = ops.film_bwd(x, gy, proj, ctx.F)
        gcond, gwp, gbp = ops.linear_bwd(cond_t, _w(proj_w, x), gproj, need_gx=ctx.needs_input_grad[1])
        return gx, (_to(gcond, cond) if gcond is not None else None), _to(gwp, proj_w), _to(gbp, proj_b), None


def film(x, cond, proj_w, proj_b, feature_dim):
    if DISPATCH == "torch_ops":
        return _t().OPS.film(x, cond, proj_w, proj_b, feature_dim)
    return _Film.apply(x, cond, proj_w, proj_b, feature_dim)


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def film2(x, spk, emo, scale_w, scale_b, shift_w, shift_b):
    """generator.py:187-199: (W_s e + b_s) * x + (W_h e + b_h), e = spk + emo.
    Training: the two projections are one FiLM projection [W_s; W_h] (parameter-sized concatenation) through `_Film`, so
    x, the embeddings and all four parameters receive gradients from the HIP backward kernels."""
    if _wants_grad(x, spk, emo, scale_w, scale_b, shift_w, shift_b):
        e = spk + emo                                             # [B, D] host-side glue
        return film(x, e, torch.cat([scale_w, shift_w], 0), torch.cat([scale_b, shift_b], 0), scale_w.shape[0])
    with torch.no_grad():
        e = ops.act(ops.cast(spk, x.dtype), N.ACT_NONE, res=ops.cast(emo, x.dtype))
        scale = ops.linear(e, _w(scale_w, x), _w(scale_b, x))
        shift = ops.linear(e, _w(shift_w, x), _w(shift_b, x))
        return ops.scale_shift(x, scale, shift)


# ----------------------------------------------------------------------------------------------- GRC + LoRA / MRF (generic shapes)
class _GrcFold(Function):
    """Parameter algebra of grc_lora.py:33-57 folded to (w_eff, b_eff); computed in the parameters' dtype."""

    @staticmethod
    def forward(ctx, conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w, proj_b, groups):
        w_eff, b_eff = ops.grc_fold_weights(conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w, proj_b, groups)
        ctx.groups = groups
        ctx.save_for_backward(conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w)
        return w_eff, b_eff

    @staticmethod
    def backward(ctx, g_weff, g_beff):
        conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w = ctx.saved_tensors
        g_weff = ops.cast(g_weff.contiguous(), torch.float32)
        g_beff = ops.cast(g_beff.contiguous(), torch.float32)
        gs = ops.grc_fold_bwd(g_weff, g_beff, conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w, ctx.groups)
        ps = (conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w)
        out = [_to(g, p) for g, p in zip(gs[:6], ps)]
        return (*out, _to(gs[6], conv_b), None)


class _Cat(Function):
    """torch.cat along channels (grc_lora.py:159) as strided copies."""

    @staticmethod
    def forward(ctx, *xs):
        B, T = xs[0].shape[0], xs[0].shape[2]
        ctx.sizes = [x.shape[1] for x in xs]
        out = torch.empty(B, sum(ctx.sizes), T, device=xs[0].device, dtype=xs[0].dtype)
        off = 0
        for x in xs:
            ops.copy_rows(x if x.stride(2) == 1 else x.contiguous(), out[:, off:off + x.shape[1]])
            off += x.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.sizes:
            o = torch.empty(g.shape[0], c, g.shape[2], device=g.device, dtype=g.dtype)
            ops.copy_rows(g[:, off:off + c], o)
            outs.append(o)
            off += c
        return tuple(outs)


class _BatchCat(Function):
    """torch.cat along the batch axis of two equally shaped tensors (one strided copy each); backward hands out views."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = (t if t.is_contiguous() else t.contiguous() for t in (a, b))
        out = torch.empty(2 * a.shape[0], *a.shape[1:], device=a.device, dtype=a.dtype)
        n = a.numel()
        ops.copy_rows(a.view(1, 1, n), out.view(2, 1, n)[0:1])
        ops.copy_rows(b.view(1, 1, n), out.view(2, 1, n)[1:2])
        ctx.B = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.B], g[ctx.B:]


def batch_cat(a, b):
    return _BatchCat.apply(a, b)


class _BatchSplit(Function):
    """The two halves of a batch as views; backward writes the two gradients into one buffer (two strided copies) instead of the
    zeros + copy per slice and the add that autograd's own slicing builds."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.B = x.shape, x.shape[0] // 2
        return x[:ctx.B], x[ctx.B:]

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        if ref is None:
            return None
        g = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype)
        n = g.numel() // 2
        halves = g.view(2, 1, n)
        for i, gh in enumerate((ga, gb)):
            if gh is None:
                halves[i:i + 1].zero_()
            else:
                ops.copy_rows((gh if gh.is_contiguous() else gh.contiguous()).view(1, 1, n), halves[i:i + 1])
        return g


def batch_split(x):
    return _BatchSplit.apply(x)


def _grc_generic(x, blk):
    """grc_lora.py:32-68 with the parameter algebra folded: conv_g + LoRA + 1x1 -> one dense dilated conv."""
    k, d = blk.kernel_size, blk.dilation
    if k % 2 == 0:
        raise RuntimeError("GRC_LoRA_Block: even kernel sizes make base/LoRA lengths differ (as in the reference)")
    w_eff, b_eff = _GrcFold.apply(blk.conv.weight, blk.conv.bias, blk.lora_A, blk.lora_B, blk.lora_scaling,
                                  blk.output_projection.weight, blk.output_projection.bias, blk.groups)
    v = conv1d(x, w_eff, b_eff, padding=(k - 1) * d // 2, dilation=d)
    if blk.in_channels != blk.out_channels:
        res = conv1d(x, blk.residual_proj.weight, blk.residual_proj.bias)
    else:
        res = x
    return group_norm(v, blk.norm.weight, blk.norm.bias, blk.norm_groups, blk.norm.eps, act="silu", res=res)


def grc_lora_block(x, blk, out=None, out_channel_offset=0):
    return _grc_generic(x if x.stride(2) == 1 else x.contiguous(), blk)


def dropout_mask(shape, p, device):
    """uint8 keep-mask of nn.Dropout(p) (grc_lora.py:151,162) from our own Philox4x32-10 kernel; the 64-bit key of each mask is
    drawn from torch's CPU generator, so `torch.manual_seed` (per rank) fixes the whole dropout stream."""
    from ctypes import c_void_p
    mask = torch.empty(shape, device=device, dtype=torch.uint8)
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    N.call("mv_dropout_mask", c_void_p(mask.data_ptr()), mask.numel(), float(p), seed, ops._stream())
    return mask


class _SplitChannels(Function):
    """[B, sum(sizes), T] -> contiguous [B, sizes[i], T] pieces (strided copies); backward writes the pieces' gradients
    back into one buffer."""

    @staticmethod
    def forward(ctx, x, sizes):
        ctx.sizes, ctx.shape = tuple(sizes), x.shape
        outs, off = [], 0
        for c in sizes:
            o = torch.empty(x.shape[0], c, x.shape[2], device=x.device, dtype=x.dtype)
            ops.copy_rows(x[:, off:off + c], o)
            outs.append(o)
            off += c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        g = torch.empty(ctx.shape, device=gs[0].device, dtype=gs[0].dtype)
        off = 0
        for c, gi in zip(ctx.sizes, gs):
            ops.copy_rows(gi if gi.stride(2) == 1 else gi.contiguous(), g[:, off:off + c])
            off += c
        return g, None


class _BranchNorm(Function):
    """The GroupNorm + SiLU + residual of all GRC branches of one MRF block (grc_lora.py:58-68) on the merged conv output
    u = [v_0 .. v_{n-1} | r_0 .. r_{n-1}] (channel blocks of `cpd`): branch i is SiLU(GN_i(v_i)) + r_i, written straight into
    channel block i of the concatenated [B, n*cpd, T] result - the kernels take channel-slice strides, so neither the split
    nor the concat costs a copy.  params = (w_0, b_0, ..., w_{n-1}, b_{n-1}); cfg = (n, cpd, G, eps)."""

    @staticmethod
    def forward(ctx, u, cfg, *params):
        nb, cpd, G, eps = cfg
        u = u if u.stride(2) == 1 else u.contiguous()
        B, _, T = u.shape
        out = torch.empty(B, nb * cpd, T, device=u.device, dtype=u.dtype)
        stats = []
        for i in range(nb):
            xs, rs = u[:, i * cpd:(i + 1) * cpd], u[:, (nb + i) * cpd:(nb + i + 1) * cpd]
            mean, rstd = ops.groupnorm_stats(xs, G, eps)
            ops.groupnorm_apply(xs, mean, rstd, _w(params[2 * i], u), _w(params[2 * i + 1], u), G, N.ACT_SILU, 0.1, rs,
                                out=out[:, i * cpd:(i + 1) * cpd])
            stats += [mean, rstd]
        ctx.cfg = cfg
        ctx.save_for_backward(u, *params, *stats)
        return out

    @staticmethod
    def backward(ctx, gy):
        nb, cpd, G, eps = ctx.cfg
        saved = ctx.saved_tensors
        u, params, stats = saved[0], saved[1:1 + 2 * nb], saved[1 + 2 * nb:]
        gy = gy if gy.stride(2) == 1 else gy.contiguous()
        gu = torch.empty_like(u)
        grads = []
        for i in range(nb):
            xs, gs_ = u[:, i * cpd:(i + 1) * cpd], gy[:, i * cpd:(i + 1) * cpd]
            gx, dgw, dgb = ops.groupnorm_bwd(xs, gs_, stats[2 * i], stats[2 * i + 1], _w(params[2 * i], u), _w(params[2 * i + 1], u),
                                             G, N.ACT_SILU, 0.1)
            ops.copy_rows(gx, gu[:, i * cpd:(i + 1) * cpd])
            ops.copy_rows(gs_, gu[:, (nb + i) * cpd:(nb + i + 1) * cpd])          # the residual passes the gradient through
            grads += [_to(dgw, params[2 * i]), _to(dgb, params[2 * i + 1])]
        return (gu, None, *grads)


def _mrf_merged_ok(x, blk):
    """All branch convs AND their residual 1x1 projections read the same x: for 16-bit storage they run as ONE dense
    'same' conv with kernel 2*max(d)+1 on the MFMA kernels (the unused taps are zero) - one launch, one data-gradient and
    one weight-gradient GEMM per block instead of six of each, and two layout transposes instead of twenty-four."""
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    gs = list(blk.conv_layers)
    if any(g.kernel_size != 3 or g.in_channels == g.out_channels for g in gs):
        return False
    return (2 * max(g.dilation for g in gs) + 1) in ops._MFMA_KS and gs[0].in_channels >= 16


def _mrf_merged_branches(x, blk):
    gs = list(blk.conv_layers)
    cpd, cin = gs[0].out_channels, gs[0].in_channels
    nb = len(gs)
    md = max(g.dilation for g in gs)
    kw, ctr = 2 * md + 1, md
    folds = [_GrcFold.apply(g.conv.weight, g.conv.bias, g.lora_A, g.lora_B, g.lora_scaling, g.output_projection.weight,
                            g.output_projection.bias, g.groups) for g in gs]
    # parameter-sized, differentiable assembly of the merged kernel: rows [0, nb*cpd) = the folded dilated branch convs,
    # rows [nb*cpd, 2*nb*cpd) = the residual projections (centre tap only).  One concat + one index_copy per tensor
    # (the scatter indices are cached on the block) instead of a slice assignment per branch and tap.
    dev = x.device
    cache = blk.__dict__.get("_mv_merge_idx")
    if cache is None or cache[0] != (dev, cin, cpd, kw, tuple(g.dilation for g in gs)):
        import numpy as np
        o, c, j = np.meshgrid(np.arange(cpd), np.arange(cin), np.arange(3), indexing="ij")
        iw = [(((i * cpd + o) * cin + c) * kw + ctr + (j - 1) * g.dilation).reshape(-1) for i, g in enumerate(gs)]
        o2, c2 = np.meshgrid(np.arange(cpd), np.arange(cin), indexing="ij")
        iw += [((((nb + i) * cpd + o2) * cin + c2) * kw + ctr).reshape(-1) for i in range(nb)]
        ib = [np.arange(i * cpd, (i + 1) * cpd) for i in range(2 * nb)]
        cache = ((dev, cin, cpd, kw, tuple(g.dilation for g in gs)),
                 torch.from_numpy(np.concatenate(iw)).to(dev), torch.from_numpy(np.concatenate(ib)).to(dev))
        blk.__dict__["_mv_merge_idx"] = cache
    _, idx_w, idx_b = cache
    wdt = folds[0][0].dtype
    src_w = torch.cat([f[0].reshape(-1) for f in folds] + [g.residual_proj.weight.reshape(-1).to(wdt) for g in gs])
    src_b = torch.cat([f[1].reshape(-1) for f in folds] + [g.residual_proj.bias.reshape(-1).to(wdt) for g in gs])
    W = _new_zeros(src_w, 2 * nb * cpd * cin * kw).index_copy(0, idx_w, src_w).view(2 * nb * cpd, cin, kw)
    bias = _new_zeros(src_b, 2 * nb * cpd).index_copy(0, idx_b, src_b)
    u = conv1d(x, W, bias, padding=md)
    if all(g.norm_groups == gs[0].norm_groups and g.norm.eps == gs[0].norm.eps for g in gs):
        prm = []
        for g in gs:
            prm += [g.norm.weight, g.norm.bias]
        return _BranchNorm.apply(u, (nb, cpd, gs[0].norm_groups, gs[0].norm.eps), *prm)      # already concatenated
    parts = _SplitChannels.apply(u, [cpd] * (2 * nb))
    return [group_norm(parts[i], g.norm.weight, g.norm.bias, g.norm_groups, g.norm.eps, act="silu", res=parts[nb + i])
            for i, g in enumerate(gs)]


def mrf_block(x, blk, force_generic=False, mask=None):
    """grc_lora.py:157-163.  Inference on 64-channel blocks of the generator's shape runs the fused MFMA kernel
    (csrc/mrf_fused.hip) in channels-last layout; training and any other shape run the generic differentiable path."""
    from .fused import mrf_fused_for
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in blk.parameters()))
    training_dropout = blk.training and blk.dropout.p > 0
    fz = None if (force_generic or needs_grad or training_dropout) else mrf_fused_for(blk)
    if fz is not None:
        with torch.no_grad():
            if DISPATCH == "torch_ops":
                t = _t()
                return t.OPS.grc_mrf_block(x, t.object_handle(fz))
            return ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x)))
    x = x if x.is_contiguous() else x.contiguous()
    branches = _mrf_merged_branches(x, blk) if _mrf_merged_ok(x, blk) else [_grc_generic(x, g) for g in blk.conv_layers]
    cat = branches if torch.is_tensor(branches) else _Cat.apply(*branches)
    f = conv1d(cat, blk.fusion.weight, blk.fusion.bias)
    scale, p = 1.0, blk.dropout.p
    if training_dropout:
        if mask is None:
            mask = dropout_mask(f.shape, p, x.device)
        scale = 1.0 / (1.0 - p)
    else:
        mask = None
    return group_norm(f, blk.norm.weight, blk.norm.bias, blk.norm_groups, blk.norm.eps, res=x, mask=mask, mask_scale=scale)


def _grouped_residual_dense_weights(blk):
    """Parameter algebra of generator.py:141-165 (parameter-sized, differentiable torch glue): the grouped conv and the
    shared-across-groups LoRA map alpha * (B A) become ONE dense block-diagonal kernel [C, C, k]; the 1x1 channel mixer
    and the residual `+ x` become ONE 1x1 kernel [C, 2C] over cat(u, x)."""
    G, C, k = blk.groups, blk.channels, blk.kernel_size
    cg = C // G
    wg = blk.grouped_conv.weight                                   # [C, C/G, k]
    M = blk.lora_alpha * (blk.lora_B @ blk.lora_A)                 # [C/G, C/G]
    dense = _new_zeros(wg, C, C, k)
    for gi in range(G):
        blkw = wg[gi * cg:(gi + 1) * cg].clone()
        blkw[:, :, k // 2] = blkw[:, :, k // 2] + M
        dense[gi * cg:(gi + 1) * cg, gi * cg:(gi + 1) * cg] = blkw
    eye = torch.eye(C, device=wg.device, dtype=wg.dtype).unsqueeze(-1)
    mix = torch.cat([blk.channel_mixer.weight, eye], dim=1)        # [C, 2C, 1]
    return dense, mix


def grouped_residual_conv1d(x, blk):
    """generator.py:141-172: LeakyReLU(GN_G(Conv1x1(conv_g(x) + alpha*LoRA_g(x)) + x)).
    Training runs the folded dense form through the differentiable conv / concat / GroupNorm functions (MFMA kernels for
    16-bit storage); inference keeps the grouped kernels."""
    if _wants_grad(x, *blk.parameters()):
        x = x if x.is_contiguous() else x.contiguous()
        k, d = blk.kernel_size, blk.dilation
        dense, mix = _grouped_residual_dense_weights(blk)
        u = conv1d(x, dense, blk.grouped_conv.bias, padding=(k - 1) * d // 2, dilation=d)
        m = conv1d(_Cat.apply(u, x), mix, blk.channel_mixer.bias)
        return group_norm(m, blk.norm.weight, blk.norm.bias, blk.groups, blk.norm.eps, act="lrelu", slope=0.1)
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
                             _zeros(1, 1, device=x.device, dtype=x.dtype)).view(C, C // G, 1)
        u = ops.conv1d(x, wl, None, None, 1, 0, 1, G, res=h)
        m = ops.conv1d(u, _w(blk.channel_mixer.weight, x), _w(blk.channel_mixer.bias, x), res=x)
        mean, rstd = ops.groupnorm_stats(m, G, blk.norm.eps)
        return ops.groupnorm_apply(m, mean, rstd, _w(blk.norm.weight, x), _w(blk.norm.bias, x), G, act=N.ACT_LRELU, slope=0.1)


# ----------------------------------------------------------------------------------------------- discriminators
class _AvgPool(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s, ctx.T = s, x.shape[2]
        return ops.avgpool1d(x, s)

    @staticmethod
    def backward(ctx, gy):
        return ops.avgpool1d_bwd(gy, ctx.T, ctx.s), None


class _MpdFold(Function):
    """discriminators.py:72-79: zero right-pad + view(B,C,period,T'/period)."""

    @staticmethod
    def forward(ctx, x, period):
        ctx.T = x.shape[2]
        return ops.mpd_fold(x, period)

    @staticmethod
    def backward(ctx, gy):
        B, C = gy.shape[0], gy.shape[1]
        g = gy.contiguous().view(B, C, -1)
        if g.shape[2] == ctx.T:
            return g, None
        gx = torch.empty(B, C, ctx.T, device=gy.device, dtype=gy.dtype)
        ops.copy_rows(g, gx)
        return gx, None


def _convs(seq):
    return [seq[i] for i in (0, 2, 4, 6, 8)]


def disc2d(x, blk):
    """discriminators.py:68-84."""
    B, C, T = x.shape
    if T % blk.period == 0:
        h = x.contiguous().view(B, C, blk.period, T // blk.period)   # exact multiple: the fold is a pure view
    else:
        h = _t().OPS.mpd_fold(x, blk.period) if DISPATCH == "torch_ops" else _MpdFold.apply(x, blk.period)
    from . import disc_fused
    if disc_fused.supported(x, blk):      # 16-bit storage: channels-last MFMA stack (csrc/disc_fused.hip)
        return disc_fused.disc_stack(h, blk)
    for li, conv in enumerate(_convs(blk.conv_layers)):
        h = conv2d(h, conv.weight, conv.bias, (1, 1), "lrelu" if li < 4 else None, 0.1)
    return h


def disc1d(x, blk):
    """discriminators.py:109-117."""
    h = x
    if blk.scale > 1:
        h = _t().OPS.avg_pool1d(x, blk.scale) if DISPATCH == "torch_ops" else _AvgPool.apply(x, blk.scale)
    from . import disc_fused
    if disc_fused.supported(x, blk):
        return disc_fused.disc_stack(h, blk)
    for li, conv in enumerate(_convs(blk.conv_layers)):
        h = conv1d(h, conv.weight, conv.bias, padding=7, act="lrelu" if li < 4 else None, slope=0.1)
    return h


# ----------------------------------------------------------------------------------------------- losses
class _Loss(Function):
    """weight * mean-reduced loss, value and gradient from one HIP pass (csrc/train.hip loss_kernel)."""

    @staticmethod
    def forward(ctx, x, y, kind, c, weight):
        need_gy = y is not None and y.requires_grad
        acc, gx, gyt = ops.loss_fwd_bwd(x, y, kind, c, weight, want_gx=True, want_gy=need_gy)
        ctx.save_for_backward(gx, gyt)
        return acc.view(())

    @staticmethod
    def backward(ctx, g):
        gx, gyt = ctx.saved_tensors
        gdev = g.reshape(1).float()
        gx = ops.scale_to(gx, 1.0, gdev)
        if gyt is not None:
            gyt = ops.scale_to(gyt, 1.0, gdev)
        return gx, gyt, None, None, None


def _loss(x, y, kind, c, weight):
    if DISPATCH == "torch_ops":
        return _t().OPS.gan_loss(x, y, kind, float(c), float(weight))
    return _Loss.apply(x, y, kind, float(c), float(weight))


def mse_const(x, c, weight=1.0):
    """mean((x - c)^2): F.mse_loss(x, ones/zeros) of complete_vocoder.py:104,157-158."""
    return _loss(x, None, 0, c, weight)


def l1(x, y, weight=1.0):
    """mean|x - y|: F.l1_loss of complete_vocoder.py:118,127."""
    return _loss(x, y, 1, 0.0, weight)


def mse(x, y, weight=1.0):
    return _loss(x, y, 4, 0.0, weight)


def hinge_g(x, weight=1.0):
    """mean(relu(1 - x)): conditioned_hifigan.py:263."""
    return _loss(x, None, 2, 0.0, weight)


def hinge_d_fake(x, weight=1.0):
    """mean(relu(1 + x)): conditioned_hifigan.py:265."""
    return _loss(x, None, 3, 0.0, weight)


class _MelL1(Function):
    """weight * mean |logmel(wave) - target| (kind 0) or mean squared (kind 1); defined by this build."""

    @staticmethod
    def forward(ctx, wave, target, fb, n_fft, hop, clampv, weight, kind):
        acc, _, gwave = ops.mel_loss(wave, fb, target, n_fft, hop, clampv, weight, backward=wave.requires_grad, kind=kind)
        ctx.save_for_backward(gwave)
        ctx.dtype = wave.dtype
        return acc.view(())

    @staticmethod
    def backward(ctx, g):
        (gwave,) = ctx.saved_tensors
        gw = ops.scale_to(gwave, 1.0, g.reshape(1).float())
        return (gw if ctx.dtype == torch.float32 else ops.cast(gw, ctx.dtype)), None, None, None, None, None, None, None


def _mel(wave, target, fb, n_fft, hop, clampv, weight, kind):
    if DISPATCH == "torch_ops":
        return _t().OPS.mel_loss(wave, target, fb, n_fft, hop, float(clampv), float(weight), kind)
    return _MelL1.apply(wave, target, fb, n_fft, hop, clampv, float(weight), kind)


def mel_l1(wave, target, fb, n_fft=1024, hop=256, clampv=1e-5, weight=1.0):
    return _mel(wave, target, fb, n_fft, hop, clampv, weight, 0)


def mel_mse(wave, target, fb, n_fft=1024, hop=256, clampv=1e-5, weight=1.0):
    """weight * mean (logmel(wave) - target)^2: the MSE mel term of conditioned_hifigan.py:238."""
    return _mel(wave, target, fb, n_fft, hop, clampv, weight, 1)


def mel_spectrogram(wave, fb, n_fft=1024, hop=256, clampv=1e-5):
    """log-mel [B, n_mels, T/hop] (fp32) of a waveform [B,1,T]."""
    with torch.no_grad():
        if DISPATCH == "torch_ops":
            return _t().OPS.mel_spectrogram(wave, fb, n_fft, hop, float(clampv))
        return ops.mel_loss(wave, fb, None, n_fft, hop, clampv, 1.0, backward=False, want_mel=True)[1]
