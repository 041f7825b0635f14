"""Tensor-level wrappers over the C ABI (include/mi355x_vocoder.h).

PyTorch is used here only for device memory (output allocation), the current HIP stream and
autograd bookkeeping; every arithmetic step on activation-sized data is a HIP kernel in
libmi355x_vocoder.so.  Inputs must live on the GPU: there is no CPU path.
"""
from __future__ import annotations

import weakref
from ctypes import c_void_p
from typing import Optional

import torch

from . import _native as N

_DT = {torch.float32: N.MV_F32, torch.bfloat16: N.MV_BF16, torch.float16: N.MV_F16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; use float32, bfloat16 or float16") from None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mi355x vocoder ops need GPU tensors: this path has no CPU fallback "
                               "(the CPU oracle lives under oracle/ and is test infrastructure only)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _c(t: Optional[torch.Tensor]):
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _need_gpu(x)
    if x.dtype == dtype:
        return x
    x = _c(x)
    y = torch.empty_like(x, dtype=dtype)
    if x.numel():
        N.call("mv_cast", _p(x), _dt(x), _p(y), _DT[dtype], x.numel(), _stream())
    return y


_EPOCH = [0]


_FINE = [0]             # bumps that are not attributed to one flat-arena optimizer (broadcast, load)


def bump_param_epoch(owner=None):
    """Called by code that updates parameters in place behind torch's back (flat-arena AdamW, broadcast): invalidates cached
    casts / packed weights.  `owner` (a flat-arena optimizer): only ITS parameters changed - caches that ask `param_epoch_of`
    keep the other optimizer's entries (the discriminator packs survive the generator's step and vice versa); `param_epoch()`
    still moves on every bump for the coarse caches."""
    _EPOCH[0] += 1
    if owner is None:
        _FINE[0] += 1
    else:
        owner._epoch = getattr(owner, "_epoch", 0) + 1


def param_epoch():
    return _EPOCH[0]


_SHADOW_OWNERS = {}     # id(parameter) -> weakref(flat-arena optimizer holding a 16-bit shadow of it)


def param_epoch_of(p):
    """Epoch of ONE parameter: (unattributed bumps, its flat-arena owner's own counter)."""
    ref = _SHADOW_OWNERS.get(id(p))
    opt = ref() if ref is not None else None
    if opt is not None and opt.owns(p):
        return (_FINE[0], getattr(opt, "_epoch", 0))
    return (_FINE[0], _EPOCH[0])     # no (live) owner: any bump may have touched it


def register_shadow_owner(opt, params):
    ref = weakref.ref(opt)
    for p in params:
        _SHADOW_OWNERS[id(p)] = ref


class _ParamCache:
    """Casts of parameters to the activation dtype.  An entry is valid only for the very same tensor
    object (weak reference - ids are recycled once a module is freed) at the same ``_version``, so an
    optimizer step (in-place update -> version bump) or ``load_state_dict`` invalidates it."""

    def __init__(self):
        self._d = {}

    def get(self, p: Optional[torch.Tensor], dtype: torch.dtype):
        if p is None:
            return None
        if p.dtype == dtype and p.is_contiguous():
            return p.detach()
        owner = _SHADOW_OWNERS.get(id(p))     # flat-arena optimizer: one cast launch per step covers every parameter
        if owner is not None and dtype != torch.float32:
            opt = owner()
            if opt is None:
                _SHADOW_OWNERS.pop(id(p), None)
            else:
                v = opt.shadow_view(p, dtype)      # validates that id(p) still names the very parameter it registered
                if v is not None:
                    return v
        key = (id(p), dtype)
        hit = self._d.get(key)
        if hit is not None and hit[0]() is p and hit[1] == (p._version, _EPOCH[0]) and hit[2].device == p.device:
            return hit[2]
        v = cast(p.detach(), dtype)
        if len(self._d) > 4096:  # drop entries whose parameter is gone
            self._d = {k: e for k, e in self._d.items() if e[0]() is not None}
        self._d[key] = (weakref.ref(p), (p._version, _EPOCH[0]), v)
        return v


def odconv_attn(x, w, bias, want_pooled=False):
    """alpha [B,K] fp32 (+ pooled mean [B,C] fp32)."""
    _need_gpu(x, w)
    x = _c(x)
    B, C, T = x.shape
    K = w.shape[0]
    alpha = torch.empty(B, K, device=x.device, dtype=torch.float32)
    pooled = torch.empty(B, C, device=x.device, dtype=torch.float32) if want_pooled else None
    N.call("mv_odconv_attn_fwd", _p(x), _p(_c(w)), _p(_c(bias)), _p(alpha), _p(pooled), B, C, T, K, _dt(x), _stream())
    return (alpha, pooled) if want_pooled else alpha


def conv1d(x, w, bias=None, alpha=None, stride=1, padding=0, dilation=1, groups=1, act=N.ACT_NONE, slope=0.1,
           res=None, out=None, out_channel_offset=0):
    """Generic (dynamic) conv1d.  w [Cout,Cin/g,ks] or [K,Cout,Cin/g,ks] with alpha [B,K].
    `out`/`out_channel_offset`: write into a channel slice of a wider [B,Ctot,Tout] buffer."""
    _need_gpu(x, w)
    B, Cin, Tin = x.shape
    if x.stride(2) != 1:
        x = x.contiguous()
    nb = 1
    if w.dim() == 4:
        nb = w.shape[0]
        Cout, ks = w.shape[1], w.shape[3]
    else:
        Cout, ks = w.shape[0], w.shape[2]
    Tout = (Tin + 2 * padding - dilation * (ks - 1) - 1) // stride + 1
    if Tout <= 0:
        raise RuntimeError(f"conv1d: non-positive output length {Tout}")
    if out is None:
        out = torch.empty(B, Cout, Tout, device=x.device, dtype=x.dtype)
        ysl = out
    else:
        ysl = out[:, out_channel_offset:out_channel_offset + Cout]
    if res is not None:
        assert res.shape == ysl.shape and res.stride() == ysl.stride(), "res must share y's layout"
    N.call("mv_conv1d_fwd", _p(x), _p(_c(w)), _p(_c(bias)), _p(alpha), _p(res), _p(ysl),
           B, Cin, Tin, Cout, Tout, ks, stride, padding, dilation, groups, nb, act, float(slope),
           x.stride(0), x.stride(1), ysl.stride(0), ysl.stride(1), _dt(x), _stream())
    return out


def conv_transpose1d(x, w, bias=None, alpha=None, stride=1, padding=0, output_padding=0, dilation=1,
                     act=N.ACT_NONE, slope=0.1):
    """Generic (dynamic) transposed conv.  w [Cin,Cout,ks] or [K,Cin,Cout,ks]."""
    _need_gpu(x, w)
    x = _c(x)
    B, Cin, Tin = x.shape
    nb = 1
    if w.dim() == 4:
        nb = w.shape[0]
        Cout, ks = w.shape[2], w.shape[3]
    else:
        Cout, ks = w.shape[1], w.shape[2]
    Tout = (Tin - 1) * stride - 2 * padding + dilation * (ks - 1) + output_padding + 1
    y = torch.empty(B, Cout, Tout, device=x.device, dtype=x.dtype)
    N.call("mv_conv_transpose1d_fwd", _p(x), _p(_c(w)), _p(_c(bias)), _p(alpha), _p(y),
           B, Cin, Tin, Cout, Tout, ks, stride, padding, dilation, nb, act, float(slope), _dt(x), _stream())
    return y


def conv2d(x, w, bias=None, padding=(1, 1), act=N.ACT_NONE, slope=0.1):
    _need_gpu(x, w)
    x = _c(x)
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    ph, pw = padding
    y = torch.empty(B, Cout, H + 2 * ph - kh + 1, W + 2 * pw - kw + 1, device=x.device, dtype=x.dtype)
    N.call("mv_conv2d_fwd", _p(x), _p(_c(w)), _p(_c(bias)), _p(y), B, Cin, H, W, Cout, kh, kw, ph, pw,
           act, float(slope), _dt(x), _stream())
    return y


def groupnorm_stats(x, G, eps=1e-5):
    _need_gpu(x)
    B, C, T = x.shape
    assert x.stride(2) == 1
    mean = torch.empty(B, G, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    N.call("mv_groupnorm_stats", _p(x), _p(mean), _p(rstd), B, C, T, G, float(eps), x.stride(0), x.stride(1),
           _dt(x), _stream())
    return mean, rstd


def groupnorm_apply(x, mean, rstd, gw, gb, G, act=N.ACT_NONE, slope=0.1, res=None, mask=None, mask_scale=1.0,
                    out=None):
    B, C, T = x.shape
    assert x.stride(2) == 1 and (res is None or res.stride(2) == 1)
    if out is None:
        out = torch.empty(B, C, T, device=x.device, dtype=x.dtype)
    rs0, rs1 = (res.stride(0), res.stride(1)) if res is not None else (0, 0)
    N.call("mv_groupnorm_apply", _p(x), _p(mean), _p(rstd), _p(_c(gw)), _p(_c(gb)), _p(res), _p(mask),
           float(mask_scale), _p(out), B, C, T, G, act, float(slope), x.stride(0), x.stride(1), rs0, rs1,
           out.stride(0), out.stride(1), _dt(x), _stream())
    return out


def grc_fold_weights(conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w, proj_b, groups):
    _need_gpu(conv_w)
    Cout, cin_g, ks = conv_w.shape
    Cin = cin_g * groups
    rank = lora_A.shape[1]
    w_eff = torch.empty(Cout, Cin, ks, device=conv_w.device, dtype=conv_w.dtype)
    b_eff = torch.empty(Cout, device=conv_w.device, dtype=conv_w.dtype)
    N.call("mv_grc_fold_weights", _p(_c(conv_w)), _p(_c(conv_b)), _p(_c(lora_A)), _p(_c(lora_B)), _p(_c(lora_scaling)),
           _p(_c(proj_w)), _p(_c(proj_b)), _p(w_eff), _p(b_eff), Cin, Cout, ks, groups, rank, _dt(conv_w), _stream())
    return w_eff, b_eff


def linear(x, w, b=None):
    _need_gpu(x, w)
    x = _c(x)
    M, Kd = x.shape
    Nn = w.shape[0]
    if w.dim() != 2 or w.shape[1] != Kd or (b is not None and b.numel() != Nn) or w.dtype != x.dtype:
        raise RuntimeError(f"linear: x {tuple(x.shape)} {x.dtype} cannot be multiplied with weight {tuple(w.shape)} {w.dtype}")
    y = torch.empty(M, Nn, device=x.device, dtype=x.dtype)
    N.call("mv_linear_fwd", _p(x), _p(_c(w)), _p(_c(b)), _p(y), M, Nn, Kd, _dt(x), _stream())
    return y


def film(x, proj, F):
    _need_gpu(x, proj)
    x = _c(x)
    B, C, T = x.shape
    y = torch.empty_like(x)
    N.call("mv_film_fwd", _p(x), _p(_c(proj)), _p(y), B, C, T, F, _dt(x), _stream())
    return y


def scale_shift(x, scale, shift):
    _need_gpu(x)
    x = _c(x)
    B, C, T = x.shape
    y = torch.empty_like(x)
    N.call("mv_scale_shift_fwd", _p(x), _p(_c(scale)), _p(_c(shift)), _p(y), B, C, T, _dt(x), _stream())
    return y


def act(x, kind, slope=0.1, res=None):
    _need_gpu(x)
    x = _c(x)
    y = torch.empty_like(x)
    N.call("mv_act_fwd", _p(x), _p(_c(res)), _p(y), x.numel(), kind, float(slope), _dt(x), _stream())
    return y


def avgpool1d(x, s):
    _need_gpu(x)
    x = _c(x)
    B, C, T = x.shape
    y = torch.empty(B, C, T // s, device=x.device, dtype=x.dtype)
    N.call("mv_avgpool1d_fwd", _p(x), _p(y), B * C, T, s, _dt(x), _stream())
    return y


def mpd_fold(x, period, want_index=False):
    """[B,C,T] -> [B,C,P,ceil(T/P)] (zero right-pad + view; discriminators.py:72-79)."""
    _need_gpu(x)
    x = _c(x)
    B, C, T = x.shape
    Tp = T if T % period == 0 else T + (period - T % period)
    index = torch.empty(Tp, device=x.device, dtype=torch.int64) if want_index else None
    if Tp == T and not want_index:
        y = x  # the fold of an exact multiple is a pure view
    else:
        y = torch.empty(B, C, Tp, device=x.device, dtype=x.dtype)
        N.call("mv_mpd_fold", _p(x), _p(y), _p(index), B * C, T, period, _dt(x), _stream())
    y = y.view(B, C, period, Tp // period)
    return (y, index.view(period, Tp // period)) if want_index else y


def nct_to_ntc(x, cpad=None):
    """[B,C,T] -> channels-last [B,T,C] (or [B,T,cpad] with zero channels C..cpad-1)."""
    _need_gpu(x)
    x = _c(x)
    B, C, T = x.shape
    if cpad is None or cpad == C:
        y = torch.empty(B, T, C, device=x.device, dtype=x.dtype)
        N.call("mv_nct_to_ntc", _p(x), _p(y), B, C, T, _dt(x), _stream())
    else:
        y = torch.empty(B, T, cpad, device=x.device, dtype=x.dtype)
        N.call("mv_nct_to_ntc_pad", _p(x), _p(y), B, C, T, cpad, _dt(x), _stream())
    return y


def ntc_to_nct(x, c=None):
    """channels-last [B,T,Cp] -> [B,C,T] (first C channels when c is given)."""
    _need_gpu(x)
    x = _c(x)
    B, T, Cp = x.shape
    if c is None or c == Cp:
        y = torch.empty(B, Cp, T, device=x.device, dtype=x.dtype)
        N.call("mv_ntc_to_nct", _p(x), _p(y), B, Cp, T, _dt(x), _stream())
    else:
        y = torch.empty(B, c, T, device=x.device, dtype=x.dtype)
        N.call("mv_ntc_to_nct_crop", _p(x), _p(y), B, c, T, Cp, _dt(x), _stream())
    return y


# ------------------------------------------------------------------------------------------------ backward / training ops
def _zeros(*shape, **kw):
    """torch.zeros as a FILL KERNEL: torch.zeros (and hipMemsetAsync) become memset nodes under stream capture, and on this ROCm
    build a replayed graph does not reliably order a memset node in front of the kernel node that follows it (common.h:
    mvi_zero_async; found with a captured training step whose accumulating kernels started from stale buffers)."""
    return torch.empty(*shape, **kw).fill_(0)


def _new_zeros(t, *shape, **kw):
    return t.new_empty(*shape, **kw).fill_(0)


def _f32(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)

# ----------------------------------------------------------------------------------------------- channels-last MFMA convs
_MFMA_KS = (1, 3, 5, 7, 11, 15)


def _up32(c):
    return (c + 31) // 32 * 32


def mfma_conv1d_ok(x, weight, stride, padding, dilation, groups) -> bool:
    """True when a Conv1d can run on the channels-last MFMA kernels (csrc/disc_fused.hip): 16-bit storage, stride 1,
    'same' padding, dense.  Channel counts are zero-padded to the 32-channel MFMA granule."""
    if x.dtype not in (torch.bfloat16, torch.float16) or not x.is_cuda:
        return False
    Cout, Cin, ks = weight.shape
    return (groups == 1 and stride == 1 and ks in _MFMA_KS and padding == dilation * (ks - 1) // 2
            and (ks - 1) * dilation <= 64 and Cin >= 16 and Cout >= 1)


def dconv_pack(w4, dtype, flip, coutp=None, cinp=None):
    """[Cout,Cin,kh,kw] weights -> MFMA A-fragment order (flip=1: the data-gradient operator), zero-padded to
    coutp x cinp channels."""
    Cout, Cin, kh, kw = w4.shape
    coutp, cinp = coutp or Cout, cinp or Cin
    w4 = w4.detach().contiguous()
    buf = torch.empty(N.lib().mv_dconv_packed_bytes(coutp, cinp, kh, kw, _DT[dtype]), dtype=torch.uint8, device=w4.device)
    N.call("mv_dconv_pack_pad", _p(w4), _DT[w4.dtype], _p(buf), Cout, Cin, kh, kw, coutp, cinp, int(flip), _DT[dtype], _stream())
    return buf


def dconv_cl(x_cl, packed, bias, Cout, kh, kw, dil=1, act=N.ACT_NONE, slope=0.1, act_save=None):
    """x_cl [B,H,W,Cin] (or [B,W,Cin]) channels-last -> [.., Cout]."""
    shp = x_cl.shape
    B, H, W, Cin = (shp[0], 1, shp[1], shp[2]) if x_cl.dim() == 3 else shp
    y = torch.empty(*shp[:-1], Cout, device=x_cl.device, dtype=x_cl.dtype)
    N.call("mv_dconv_cl_fwd", _p(x_cl), _p(packed), _p(bias), _p(act_save), _p(y), B, H, W, Cin, Cout, kh, kw, dil, act, float(slope),
           _dt(x_cl), _stream())
    return y


_WGRAD_WS = {}        # (device, stream, floats) -> workspace kept zero between weight-gradient calls


def wgrad_cl_into(x_cl, g_cl, gw, gb, B, H, W, Cin, Cout, kh, kw, dil):
    """mv_dconv_wgrad_cl_pz on a persistent workspace (zero between calls: its reorder pass clears what it reads); falls back to the
    fill-per-call entry while a stream is being captured.  gb may be None."""
    dev = x_cl.device
    n = kh * kw * Cout * Cin + Cout
    if torch.cuda.is_current_stream_capturing():
        ws = _f32(n, device=dev)
        N.call("mv_dconv_wgrad_cl", _p(x_cl), _p(g_cl), _p(gw), _p(ws[n - Cout:]) if gb is not None else None, _p(ws), B, H, W, Cin, Cout,
               kh, kw, dil, _dt(x_cl), _stream())
        if gb is not None:
            gb.copy_(ws[n - Cout:])
        return
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, n)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = _zeros(n, device=dev, dtype=torch.float32)
    try:
        N.call("mv_dconv_wgrad_cl_pz", _p(x_cl), _p(g_cl), _p(gw), _p(gb), _p(ws), B, H, W, Cin, Cout, kh, kw, dil, _dt(x_cl), _stream())
    except Exception:
        _WGRAD_WS.pop(key, None)      # the workspace may be dirty: never reuse it
        raise


def dconv_wgrad_cl(x_cl, g_cl, kh, kw, dil=1, want_bias=False):
    """fp32 [Cout,Cin,kh,kw] = sum_pos g x (transposed-LDS-read MFMA GEMM); with want_bias also gb fp32 [Cout] = sum_pos g."""
    shp = x_cl.shape
    B, H, W, Cin = (shp[0], 1, shp[1], shp[2]) if x_cl.dim() == 3 else shp
    Cout = g_cl.shape[-1]
    gw = _f32(Cout, Cin, kh, kw, device=x_cl.device)
    gb = _f32(Cout, device=x_cl.device) if want_bias else None
    wgrad_cl_into(x_cl, g_cl, gw, gb, B, H, W, Cin, Cout, kh, kw, dil)
    return (gw, gb) if want_bias else gw


def colsum_cl(g_cl):
    C = g_cl.shape[-1]
    out = _f32(C, device=g_cl.device)
    N.call("mv_colsum_cl", _p(g_cl), _p(out), g_cl.numel() // C, C, _dt(g_cl), _stream())
    return out



def act_bwd(gy, y, kind, slope=0.1):
    gy, y = _c(gy), _c(y)
    gx = torch.empty_like(gy)
    N.call("mv_act_bwd", _p(gy), _p(y), _p(gx), gy.numel(), kind, float(slope), _dt(gy), _stream())
    return gx


def conv1d_wgrad(x, gy, w, alpha, ks, stride, padding, dilation):
    """x [B,Cin,Tin], gy [B,Cout,Tout] -> gw fp32 [nb?,Cout,Cin,ks] (+ galpha fp32 [B,nb] when alpha is given)."""
    B, Cin, Tin = x.shape
    _, Cout, Tout = gy.shape
    assert x.stride(2) == 1 and gy.stride(2) == 1
    nb = 1 if alpha is None else alpha.shape[1]
    gw = _f32(*((nb, Cout, Cin, ks) if alpha is not None else (Cout, Cin, ks)), device=x.device)
    galpha = _zeros(B, nb, device=x.device, dtype=torch.float32) if alpha is not None else None
    wsb = N.lib().mv_conv1d_wgrad_workspace_bytes(B, Cin, Cout, ks, nb)
    ws = torch.empty(wsb // 4, device=x.device, dtype=torch.float32) if wsb else None
    N.call("mv_conv1d_wgrad", _p(x), _p(gy), _p(_c(w)) if w is not None else None, _p(alpha), _p(gw), _p(galpha), _p(ws),
           B, Cin, Tin, Cout, Tout, ks, stride, padding, dilation, nb, x.stride(0), x.stride(1), gy.stride(0), gy.stride(1),
           _dt(x), _stream())
    return gw, galpha


def bias_grad(gy, alpha=None, bias=None, galpha=None):
    """gy [B,C,T] (or [B,C,H,W] flattened by the caller) -> gbias fp32 [K,C] / [C]; adds the bias term to galpha."""
    B, C, T = gy.shape
    assert gy.stride(2) == 1
    K = 1 if alpha is None else alpha.shape[1]
    ws = _f32(B * C, device=gy.device)
    gb = _f32(K, C, device=gy.device)
    N.call("mv_bias_grad", _p(gy), _p(alpha), _p(_c(bias)) if bias is not None else None, _p(ws), _p(gb), _p(galpha),
           B, C, T, K, gy.stride(0), gy.stride(1), _dt(gy), _stream())
    return gb if alpha is not None else gb[0]


def odconv_attn_bwd(alpha, galpha, pooled, wa, T):
    B, K = alpha.shape
    C = pooled.shape[1]
    gwa, gba, gm = _f32(K, C, device=alpha.device), _f32(K, device=alpha.device), _f32(B, C, device=alpha.device)
    N.call("mv_odconv_attn_bwd", _p(alpha), _p(galpha), _p(pooled), _p(_c(wa)), _p(gwa), _p(gba), _p(gm), B, C, T, K,
           _dt(wa), _stream())
    return gwa, gba, gm


def add_rowconst_(x, v):
    B, C, T = x.shape
    assert x.is_contiguous()
    N.call("mv_add_rowconst", _p(x), _p(v), B * C, T, _dt(x), _stream())
    return x


def groupnorm_bwd(x, gy, mean, rstd, gw, gb, G, act=N.ACT_NONE, slope=0.1, mask=None, mask_scale=1.0):
    B, C, T = x.shape
    assert x.stride(2) == 1 and gy.stride(2) == 1
    gz = torch.empty(B, C, T, device=x.device, dtype=x.dtype)
    ws = _f32(2 * B * G + 2 * B * C, device=x.device)
    gx = torch.empty(B, C, T, device=x.device, dtype=x.dtype)
    dgw, dgb = _f32(C, device=x.device), _f32(C, device=x.device)
    N.call("mv_groupnorm_bwd", _p(x), _p(gy), _p(mean), _p(rstd), _p(_c(gw)), _p(_c(gb)), _p(mask), float(mask_scale),
           act, float(slope), _p(gz), _p(ws), _p(gx), _p(dgw), _p(dgb), B, C, T, G, x.stride(0), x.stride(1),
           gy.stride(0), gy.stride(1), _dt(x), _stream())
    return gx, dgw, dgb


def film_bwd(x, gy, proj, F):
    x, gy = _c(x), _c(gy)
    B, C, T = x.shape
    gx = torch.empty_like(x)
    gproj = _f32(B, 2 * F, device=x.device)
    N.call("mv_film_bwd", _p(x), _p(gy), _p(_c(proj)), _p(gx), _p(gproj), B, C, T, F, _dt(x), _stream())
    return gx, gproj


def linear_bwd(x, w, gy32, need_gx=True):
    x = _c(x)
    M, Kd = x.shape
    Nn = w.shape[0]
    gx = _f32(M, Kd, device=x.device) if need_gx else None
    gw, gb = _f32(Nn, Kd, device=x.device), _f32(Nn, device=x.device)
    N.call("mv_linear_bwd", _p(x), _p(_c(w)), _p(_c(gy32)), _p(gx), _p(gw), _p(gb), M, Nn, Kd, _dt(x), _stream())
    return gx, gw, gb


def avgpool1d_bwd(gy, T, s):
    gy = _c(gy)
    B, C, To = gy.shape
    gx = torch.empty(B, C, T, device=gy.device, dtype=gy.dtype)
    N.call("mv_avgpool1d_bwd", _p(gy), _p(gx), B * C, T, s, _dt(gy), _stream())
    return gx


def copy_rows(src, dst):
    """dst[b, c, :n] = src[b, c, :n] for 3-D tensors with unit inner stride (channel concat / slice / fold backward)."""
    B, C, n = src.shape[0], src.shape[1], min(src.shape[2], dst.shape[2])
    assert src.stride(2) == 1 and dst.stride(2) == 1 and dst.shape[0] == B and dst.shape[1] == C
    N.call("mv_copy2d", _p(src), _p(dst), n, B, C, src.stride(0), src.stride(1), dst.stride(0), dst.stride(1), _dt(src), _stream())
    return dst


def conv2d_flip_weights(w):
    w = _c(w)
    Cout, Cin, kh, kw = w.shape
    wt = torch.empty(Cin, Cout, kh, kw, device=w.device, dtype=w.dtype)
    N.call("mv_conv2d_flip_weights", _p(w), _p(wt), Cout, Cin, kh, kw, _dt(w), _stream())
    return wt


def conv2d_wgrad(x, gy, kh, kw, ph, pw):
    x, gy = _c(x), _c(gy)
    B, Cin, H, W = x.shape
    Cout = gy.shape[1]
    gw = _f32(Cout, Cin, kh, kw, device=x.device)
    N.call("mv_conv2d_wgrad", _p(x), _p(gy), _p(gw), B, Cin, H, W, Cout, kh, kw, ph, pw, _dt(x), _stream())
    return gw


def grc_fold_bwd(g_weff, g_beff, conv_w, conv_b, lora_A, lora_B, lora_scaling, proj_w, groups):
    Cout, cin_g, ks = conv_w.shape
    Cin, rank, dev = cin_g * groups, lora_A.shape[1], conv_w.device
    outs = [_f32(*s, device=dev) for s in ((Cout, cin_g, ks), (Cout,), (Cin, rank), (rank, Cout), (1,), (Cout, Cout, 1), (Cout,))]
    N.call("mv_grc_fold_bwd", _p(_c(g_weff)), _p(_c(g_beff)), _p(_c(conv_w)), _p(_c(conv_b)), _p(_c(lora_A)), _p(_c(lora_B)),
           _p(_c(lora_scaling)), _p(_c(proj_w)), *[_p(o) for o in outs], Cin, Cout, ks, groups, rank, _dt(conv_w), _stream())
    return outs


class _ScalarArena:
    """fp32 accumulators for the loss kernels: one zeroed element each (a fill kernel: 16 bytes).  They used to be slots of one
    zero-filled block; a loss returned by an operator is then a view at a non-zero storage offset, which a fake (meta) kernel cannot
    reproduce (torch.library.opcheck), and a captured step needed fresh ones anyway."""

    def take(self, device):
        return _zeros(1, device=device, dtype=torch.float32)


_scalars = _ScalarArena()


def loss_fwd_bwd(x, y, kind, c=0.0, weight=1.0, want_gx=True, want_gy=False, acc=None):
    """Returns (loss_acc fp32 [1], gx, gy).  acc: existing accumulator to add into."""
    x = _c(x)
    y = _c(y)
    if acc is None:
        acc = _scalars.take(x.device)
    gx = torch.empty_like(x) if want_gx else None
    gyt = torch.empty_like(x) if want_gy else None
    N.call("mv_loss_fwd_bwd", _p(x), _p(y), float(c), float(weight), _p(acc), _p(gx), _p(gyt), x.numel(), kind, _dt(x), _stream())
    return acc, gx, gyt


def scale_to(x, factor=1.0, factor_dev=None):
    """x * factor * factor_dev[0] into a new tensor (one launch; x is left untouched)."""
    x = _c(x)
    y = torch.empty_like(x)
    N.call("mv_scale_to", _p(x), _p(y), _p(factor_dev), float(factor), x.numel(), _dt(x), _stream())
    return y


def scale_(x, factor=1.0, factor_dev=None):
    assert x.is_contiguous()
    N.call("mv_scale", _p(x), _p(factor_dev), float(factor), x.numel(), _dt(x), _stream())
    return x


def mel_loss(wave, fb, target=None, n_fft=1024, hop=256, clampv=1e-5, weight=1.0, backward=False, want_mel=False,
             kind=0):
    # the C entry point receives no tensor sizes for target / fb: a wrong shape would be an out-of-bounds device read or a
    # silently misaligned loss, where the reference's F.l1_loss / F.mse_loss raise - so raise here
    if wave.dim() != 3 or wave.shape[1] != 1:
        raise ValueError(f"mel_loss: wave must be [B, 1, T], got {tuple(wave.shape)}")
    wave = _c(wave)
    B, _, T = wave.shape
    if T % hop != 0:
        raise ValueError(f"mel_loss: T = {T} is not a multiple of hop = {hop}")
    if fb.dim() != 2 or fb.shape[1] != n_fft // 2 + 1 or fb.dtype != torch.float32:
        raise ValueError(f"mel_loss: filterbank must be fp32 [n_mels, {n_fft // 2 + 1}] for n_fft = {n_fft}, got {tuple(fb.shape)} {fb.dtype}")
    fb = _c(fb)
    n_mels = fb.shape[0]
    if target is not None and tuple(target.shape) != (B, n_mels, T // hop):
        raise ValueError(f"mel_loss: target must be [B, n_mels, T/hop] = {(B, n_mels, T // hop)}, got {tuple(target.shape)}")
    if (backward or not want_mel) and target is None:
        raise ValueError("mel_loss: a loss (or its gradient) needs a target mel")
    acc = _scalars.take(wave.device)
    mel = _f32(B, n_mels, T // hop, device=wave.device) if want_mel else None
    gwave = _zeros(B, 1, T, device=wave.device, dtype=torch.float32) if backward else None
    if target is not None:
        target = cast(_c(target), torch.float32)
    N.call("mv_mel_loss", _p(wave), _p(fb), _p(target), _p(mel), _p(acc), _p(gwave), B, T, n_fft, hop, n_mels,
           float(clampv), float(weight), int(kind), int(backward), _dt(wave), _stream())
    return acc, mel, gwave
