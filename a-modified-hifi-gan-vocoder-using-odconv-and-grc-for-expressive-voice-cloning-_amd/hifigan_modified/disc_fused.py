"""Channels-last MFMA path of one discriminator stack (forward + backward), 16-bit storage.

One ``torch.autograd.Function`` runs all five layers of a Discriminator2D / Discriminator1D
(discriminators.py:56-66, :97-107): first layer (1 -> 32) and head (256 -> 1) as small dedicated kernels, the three
wide layers on MFMA (``mv_dconv_cl_fwd``), their data gradients with the same kernel on flipped weights (LeakyReLU'
fused in the epilogue) and their weight gradients with the transposed-LDS-read GEMM (``mv_dconv_wgrad_cl``).
fp32 storage keeps using the generic kernels (functional.disc2d / disc1d).
"""
from __future__ import annotations

import weakref
from ctypes import c_void_p

import torch
from torch.autograd import Function

from . import _native as N
from . import ops

_P = lambda t: None if t is None else c_void_p(t.data_ptr())


class _PackCache:
    """Packed weights per (parameter, dtype, kind).  Entries are validated against the LIVE parameter object through a
    weak reference (Python ids and device addresses are recycled once a module is freed) plus its version / the
    parameter epoch, so a freed module can never alias another one's packed buffer."""

    def __init__(self):
        self.d = {}

    def _lookup(self, param, kind):
        key = (id(param), kind)
        ver = (param._version, ops.param_epoch_of(param), param.data_ptr())
        hit = self.d.get(key)
        if hit is not None and hit[0]() is param and hit[1] == ver:
            return key, ver, hit[2]
        return key, ver, None

    def _store(self, key, param, ver, buf):
        if len(self.d) > 1024:
            self.d = {k: e for k, e in self.d.items() if e[0]() is not None}
        self.d[key] = (weakref.ref(param), ver, buf)
        return buf

    def refresh_owned(self, opt):
        """After `opt.step()` (flat-arena AdamW): re-pack every plain conv pack whose fp32 master belongs to `opt` with ONE launch and
        mark the entries current - instead of one pack launch per layer and direction as the layers run."""
        import numpy as np
        ents = []
        for key, e in self.d.items():
            kind, p = key[1], e[0]()
            if (p is None or not isinstance(kind, tuple) or len(kind) != 2 or kind[0] not in (torch.bfloat16, torch.float16)
                    or p.dtype != torch.float32 or not p.is_contiguous() or not opt.owns(p)):
                continue
            w = _as4d(p)
            if w.shape[0] % 16 or w.shape[1] % 32:
                continue
            ents.append((key, p, e[2], kind, w.shape))
        if not ents:
            return 0
        sig = tuple((id(p), p.data_ptr(), buf.data_ptr()) for _, p, buf, _, _ in ents)
        hit = self.__dict__.setdefault("_mp", {}).get(id(opt))
        if hit is None or hit[0] != sig:
            d = np.zeros(len(ents), dtype=[("src", "u8"), ("dst", "u8"), ("cout", "i4"), ("cin", "i4"), ("kh", "i4"), ("kw", "i4"),
                                           ("flip", "i4"), ("dtype", "i4")])
            for i, (_, p, buf, kind, shp) in enumerate(ents):
                d[i] = (p.data_ptr(), buf.data_ptr(), shp[0], shp[1], shp[2], shp[3], int(kind[1]), ops._DT[kind[0]])
            table = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(ents[0][1].device)
            hit = self._mp[id(opt)] = (sig, table)
        N.call("mv_dconv_multi_pack", _P(hit[1]), len(ents), ops._stream())
        for key, p, buf, _, _ in ents:
            self.d[key] = (weakref.ref(p), (p._version, ops.param_epoch_of(p), p.data_ptr()), buf)
        return len(ents)

    def get(self, param, dtype, flip):
        key, ver, buf = self._lookup(param, (dtype, flip))
        if buf is not None:
            return buf
        w = _as4d(param)
        Cout, Cin, kh, kw = w.shape
        buf = torch.empty(N.lib().mv_dconv_packed_bytes(Cout, Cin, kh, kw, ops._DT[dtype]), dtype=torch.uint8, device=w.device)
        wd = w.detach().contiguous()
        N.call("mv_dconv_pack", _P(wd), ops._DT[wd.dtype], _P(buf), Cout, Cin, kh, kw, int(flip), ops._DT[dtype], ops._stream())
        return self._store(key, param, ver, buf)

    def head_padded(self, wparam, bparam, dtype, flip, pad_to=32):
        """The [1,C,kh,kw] head zero-padded to `pad_to` output channels, packed for the MFMA conv (forward: flip 0, with the
        padded bias; data gradient: flip 1)."""
        key, ver, hit = self._lookup(wparam, ("headpad", dtype, flip))
        if hit is not None:
            return hit
        w = _as4d(wparam).detach()
        C, kh, kw = w.shape[1], w.shape[2], w.shape[3]
        wp = torch.zeros(pad_to, C, kh, kw, device=w.device, dtype=w.dtype)   # parameter-sized host-side glue
        wp[0] = w[0]
        buf = torch.empty(N.lib().mv_dconv_packed_bytes(pad_to, C, kh, kw, ops._DT[dtype]), dtype=torch.uint8, device=w.device)
        N.call("mv_dconv_pack", _P(wp), ops._DT[wp.dtype], _P(buf), pad_to, C, kh, kw, int(flip), ops._DT[dtype], ops._stream())
        bias = torch.zeros(pad_to, device=w.device, dtype=dtype)
        bias[0:1] = bparam.detach().to(dtype)
        return self._store(key, wparam, ver, (buf, bias))

    def head_mfma(self, param, dtype):
        """[1,C,kh,kw] head -> the tap-row MFMA operators of mv_dhead_fwd / mv_dhead_dgrad."""
        key, ver, buf = self._lookup(param, ("headmfma", dtype))
        if buf is not None:
            return buf
        w = _as4d(param)
        C, kh, kw = w.shape[1], w.shape[2], w.shape[3]
        buf = torch.empty(N.lib().mv_dhead_packed_bytes(C, ops._DT[dtype]), dtype=torch.uint8, device=w.device)
        wd = w.detach().contiguous()
        N.call("mv_dhead_pack", _P(wd), ops._DT[wd.dtype], _P(buf), C, kh, kw, ops._DT[dtype], ops._stream())
        return self._store(key, param, ver, buf)

    def head(self, param):
        """[1,C,kh,kw] (or [1,C,k]) -> fp32 [taps][C]."""
        key, ver, buf = self._lookup(param, "head")
        if buf is not None:
            return buf
        w = _as4d(param)
        C, taps = w.shape[1], w.shape[2] * w.shape[3]
        wt = torch.empty(taps, C, device=w.device, dtype=torch.float32)
        wd = w.detach().contiguous()
        N.call("mv_conv_out_pack", _P(wd), ops._DT[wd.dtype], _P(wt), C, taps, ops._stream())
        return self._store(key, param, ver, wt)


_packs = _PackCache()


def _as4d(w):
    return w if w.dim() == 4 else w.unsqueeze(2)        # Conv1d [O,C,k] -> [O,C,1,k]


class _DiscStack(Function):
    @staticmethod
    def forward(ctx, x0, slope, *params):
        """x0 [B,1,H,W] (MPD fold) or [B,1,W] (MSD, pooled); params = w1,b1,...,w5,b5."""
        ws = [_as4d(p) for p in params[0::2]]
        bs = list(params[1::2])
        dt = x0.dtype
        x0 = x0.contiguous()
        B = x0.shape[0]
        H, W = (x0.shape[2], x0.shape[3]) if x0.dim() == 4 else (1, x0.shape[2])
        kh, kw = ws[0].shape[2], ws[0].shape[3]
        st, dev = ops._stream, x0.device
        cast = lambda t: _cache.get(t, dt)
        acts = []
        C1 = ws[0].shape[0]
        a = torch.empty(B, H, W, C1, device=dev, dtype=dt)
        N.call("mv_dfirst_fwd_cl", _P(x0), _P(cast(params[0])), _P(cast(bs[0])), _P(a), B, H, W, C1, kh, kw, float(slope), ops._dt(x0), st())
        acts.append(a)
        C4 = ws[4].shape[1]
        hkh, hkw = ws[4].shape[2], ws[4].shape[3]
        mfma_head = C4 == 256 and hkh * hkw <= 16
        zws, head_fused = None, False
        for li in (1, 2, 3):
            Cout, Cin = ws[li].shape[0], ws[li].shape[1]
            y = torch.empty(B, H, W, Cout, device=dev, dtype=dt)
            if li == 3 and mfma_head and Cout == 256 and dt != torch.float32:
                # the head's per-tap partial sums come out of the last conv's epilogue tile (LDS): its 256-channel output is not read back
                zws = torch.empty(16, B * H * W, device=dev, dtype=torch.float32)
                rc = N.lib().mv_dconv_cl_fwd_head(_P(acts[-1]), _P(_packs.get(params[2 * li], dt, 0)), _P(cast(bs[li])), _P(y),
                                                  _P(_packs.head_mfma(params[8], dt)), _P(zws), hkh, hkw, B, H, W, Cin, Cout, kh, kw,
                                                  N.ACT_LRELU, float(slope), ops._dt(x0), st())
                if rc != -3:            # MV_ERR_UNSUPPORTED: no 256-row variant for this geometry - separate launches below
                    N.check(rc, "mv_dconv_cl_fwd_head")
                    head_fused = True
            if not (li == 3 and head_fused):
                N.call("mv_dconv_cl_fwd", _P(acts[-1]), _P(_packs.get(params[2 * li], dt, 0)), _P(cast(bs[li])), None, _P(y), B, H, W, Cin, Cout,
                       kh, kw, 1, N.ACT_LRELU, float(slope), ops._dt(x0), st())
            acts.append(y)
        out = torch.empty((B, 1, H, W) if x0.dim() == 4 else (B, 1, W), device=dev, dtype=dt)
        if head_fused:
            N.call("mv_dhead_sum", _P(zws), _P(cast(bs[4])), _P(out), B, H, W, hkh, hkw, ops._dt(x0), st())
        elif mfma_head:
            # head (C4 -> 1): taps on the MFMA rows, x read once (mv_dhead_fwd)
            zws = torch.empty(16, B * H * W, device=dev, dtype=torch.float32) if zws is None else zws
            N.call("mv_dhead_fwd", _P(acts[-1]), _P(_packs.head_mfma(params[8], dt)), _P(cast(bs[4])), _P(zws), _P(out), B, H, W, C4,
                   kh, kw, ops._dt(x0), st())
        else:
            # generic width: MFMA conv with the output zero-padded to 32 channels, then channel 0 is extracted
            hp, hb = _packs.head_padded(params[8], bs[4], dt, 0)
            y32 = torch.empty(B, H, W, 32, device=dev, dtype=dt)
            N.call("mv_dconv_cl_fwd", _P(acts[-1]), _P(hp), _P(hb), None, _P(y32), B, H, W, C4, 32, kh, kw, 1, N.ACT_NONE, float(slope),
                   ops._dt(x0), st())
            N.call("mv_take_channel", _P(y32), _P(out), B * H * W, 32, 0, ops._dt(x0), st())
        ctx.geom = (B, H, W, kh, kw, float(slope), x0.dim())
        ctx.save_for_backward(x0, *acts, *params)
        return out

    @staticmethod
    def backward(ctx, gy):
        B, H, W, kh, kw, slope, xdim = ctx.geom
        saved = ctx.saved_tensors
        x0, acts, params = saved[0], list(saved[1:5]), list(saved[5:])
        ws = [_as4d(p) for p in params[0::2]]
        bs = list(params[1::2])
        dt, dev, st = x0.dtype, x0.device, ops._stream
        gy = gy.contiguous()
        need = ctx.needs_input_grad
        need_w = any(need[2:])
        grads = [None] * 10
        f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        to = lambda g, p: (g.view(p.shape) if g.dtype == p.dtype else ops.cast(g.view(p.shape).contiguous(), p.dtype))
        # ---- head
        C4 = ws[4].shape[1]
        if need_w:
            taps = kh * kw
            if taps <= 16 and C4 % 8 == 0:
                # gw[tap][c] = sum_pos G[pos][tap] x[pos][c] with G = the flipped tap matrix of gy: a 1x1 weight-gradient GEMM
                # on the MFMA kernel; the centre column of G sums to the bias gradient
                tm = torch.empty(B, H, W, 16, device=dev, dtype=dt)
                N.call("mv_tap_matrix", _P(gy), _P(tm), B, H, W, kh, kw, 1, ops._dt(gy), st())
                gw16, gb16 = ops.dconv_wgrad_cl(acts[3], tm, 1, 1, want_bias=True)
                gwt = gw16.view(16, C4)[:taps]
                gb5 = gb16[taps // 2:taps // 2 + 1]
            else:
                gwt, gb5 = f32(taps, C4), f32(1)
                N.call("mv_dhead_wgrad", _P(gy), _P(acts[3]), _P(gwt), _P(gb5), B, H, W, C4, kh, kw, ops._dt(gy), st())
            gw5 = f32(C4, taps)                          # [taps][C] -> [C][taps] with our own transpose kernel
            N.call("mv_ntc_to_nct", _P(gwt), _P(gw5), 1, C4, taps, N.MV_F32, st())
            grads[8], grads[9] = to(gw5, params[8]), to(gb5, params[9])
        g = torch.empty(B, H, W, C4, device=dev, dtype=dt)
        if C4 == 256 and kh * kw <= 16:
            N.call("mv_dhead_dgrad", _P(gy), _P(_packs.head_mfma(params[8], dt)), _P(acts[3]), _P(g), B, H, W, C4, kh, kw, slope,
                   ops._dt(gy), st())
        else:
            g32 = torch.empty(B, H, W, 32, device=dev, dtype=dt)
            N.call("mv_put_channel", _P(gy), _P(g32), B * H * W, 32, 0, ops._dt(gy), st())
            hpf, _ = _packs.head_padded(params[8], bs[4], dt, 1)
            N.call("mv_dconv_cl_fwd", _P(g32), _P(hpf), None, _P(acts[3]), _P(g), B, H, W, 32, C4, kh, kw, 1, N.ACT_NONE, slope,
                   ops._dt(gy), st())
        # ---- wide layers 4, 3, 2 (indices 3, 2, 1)
        for li in (3, 2, 1):
            Cout, Cin = ws[li].shape[0], ws[li].shape[1]
            if need_w:
                gw, gb = f32(Cout, Cin, kh, kw), f32(Cout)
                ops.wgrad_cl_into(acts[li - 1], g, gw, gb, B, H, W, Cin, Cout, kh, kw, 1)
                grads[2 * li], grads[2 * li + 1] = to(gw, params[2 * li]), to(gb, params[2 * li + 1])
            gprev = torch.empty(B, H, W, Cin, device=dev, dtype=dt)
            N.call("mv_dconv_cl_fwd", _P(g), _P(_packs.get(params[2 * li], dt, 1)), None, _P(acts[li - 1]), _P(gprev), B, H, W, Cout, Cin,
                   kh, kw, 1, N.ACT_NONE, slope, ops._dt(g), st())
            g = gprev
        # ---- first layer
        C1 = ws[0].shape[0]
        if need_w:
            taps = kh * kw
            if taps <= 16 and C1 % 8 == 0:
                # gw[o][tap] = sum_pos g1[pos][o] X[pos][tap] with X = the tap matrix of x0 (1x1 weight-gradient GEMM on MFMA)
                tm = torch.empty(B, H, W, 16, device=dev, dtype=dt)
                N.call("mv_tap_matrix", _P(x0), _P(tm), B, H, W, kh, kw, 0, ops._dt(g), st())
                gw16, gb1 = ops.dconv_wgrad_cl(tm, g, 1, 1, want_bias=True)
                gw1 = f32(C1, taps)
                ops.copy_rows(gw16.view(1, C1, 16), gw1.view(1, C1, taps))
            else:
                gw1, gb1 = f32(C1, taps), f32(C1)
                N.call("mv_dfirst_wgrad_cl", _P(g), _P(x0), _P(gw1), _P(gb1), B, H, W, C1, kh, kw, ops._dt(g), st())
            grads[0], grads[1] = to(gw1, params[0]), to(gb1, params[1])
        gx0 = None
        if need[0]:
            gx0 = torch.empty_like(x0)
            ws = torch.empty(N.lib().mv_dfirst_dgrad_workspace_bytes(B, H, W) // 4, device=dev, dtype=torch.float32)
            N.call("mv_dfirst_dgrad_cl", _P(g), _P(_cache.get(params[0], dt)), _P(gx0), _P(ws), B, H, W, C1, kh, kw, ops._dt(g), st())
        return (gx0, None, *grads)


_cache = ops._ParamCache()


def supported(x, blk) -> bool:
    return x.dtype in (torch.bfloat16, torch.float16)


def disc_stack(x0, blk, slope=0.1):
    convs = [blk.conv_layers[i] for i in (0, 2, 4, 6, 8)]
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    from . import functional as Fn
    if Fn.DISPATCH == "torch_ops":
        return Fn._t().OPS.disc_conv_stack(x0, float(slope), *params)
    return _DiscStack.apply(x0, slope, *params)
