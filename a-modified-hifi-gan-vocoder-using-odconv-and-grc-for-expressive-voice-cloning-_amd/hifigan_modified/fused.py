"""Channels-last ("NTC") fused fast path: MFMA kernels that keep the residual stream in HBM as
[B, T, C] between launches.  The public modules stay NCT; the generator converts once at its
boundary (mel in, waveform out - a [B,1,T] waveform is the same memory in both layouts).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_void_p

import torch

from . import _native as N
from . import ops
from .ops import _zeros, _new_zeros

_INT3 = ctypes.c_int * 3


def _versions(params):
    return (ops.param_epoch(),) + tuple((id(p), p._version, p.data_ptr()) for p in params)


class MrfFused:
    """Packed-weight cache + launcher of mv_mrf_block_fwd_cl for one MultiReceptiveFieldBlock."""

    def __init__(self, blk):
        self.blk = blk
        self._packed = {}     # dtype -> (versions, tensor)
        self._ws = {}

    @staticmethod
    def supported(blk) -> bool:
        if blk.in_channels != 64 or blk.out_channels != 64 or len(blk.dilations) != 3:
            return False
        if blk.channels_per_dilation != 20 or blk.norm_groups != 8:
            return False
        for g in blk.conv_layers:
            if (g.kernel_size != 3 or g.groups != 4 or g.norm_groups != 5 or g.in_channels != 64
                    or not hasattr(g, "residual_proj")):
                return False
        offs = {0}
        for d in blk.dilations:
            if not 1 <= d <= 8:
                return False
            offs |= {d, -d}
        return len(offs) <= 7

    def _params(self):
        b = self.blk
        ps = []
        for g in b.conv_layers:
            ps += [g.conv.weight, g.conv.bias, g.lora_A, g.lora_B, g.lora_scaling, g.output_projection.weight,
                   g.output_projection.bias, g.norm.weight, g.norm.bias, g.residual_proj.weight, g.residual_proj.bias]
        ps += [b.fusion.weight, b.fusion.bias, b.norm.weight, b.norm.bias]
        return ps

    def packed(self, dtype, device):
        ps = self._params()
        ver = _versions(ps)
        hit = self._packed.get(dtype)
        if hit is not None and hit[0] == ver and hit[1].device == device:
            return hit[1]
        pd = ps[0].dtype
        if any(p.dtype != pd or not p.is_contiguous() or p.device != device for p in ps):
            raise RuntimeError("MRF parameters must share one dtype, be contiguous and live on the input's device")
        st = N.MrfParams()
        names = ("conv_w", "conv_b", "lora_A", "lora_B", "lora_scaling", "proj_w", "proj_b", "norm_w", "norm_b",
                 "res_w", "res_b")
        for i in range(3):
            for j, nme in enumerate(names):
                getattr(st, nme)[i] = ps[i * 11 + j].data_ptr()
        st.fusion_w, st.fusion_b, st.norm2_w, st.norm2_b = (p.data_ptr() for p in ps[33:37])
        nbytes = N.lib().mv_mrf_packed_bytes(ops._DT[dtype])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dil = _INT3(*self.blk.dilations)
        rank = self.blk.conv_layers[0].lora_A.shape[1]
        N.call("mv_mrf_pack", ctypes.byref(st), ops._DT[pd], dil, rank, c_void_p(buf.data_ptr()), ops._DT[dtype],
               ops._stream())
        self._packed[dtype] = (ver, buf)
        return buf

    def forward_cl(self, x_cl, mask=None, mask_scale=1.0):
        """x_cl [B, T, 64] contiguous -> new tensor [B, T, 64]."""
        B, T, C = x_cl.shape
        assert C == 64 and x_cl.is_contiguous()
        dt = ops._dt(x_cl)
        packed = self.packed(x_cl.dtype, x_cl.device)
        wsb = N.lib().mv_mrf_workspace_bytes(B, T, dt)
        key = (wsb, x_cl.device)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(wsb, dtype=torch.uint8, device=x_cl.device)
            self._ws[key] = ws                     # never evicted: a captured HIP graph may have baked this pointer in
        out = torch.empty_like(x_cl)
        dil = _INT3(*self.blk.dilations)
        N.call("mv_mrf_block_fwd_cl", c_void_p(x_cl.data_ptr()), c_void_p(out.data_ptr()), c_void_p(packed.data_ptr()),
               dil, c_void_p(ws.data_ptr()), None if mask is None else c_void_p(mask.data_ptr()), float(mask_scale),
               B, T, float(self.blk.norm.eps), dt, ops._stream())
        return out


class MrfChain:
    """mv_mrf_chain_fwd_cl over consecutive fused MultiReceptiveFieldBlocks (the generator's `for blk in mrf_blocks`)."""

    def __init__(self, mrfs):
        self.mrfs = list(mrfs)
        self._ws = {}

    def _mode(self, x_cl, w16):
        """(mv_dtype code, dtype of the packed weight image): w16 = fp32 storage with single-f16 weights (MV_F32_W16)."""
        if w16:
            assert x_cl.dtype == torch.float32
            return N.MV_F32_W16, torch.float16
        return ops._dt(x_cl), x_cl.dtype

    def forward_cl(self, x_cl, nblocks=None, w16=False, materialize=True, x_pair=False):
        """x_cl [B, T, 64] contiguous -> output of block nblocks-1 (default: the last), a new tensor.  materialize=False: the
        chain's passes only (what the generator runs in front of its fused output conv) - returns None; bench.py times that."""
        n = len(self.mrfs) if nblocks is None else nblocks
        B, T, C = x_cl.shape
        assert C == 64 and x_cl.is_contiguous() and 1 <= n <= len(self.mrfs)
        dt, pdt = self._mode(x_cl, w16)
        packs = [m.packed(pdt, x_cl.device) for m in self.mrfs[:n]]
        ptrs = (c_void_p * n)(*[p.data_ptr() for p in packs])
        dil = (ctypes.c_int * (3 * n))(*[d for m in self.mrfs[:n] for d in m.blk.dilations])
        wsb = N.lib().mv_mrf_chain_workspace_bytes(B, T, dt)
        key = (wsb, x_cl.device)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(wsb + 256, dtype=torch.uint8, device=x_cl.device)
            self._ws[key] = ws                     # never evicted: a captured HIP graph may have baked this pointer in
        base = (ws.data_ptr() + 255) // 256 * 256
        out = torch.empty_like(x_cl) if materialize else None
        eps = float(self.mrfs[0].blk.norm.eps)
        assert not x_pair or w16
        N.call("mv_mrf_chain_fwd_cl", c_void_p(x_cl.data_ptr()), None if out is None else c_void_p(out.data_ptr()), ptrs, dil, n,
               c_void_p(base), B, T, eps, N.MV_F32_W16P if x_pair else dt, ops._stream())
        return out

    def forward_out_cl(self, x_cl, conv_packed, conv_bias, ks, act, w16=False, x_pair=False):
        """Chain + output projection + activation in one call (mv_mrf_chain_out_fwd_cl): x_cl [B, T, 64] -> wave [B, 1, T].
        x_pair: x_cl holds pair rows (OdconvFused.forward_cl(out_pair=True)); w16 only (MV_F32_W16P)."""
        n = len(self.mrfs)
        B, T, C = x_cl.shape
        assert C == 64 and x_cl.is_contiguous()
        dt, pdt = self._mode(x_cl, w16)
        packs = [m.packed(pdt, x_cl.device) for m in self.mrfs]
        ptrs = (c_void_p * n)(*[p.data_ptr() for p in packs])
        dil = (ctypes.c_int * (3 * n))(*[d for m in self.mrfs for d in m.blk.dilations])
        wsb = N.lib().mv_mrf_chain_workspace_bytes(B, T, dt)
        key = (wsb, x_cl.device)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(wsb + 256, dtype=torch.uint8, device=x_cl.device)
            self._ws[key] = ws                     # never evicted: a captured HIP graph may have baked this pointer in
        base = (ws.data_ptr() + 255) // 256 * 256
        wave = torch.empty(B, 1, T, device=x_cl.device, dtype=x_cl.dtype)
        assert not x_pair or w16
        N.call("mv_mrf_chain_out_fwd_cl", c_void_p(x_cl.data_ptr()), c_void_p(wave.data_ptr()), ptrs, dil, n, c_void_p(base),
               c_void_p(conv_packed.data_ptr()), float(conv_bias), int(ks), int(act), B, T, float(self.mrfs[0].blk.norm.eps),
               N.MV_F32_W16P if x_pair else dt, ops._stream())
        return wave


def mrf_fused_for(blk):
    f = getattr(blk, "_mv_fused", None)
    if f is None:
        f = MrfFused(blk) if MrfFused.supported(blk) else False
        object.__setattr__(blk, "_mv_fused", f)
    return f or None


class OdconvFused:
    """Packed-bank cache + launcher of mv_odconv_cl_fwd for one ODConv1d / ODConvTranspose1d."""

    def __init__(self, mod):
        self.mod = mod
        self.transposed = bool(getattr(mod, "_transposed", False))
        self._packed = {}

    def geometry(self):
        m = self.mod
        return (m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, int(self.transposed), m.K)

    def supported(self) -> bool:
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        if cin % 8 or cout % 8 or K > 8:
            return False
        rows = stride * cout if tr else cout
        if rows % 16:
            return False
        if tr:
            return dil == 1 and ks % stride == 0
        return stride == 1

    def packed(self, dtype, device):
        w = self.mod.kernels
        ver = (w._version, w.data_ptr(), ops.param_epoch())
        hit = self._packed.get(dtype)
        if hit is not None and hit[0] == ver and hit[1].device == device:
            return hit[1]
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        nbytes = N.lib().mv_odconv_cl_packed_bytes(cin, cout, ks, stride, tr, K, ops._DT[dtype])
        if nbytes == 0:
            raise RuntimeError("odconv_cl: unsupported geometry")
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        wd = w.detach()
        wd = wd if wd.is_contiguous() else wd.contiguous()
        N.call("mv_odconv_cl_pack", c_void_p(wd.data_ptr()), ops._DT[wd.dtype], c_void_p(buf.data_ptr()), cin, cout, ks,
               stride, tr, K, ops._DT[dtype], ops._stream())
        self._packed[dtype] = (ver, buf)
        return buf

    def dgrad_supported(self) -> bool:
        """Data gradient on the fused kernel: transposed layers with kernel_size = 2*stride (every upsampler of the generator)."""
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        return bool(tr) and dil == 1 and ks == 2 * stride and self.mod.output_padding <= pad and cin % 16 == 0 and \
            (stride * cout) % 8 == 0 and K <= 8

    def packed_dgrad(self, dtype, device):
        """The adjoint operator (2-tap stride-1 ODConv over rows of `stride` output steps), packed for mv_odconv_cl_fwd."""
        w = self.mod.kernels
        ver = (w._version, w.data_ptr(), ops.param_epoch())
        hit = self._packed.get(("dgrad", dtype))
        if hit is not None and hit[0] == ver and hit[1].device == device:
            return hit[1]
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        wd = w.detach()
        wd = wd if wd.is_contiguous() else wd.contiguous()
        adj = torch.empty(K, cin, stride * cout, 2, device=device, dtype=wd.dtype)
        P = lambda t: c_void_p(t.data_ptr())
        N.call("mv_odconvT_adjoint_weights", P(wd), P(adj), K, cin, cout, ks, stride, ops._DT[wd.dtype], ops._stream())
        nbytes = N.lib().mv_odconv_cl_packed_bytes(stride * cout, cin, 2, 1, 0, K, ops._DT[dtype])
        if nbytes == 0:
            raise RuntimeError("odconv_cl: unsupported adjoint geometry")
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        N.call("mv_odconv_cl_pack", P(adj), ops._DT[adj.dtype], P(buf), stride * cout, cin, 2, 1, 0, K, ops._DT[dtype], ops._stream())
        self._packed[("dgrad", dtype)] = (ver, buf)
        return buf

    def pad_grad(self, g, Tin):
        """g [B,Cout,Tout] (NCT) -> the time-padded channels-last gradient [B][(Tin+1)*stride][Cout]: row q*stride + r holds
        g[:, :, q*stride + r - pad] (zeros outside), i.e. viewed as [B][Tin+1][stride*Cout] it is the input of the adjoint ops."""
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        B, _, Tout = g.shape
        rows = (Tin + 1) * stride
        gp = _zeros(B, rows, cout, device=g.device, dtype=g.dtype)
        N.call("mv_nct_to_ntc_window", c_void_p(g.data_ptr()), c_void_p(gp.data_ptr() + pad * cout * g.element_size()), B, cout, Tout,
               rows * cout, ops._dt(g), ops._stream())
        return gp

    def dgrad(self, gp, alpha, Tin):
        """padded gradient -> gx [B,Cin,Tin] (NCT): the alpha-aggregated adjoint conv on the MFMA kernel (None: tile too wide)."""
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        B = gp.shape[0]
        P = lambda t: None if t is None else c_void_p(t.data_ptr())
        gx_cl = torch.empty(B, Tin, cin, device=gp.device, dtype=gp.dtype)
        rc = N.lib().mv_odconv_cl_fwd(P(gp), P(self.packed_dgrad(gp.dtype, gp.device)), None, P(alpha), None, 0, None, None, None, 0,
                                      P(gx_cl), None, B, stride * cout, Tin + 1, cin, Tin, 2, 1, 0, 1, 0, K, int(N.ACT_NONE), 0.1,
                                      ops._dt(gp), ops._stream())
        if rc == -3:                         # MV_ERR_UNSUPPORTED: tile does not fit LDS (very wide adjoint input): the caller uses the generic HIP kernel
            return None
        N.check(rc, "mv_odconv_cl_fwd")
        return ops.ntc_to_nct(gx_cl)

    def wgrad(self, x_cl, gp, wk, alpha):
        """Bank gradients gw fp32 [K,Cin,Cout,ks] and d alpha fp32 [B,K] on MFMA (per-sample two-tap weight-gradient GEMMs +
        the alpha-chain reduction); None when the shape is outside the kernel's envelope."""
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        B, Tin, _ = x_cl.shape
        P = lambda t: c_void_p(t.data_ptr())
        gw = torch.empty(K, cin, cout, ks, device=x_cl.device, dtype=torch.float32)
        galpha = _zeros(B, K, device=x_cl.device, dtype=torch.float32)
        ws = torch.empty(N.lib().mv_odconvT_wgrad_workspace_bytes(B, cin, cout, ks, K, ops._dt(x_cl)), dtype=torch.uint8, device=x_cl.device)
        rc = N.lib().mv_odconvT_wgrad_mfma(P(x_cl), P(gp), P(wk), P(alpha), P(gw), P(galpha), P(ws), B, Tin, cin, cout, ks, stride, K,
                                           ops._dt(x_cl), ops._stream())
        if rc == -3:
            return None
        N.check(rc, "mv_odconvT_wgrad_mfma")
        return gw, galpha

    def out_len(self, Tin):
        m = self.mod
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        return (Tin - 1) * stride - 2 * pad + ks + m.output_padding if tr else Tin + 2 * pad - dil * (ks - 1)

    def pool_floats(self, B, Tin, dtype, act=N.ACT_NONE, has_film=False, in_f16=False):
        """floats per sample of the partial-sum buffer (`pooled_out`) this layer's launch writes for its consumer's attention.
        in_f16: the launch is the mixed mode's first fp32 stage reading the fp16 stream (mv_odconv_cl_fwd_in16)."""
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        n = N.lib().mv_odconv_cl_pool_floats_in(B, cin, Tin, cout, self.out_len(Tin), ks, stride, pad, dil, int(tr), K, int(act),
                                                int(has_film), ops._DT[dtype], int(in_f16))
        if n == 0 and in_f16:                # no input-widening variant for this geometry: forward_cl casts and takes the plain entry
            n = N.lib().mv_odconv_cl_pool_floats_in(B, cin, Tin, cout, self.out_len(Tin), ks, stride, pad, dil, int(tr), K, int(act),
                                                    int(has_film), ops._DT[dtype], 0)
        if n == 0:
            raise RuntimeError("odconv_cl: unsupported geometry")
        return n

    def forward_cl(self, x_cl, cache, alpha=None, pooled_in=None, film_proj=None, film_F=0, pooled_out=None,
                   act=N.ACT_NONE, slope=0.1, storage=None, out_pair=False):
        """pooled_in / pooled_out: fp32 [B, n] partial channel sums (n = the producer's / this layer's pool_floats).
        storage = torch.float32 with an fp16 x_cl: the first fp32 stage of the mixed mode - the kernel widens its input while
        staging it (mv_odconv_cl_fwd_in16) where that variant exists, else x_cl is cast first.
        out_pair=True: returns (y, is_pair) - where the geometry has the variant (mv_odconv_cl_fwd_pair: fp32 storage, 64 output
        channels, streaming kernel) y holds the MRF chain's pair rows instead of fp32 values (same shape and byte size)."""
        m = self.mod
        cin, cout, ks, stride, pad, dil, tr, K = self.geometry()
        B, Tin, C = x_cl.shape
        assert C == cin and x_cl.is_contiguous()
        Tout = self.out_len(Tin)
        att = m.kernel_attention[1]
        P = lambda t: None if t is None else c_void_p(t.data_ptr())
        if storage is not None and storage != x_cl.dtype:
            if storage == torch.float32 and x_cl.dtype == torch.float16 and film_proj is None and not out_pair:
                y = torch.empty(B, Tout, cout, device=x_cl.device, dtype=storage)
                rc = N.lib().mv_odconv_cl_fwd_in16(P(x_cl), P(self.packed(storage, x_cl.device)), P(cache.get(m.bias, storage)), P(alpha),
                                                   P(pooled_in), 0 if pooled_in is None else pooled_in.shape[1],
                                                   P(cache.get(att.weight, storage)), P(cache.get(att.bias, storage)), P(y), P(pooled_out),
                                                   B, cin, Tin, cout, Tout, ks, stride, pad, dil, int(tr), K, int(act), float(slope),
                                                   ops._stream())
                if rc != -3:                                   # MV_ERR_UNSUPPORTED: no input-widening variant for this geometry
                    N.check(rc, "mv_odconv_cl_fwd_in16")
                    return y
            x_cl = ops.cast(x_cl, storage)
        y = torch.empty(B, Tout, cout, device=x_cl.device, dtype=x_cl.dtype)
        if out_pair:
            if x_cl.dtype == torch.float32 and film_proj is None:
                rc = N.lib().mv_odconv_cl_fwd_pair(P(x_cl), P(self.packed(x_cl.dtype, x_cl.device)), P(cache.get(m.bias, x_cl.dtype)), P(alpha),
                                                   P(pooled_in), 0 if pooled_in is None else pooled_in.shape[1],
                                                   P(cache.get(att.weight, x_cl.dtype)), P(cache.get(att.bias, x_cl.dtype)), P(y), P(pooled_out),
                                                   B, cin, Tin, cout, Tout, ks, stride, pad, dil, int(tr), K, int(act), float(slope),
                                                   ops._stream())
                if rc != -3:                                   # MV_ERR_UNSUPPORTED: no pair-row variant for this geometry
                    N.check(rc, "mv_odconv_cl_fwd_pair")
                    return y, True
            return self.forward_cl(x_cl, cache, alpha, pooled_in, film_proj, film_F, pooled_out, act, slope), False
        N.call("mv_odconv_cl_fwd", P(x_cl), P(self.packed(x_cl.dtype, x_cl.device)), P(cache.get(m.bias, x_cl.dtype)),
               P(alpha), P(pooled_in), 0 if pooled_in is None else pooled_in.shape[1],
               P(cache.get(att.weight, x_cl.dtype)), P(cache.get(att.bias, x_cl.dtype)),
               P(film_proj), int(film_F), P(y), P(pooled_out), B, cin, Tin, cout, Tout, ks, stride, pad, dil, int(tr), K,
               int(act), float(slope), ops._dt(x_cl), ops._stream())
        return y


_GEN_PAIR = os.environ.get("MV_GEN_PAIR", "0") == "1"


class GeneratorFused:
    """The whole generator in channels-last layout: 1 transpose + 1 attention launch for the mel input, then
    input_proj(+FiLM) -> 4x ODConvT(+LeakyReLU, +pooling for the next layer) -> 3x fused MRF -> out conv + tanh."""

    def __init__(self, gen):
        self.gen = gen
        self.inp = OdconvFused(gen.input_proj)
        self.ups = [OdconvFused(l[0]) for l in gen.upsample_layers]
        self.mrfs = [mrf_fused_for(b) for b in gen.mrf_blocks]
        eps = {b.norm.eps for b in gen.mrf_blocks} | {g.norm.eps for b in gen.mrf_blocks for g in b.conv_layers}
        self.chain = MrfChain(self.mrfs) if (self.mrfs and all(m is not None for m in self.mrfs) and len(eps) == 1) else None
        self._wt = {}

    def supported(self) -> bool:
        g = self.gen
        ok = self.inp.supported() and all(u.supported() for u in self.ups) and all(m is not None for m in self.mrfs)
        k = g.output_proj.kernel_size[0]
        return bool(ok and g.output_proj.in_channels == 64 and k % 2 == 1 and g.output_proj.padding[0] == k // 2)

    def out_weights(self, device):
        w, b = self.gen.output_proj.weight, self.gen.output_proj.bias
        ver = (w._version, w.data_ptr(), b._version, ops.param_epoch())
        hit = self._wt.get("w")
        if hit is not None and hit[0] == ver and hit[1].device == device:
            return hit[1], hit[2]
        C, ks = w.shape[1], w.shape[2]
        wt = torch.empty(N.lib().mv_conv_out_packed_bytes(C, ks), device=device, dtype=torch.uint8)   # fp32 | bf16 | f16 images
        wd = w.detach().contiguous()
        N.call("mv_conv_out_pack_all", c_void_p(wd.data_ptr()), ops._DT[wd.dtype], c_void_p(wt.data_ptr()), C, ks, ops._stream())
        bias = float(b.detach().float().item())   # host read of one scalar, once per weight version
        self._wt["w"] = (ver, wt, bias)
        return wt, bias

    def forward(self, mel, speaker_emb=None, emotion_emb=None, cache=None, return_stages=False):
        g = self.gen
        dt = mel.dtype
        st = {}
        B = mel.shape[0]
        # Mixed storage (ModifiedHiFiGANGenerator.set_mixed_precision; fp32-storage models only): the prologue, input_proj and the
        # first `n_early` upsamplers run in `edt` (fp16) storage, the stream is cast to fp32 once, and everything behind runs with
        # split operands.  tools/error_budget.py: fp16 through up1 costs 5.9e-4 of the 1e-3 waveform budget at 22 kHz.
        mixed = getattr(g, "_mv_mixed", None)
        n_early, edt = (mixed if (mixed is not None and dt == torch.float32) else (-1, dt))
        sdt = lambda j: edt if j <= n_early else dt            # storage type of producer j (0 = input_proj, j = ups[j-1])
        mel = mel if mel.is_contiguous() else mel.contiguous()
        att = g.input_proj.kernel_attention[1]
        K0, C0 = att.weight.shape[0], att.weight.shape[1]
        # partial channel sums handed from each producer to its consumer (ODConv attention pooling, odconv.py:36-40,85): views of ONE
        # buffer, each sized by its PRODUCER (slots x GEMM rows per sample); every element is written exactly once, nothing is
        # accumulated with atomics, so two runs agree bit for bit
        has_cond = speaker_emb is not None or emotion_emb is not None
        prods = [self.inp] + self.ups[:-1]
        lens, T_ = [], mel.shape[2]
        for j, pr in enumerate(prods):
            lens.append(pr.pool_floats(B, T_, sdt(j), N.ACT_NONE if j == 0 else N.ACT_LRELU, has_film=(j == 0 and has_cond),
                                       in_f16=(j >= 1 and sdt(j) == torch.float32 and sdt(j - 1) == torch.float16)))
            T_ = pr.out_len(T_)
        nflat = sum(B * n for n in lens)
        P = lambda t: None if t is None else c_void_p(t.data_ptr())
        fp = g.final_film.condition_projection
        F = g.final_film.feature_dim if has_cond else 0
        # one launch: attention of input_proj, the channels-last copy of the mel, the FiLM projection, the zero fill
        d0 = sdt(0)
        flat = torch.empty(nflat, device=mel.device, dtype=torch.float32)
        alpha0 = torch.empty(B, K0, device=mel.device, dtype=torch.float32)
        x = torch.empty(B, mel.shape[2], mel.shape[1], device=mel.device, dtype=d0)
        film_proj = torch.empty(B, 2 * F, device=mel.device, dtype=d0) if has_cond else None
        # the prologue reads mel / spk / emo in the caller's type (mixed mode: fp32 inputs, fp16 outputs - no cast launches)
        idt = mel.dtype
        spk = None if speaker_emb is None else ops.cast(speaker_emb, idt).contiguous()
        emo = None if emotion_emb is None else ops.cast(emotion_emb, idt).contiguous()
        rc = N.lib().mv_gen_prologue_in(P(mel), P(cache.get(att.weight, d0)), P(cache.get(att.bias, d0)), P(spk), P(emo),
                                        P(cache.get(fp.weight, d0)) if has_cond else None, P(cache.get(fp.bias, d0)) if has_cond else None,
                                        P(alpha0), P(x), P(film_proj), None, 0, B, mel.shape[1], mel.shape[2], K0,
                                        0 if spk is None else spk.shape[1], 0 if emo is None else emo.shape[1], fp.in_features, 2 * F,
                                        ops._DT[idt], ops._DT[d0], ops._stream())
        if rc == -3:    # MV_ERR_UNSUPPORTED: a long utterance does not fit one workgroup's LDS - separate launches
            mel = ops.cast(mel, d0)
            alpha0 = ops.odconv_attn(mel, cache.get(att.weight, d0).view(K0, C0), cache.get(att.bias, d0))
            x = ops.nct_to_ntc(mel)
            film_proj = None
            cond = g.final_film.condition(speaker_emb, emotion_emb)
            if cond is not None:
                film_proj = ops.linear(ops.cast(cond, d0), cache.get(fp.weight, d0), cache.get(fp.bias, d0))
        else:
            N.check(rc, "mv_gen_prologue")
        cond = film_proj
        views, o = [], 0
        for n in lens:
            views.append(flat[o:o + B * n].view(B, n))
            o += B * n
        x = self.inp.forward_cl(x, cache, alpha=alpha0, film_proj=film_proj, film_F=F, pooled_out=views[0])
        if return_stages:
            st["film" if cond is not None else "input_proj"] = x
        # two-product chain (streaming form): the last upsampler CAN write the chain's pair rows directly, so that the chain's first
        # pass takes its input by LDS-DMA like every later one.  Off by default (MV_GEN_PAIR=1 turns it on): measured at C2, that first
        # pass is bound by its stage-1 matrix work, not by its input path (21.8 us either way), while the upsampler's two 8-byte stores
        # per lane instead of one 16-byte store cost it 4 us (33 -> 37 us).
        w16 = bool(getattr(g, "_mv_mrf_w16", False)) and mixed is not None
        want_pair = (_GEN_PAIR and w16 and not return_stages and self.chain is not None and dt == torch.float32 and len(self.ups) >= 2
                     and sdt(len(self.ups)) == torch.float32 and sdt(len(self.ups) - 1) == torch.float32)
        x_pair = False
        for i, u in enumerate(self.ups):
            nxt = views[i + 1] if i + 1 < len(self.ups) else None
            # (mixed mode: the first fp32-storage upsampler takes the fp16 stream as it is - the one storage-type change)
            if want_pair and i == len(self.ups) - 1:
                x, x_pair = u.forward_cl(x, cache, pooled_in=views[i], pooled_out=nxt, act=N.ACT_LRELU,
                                         slope=g.upsample_layers[i][1].negative_slope, storage=sdt(i + 1), out_pair=True)
                continue
            x = u.forward_cl(x, cache, pooled_in=views[i], pooled_out=nxt, act=N.ACT_LRELU,
                             slope=g.upsample_layers[i][1].negative_slope, storage=sdt(i + 1))
            if return_stages:
                st[f"up{i}"] = x
        if x.dtype != dt:
            x = ops.cast(x, dt)                                # every upsampler ran in the early type
        # fp32 storage (split operands: matrix-pipe bound) runs the blocks as ONE chain - each block's GroupNorm(8,64) + residual is
        # applied by the next block's first pass, the last one by the output conv - 348 -> 281 us for the three blocks + output conv
        # at C2.  16-bit storage keeps the per-block kernels (the chain's extra stream transfer costs more than the MFMAs it saves:
        # 174 vs 166 us).  The per-stage outputs, when asked for, are chains over the first i+1 blocks.
        use_chain = self.chain is not None and x.dtype == torch.float32
        if use_chain and not return_stages:
            wt, bias = self.out_weights(mel.device)
            return self.chain.forward_out_cl(x, wt, bias, g.output_proj.kernel_size[0], N.ACT_TANH, w16=w16, x_pair=x_pair)
        if use_chain:
            x_in = x
            if return_stages:
                for i in range(len(self.mrfs) - 1):
                    st[f"mrf{i}"] = self.chain.forward_cl(x_in, i + 1, w16=w16)
            x = self.chain.forward_cl(x_in, w16=w16)
            if return_stages:
                st[f"mrf{len(self.mrfs) - 1}"] = x
        else:
            for i, m in enumerate(self.mrfs):
                x = m.forward_cl(x)
                if return_stages:
                    st[f"mrf{i}"] = x
        wt, bias = self.out_weights(mel.device)
        Bx, T, C = x.shape
        k = g.output_proj.kernel_size[0]
        wave = torch.empty(Bx, 1, T, device=mel.device, dtype=dt)
        N.call("mv_conv_out_act_packed_cl", c_void_p(x.data_ptr()), c_void_p(wt.data_ptr()), bias, c_void_p(wave.data_ptr()),
               Bx, T, C, k, k // 2, N.ACT_TANH, ops._dt(x), ops._stream())
        if return_stages:
            st = {kk: ops.ntc_to_nct(v) for kk, v in st.items()}   # stages are reported in the public NCT layout
            st["wave"] = wave
            return st
        return wave


def generator_fused_for(gen):
    f = getattr(gen, "_mv_fused", None)
    if f is None:
        f = GeneratorFused(gen)
        f = f if f.supported() else False
        object.__setattr__(gen, "_mv_fused", f)
    return f or None
