"""Channels-last ("NTC") fused fast path: MFMA kernels that keep the residual stream in HBM as
[B, T, C] between launches.  The public modules stay NCT; the generator converts once at its
boundary (mel in, waveform out - a [B,1,T] waveform is the same memory in both layouts).
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import torch

from . import _native as N
from . import ops

_INT3 = ctypes.c_int * 3


def _versions(params):
    return tuple((id(p), p._version, p.data_ptr()) for p in params)


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
            if not 1 <= d <= 16:
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
            self._ws = {key: ws}
        out = torch.empty_like(x_cl)
        dil = _INT3(*self.blk.dilations)
        N.call("mv_mrf_block_fwd_cl", c_void_p(x_cl.data_ptr()), c_void_p(out.data_ptr()), c_void_p(packed.data_ptr()),
               dil, c_void_p(ws.data_ptr()), None if mask is None else c_void_p(mask.data_ptr()), float(mask_scale),
               B, T, float(self.blk.norm.eps), dt, ops._stream())
        return out


def mrf_fused_for(blk):
    f = getattr(blk, "_mv_fused", None)
    if f is None:
        f = MrfFused(blk) if MrfFused.supported(blk) else False
        object.__setattr__(blk, "_mv_fused", f)
    return f or None
