"""ODConv1d / ODConvTranspose1d on MI355X.

Drop-in for the reference's ``hifigan_modified/odconv.py`` (ctor: odconv.py:17-18,116-117; forward:
:73-108,:172-205): same constructor arguments, parameter names/shapes (``kernels``, ``bias``,
``kernel_attention.1.*`` and the never-used ``spatial_attention`` / ``in_channel_attention`` /
``out_channel_attention`` parameter sets) and the same initialisation order, so a seed or a
reference ``state_dict`` gives identical weights.  The arithmetic is one HIP launch for the kernel
attention and one for attention-weighted kernel aggregation + convolution (+ optional activation).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn


def _holder(*convs):
    """nn.Sequential whose conv children sit at indices 1, 3, ... (the reference's key layout:
    pooling at 0, conv at 1, nonlinearity at 2, conv at 3, ...).  Only the parameters matter."""
    mods = [nn.Identity()]
    for c in convs:
        mods += [c, nn.Identity()]
    return nn.Sequential(*mods)


class _ODConvBase(nn.Module):
    _transposed = False

    def _build(self, in_channels, out_channels, kernel_size, K, reduction_factor):
        shape = ((K, in_channels, out_channels, kernel_size) if self._transposed
                 else (K, out_channels, in_channels, kernel_size))
        self.kernels = nn.Parameter(torch.randn(*shape))
        self.bias = nn.Parameter(torch.randn(K, out_channels))
        self.kernel_attention = _holder(nn.Conv1d(in_channels, K, 1))
        # constructed-but-unused attention heads (odconv.py:42-62): kept for state_dict / RNG parity,
        # never read by forward and never given a gradient
        self.spatial_attention = _holder(nn.Conv1d(in_channels, kernel_size, 1))
        self.in_channel_attention = _holder(nn.Conv1d(in_channels, in_channels // reduction_factor, 1),
                                            nn.Conv1d(in_channels // reduction_factor, in_channels, 1))
        self.out_channel_attention = _holder(nn.Conv1d(in_channels, out_channels // reduction_factor, 1),
                                             nn.Conv1d(out_channels // reduction_factor, out_channels, 1))
        self._initialize_weights()

    def _initialize_weights(self):
        for bank in self.kernels:
            nn.init.kaiming_normal_(bank, mode="fan_out", nonlinearity="relu")
        nn.init.zeros_(self.bias)

    def unused_parameters(self):
        """Parameters that never receive a gradient (excluded from the data-parallel buckets)."""
        for name in ("spatial_attention", "in_channel_attention", "out_channel_attention"):
            yield from getattr(self, name).parameters()

    def _fused(self):
        """Channels-last MFMA launcher of this layer (None when the geometry is outside the fused kernel's support)."""
        f = getattr(self, "_mv_fused", None)
        if f is None:
            from .fused import OdconvFused
            f = OdconvFused(self)
            f = f if f.supported() else False
            object.__setattr__(self, "_mv_fused", f)
        return f or None

    def attention(self, x):
        """alpha [B,K] (fp32) = softmax_K(Conv1x1(mean_t x))."""
        att = self.kernel_attention[1]
        return Fn.odconv_attention(x, att.weight, att.bias)


class ODConv1d(_ODConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, K=4, reduction_factor=4):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups, self.K = stride, padding, dilation, groups, K
        if groups != 1:
            # the reference passes `groups` to F.conv1d with full-width kernels, which raises there too
            raise ValueError("ODConv1d: kernels are [K,Cout,Cin,ks]; groups != 1 is not a valid configuration")
        self._build(in_channels, out_channels, kernel_size, K, reduction_factor)

    def forward(self, x, act=None, slope=0.1):
        att = self.kernel_attention[1]
        return Fn.odconv1d(x, self.kernels, self.bias, att.weight, att.bias, self.stride, self.padding,
                           self.dilation, act, slope, fused=self._fused())


class ODConvTranspose1d(_ODConvBase):
    _transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 output_padding=0, dilation=1, groups=1, K=4, reduction_factor=4):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.output_padding = stride, padding, output_padding
        self.dilation, self.groups, self.K = dilation, groups, K
        if groups != 1:
            raise ValueError("ODConvTranspose1d: kernels are [K,Cin,Cout,ks]; groups != 1 is not a valid configuration")
        self._build(in_channels, out_channels, kernel_size, K, reduction_factor)

    def forward(self, x, act=None, slope=0.1):
        att = self.kernel_attention[1]
        return Fn.odconv_transpose1d(x, self.kernels, self.bias, att.weight, att.bias, self.stride, self.padding,
                                     self.output_padding, self.dilation, act, slope, fused=self._fused())
