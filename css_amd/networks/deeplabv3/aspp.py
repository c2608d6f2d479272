"""ASPP head on HIP kernels (reference generalframeworks/networks/deeplabv3/aspp.py:17-72)."""
from __future__ import annotations

import torch.nn as nn

from ... import ops
from ...nn import ConvBNReLU, HipBatchNorm2d, HipConv2d


class ASPPConv(ConvBNReLU):
    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(HipConv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         HipBatchNorm2d(out_channels))


class ASPPPooling(nn.Sequential):
    """global average pool -> 1x1 conv -> BN -> ReLU -> broadcast back (aspp.py:27-38).  Child indices
    0 (pool), 1 (conv), 2 (bn), 3 (relu) match the reference's state_dict keys ``convs.4.{1,2}.*``."""

    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), HipConv2d(in_channels, out_channels, 1, bias=False),
                         HipBatchNorm2d(out_channels), nn.ReLU())

    def forward(self, x, out_into=None):
        h, w = x.shape[1], x.shape[2]
        p = ops.global_avg_pool(x)
        p = self[2](self[1](p), relu=True)
        return ops.broadcast_hw(p, h, w, out_into)   # bilinear(align_corners=False) from a 1x1 map == broadcast


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates):
        super().__init__()
        out_channels = 256
        modules = nn.ModuleList()
        modules.append(ConvBNReLU(HipConv2d(in_channels, out_channels, 1, bias=False), HipBatchNorm2d(out_channels)))
        for rate in tuple(atrous_rates):
            modules.append(ASPPConv(in_channels, out_channels, rate))
        modules.append(ASPPPooling(in_channels, out_channels))
        self.convs = modules
        self.project = ConvBNReLU(HipConv2d(5 * out_channels, out_channels, 1, bias=False), HipBatchNorm2d(out_channels))

    def forward(self, x):
        import torch
        # the five branches write their outputs straight into the channel slices of ONE [N,H,W,1280] buffer (no concat copies,
        # and the backward reads each slice of the gradient in place)
        n, h, w, _ = x.shape
        oc = self.project[0].in_channels // len(self.convs)
        buf = torch.empty((n, h, w, oc * len(self.convs)), dtype=x.dtype, device=x.device)
        chain = torch.is_grad_enabled() and x.requires_grad
        # x feeds five branches; chaining the four convolutions through their input taps (ops.conv2d) makes every branch's
        # dgrad add the gradient accumulated so far in its own store pass instead of four torch adds over 554 MB tensors
        outs, cur = [], x
        for i, m in enumerate(list(self.convs)[:-1]):
            if chain:
                y, cur = m[0](cur, tap=True)
            else:
                y = m[0](cur)
            outs.append(m[1](y, relu=True, out_into=(buf, i * oc)))
        outs.append(self.convs[-1](cur, out_into=(buf, (len(self.convs) - 1) * oc)))
        return self.project(ops.cat_from_views(buf, *outs))
