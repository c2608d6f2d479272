"""DeepLabv3+ with a classifier and a 256-d representation head on HIP kernels.

Drop-in for ``generalframeworks.networks.deeplabv3.deeplabv3.DeepLabv3Plus_with_rep``
(reference deeplabv3.py:90-169): same constructor, same submodule names (checkpoint keys),
``forward(x[B,3,H,W]) -> (prediction[B,K,h,w], representation[B,output_dim,h,w])``.
Outputs are logical NCHW tensors in channels_last memory (physically the NHWC buffers the
kernels wrote), carrying autograd history to the student parameters.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ...nn import ConvBNReLU, HipBatchNorm2d, HipConv2d, convert_module
from .aspp import ASPP


class _Head(nn.Sequential):
    """conv3x3 -> BN -> ReLU -> conv1x1(bias); child indices 0,1,2,3 as in deeplabv3.py:121-133."""

    def __init__(self, cin, cout):
        super().__init__(HipConv2d(cin, 256, 3, padding=1, bias=False), HipBatchNorm2d(256), nn.ReLU(),
                         HipConv2d(256, cout, 1))

    def forward(self, x, tap=False):
        if tap:      # also hand back the input alias whose gradient this head's first dgrad absorbs (ops.conv2d)
            y, xa = self[0](x, tap=True)
            return self[3](self[1](y, relu=True)), xa
        return self[3](self[1](self[0](x), relu=True))


class DeepLabv3Plus_with_rep(nn.Module):
    def __init__(self, orig_resnet, dilate_scale=16, num_classes=21, output_dim=256):
        super().__init__()
        orig_resnet = convert_module(orig_resnet)      # nn.Conv2d/BatchNorm2d/MaxPool2d -> HIP-backed twins
        if dilate_scale == 8:
            orig_resnet.layer3.apply(partial(self._nostride_dilate, dilate=2))
            orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=4))
            aspp_dilate = [12, 24, 36]
        elif dilate_scale == 16:
            orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=2))
            aspp_dilate = [6, 12, 18]
        else:
            raise ValueError("dilate_scale must be 8 or 16")
        self.resnet_conv1 = orig_resnet.conv1
        self.resnet_bn1 = orig_resnet.bn1
        self.resnet_relu1 = orig_resnet.relu
        self.resnet_maxpool = orig_resnet.maxpool
        self.resnet_layer1 = orig_resnet.layer1
        self.resnet_layer2 = orig_resnet.layer2
        self.resnet_layer3 = orig_resnet.layer3
        self.resnet_layer4 = orig_resnet.layer4
        self.ASPP = ASPP(2048, aspp_dilate)
        self.project = ConvBNReLU(HipConv2d(256, 48, 1, bias=False), HipBatchNorm2d(48))
        self.classifier = _Head(304, num_classes)
        self.representation = _Head(304, output_dim)
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        """torch.float32 (parity path, exact-fp32 MFMA) or torch.bfloat16 (bf16 MFMA, fp32 accumulate)."""
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    @staticmethod
    def _nostride_dilate(m, dilate):
        # same rule as the reference (deeplabv3.py:135-149), applied to HipConv2d
        if m.__class__.__name__.find("Conv") != -1 and hasattr(m, "kernel_size") and hasattr(m, "stride"):
            if m.stride == (2, 2):
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation = (dilate // 2, dilate // 2)
                    m.padding = (dilate // 2, dilate // 2)
            elif m.kernel_size == (3, 3):
                m.dilation = (dilate, dilate)
                m.padding = (dilate, dilate)

    def forward_nhwc(self, x):
        """x: staged NHWC tensor -> (prediction, representation) NHWC."""
        x = self.resnet_bn1(self.resnet_conv1(x), relu=True, pool=self.resnet_maxpool)      # (bn + ReLU + max pool in one pass)
        x_low = self.resnet_layer1(x)
        fuse = torch.is_grad_enabled() and x_low.requires_grad    # fold fan-out gradient sums into dgrad store passes (ops.conv2d taps)
        # decoder input = [project(x_low) | up-sampled ASPP feature]: both producers write into one buffer (no concat copy)
        n, hl, wl, _ = x_low.shape
        c_low, c_up = self.project[0].out_channels, self.ASPP.project[0].out_channels
        dbuf = torch.empty((n, hl, wl, c_low + c_up), dtype=x_low.dtype, device=x_low.device)
        if fuse:
            p, x_low = self.project[0](x_low, tap=True)
        else:
            p = self.project[0](x_low)
        low = self.project[1](p, relu=True, out_into=(dbuf, 0))
        x = self.resnet_layer2(x_low)
        x = self.resnet_layer3(x)
        x = self.resnet_layer4(x)
        feature = self.ASPP(x)
        up = ops.bilinear(feature, hl, wl, out_into=(dbuf, c_low))
        dec = ops.cat_from_views(dbuf, low, up)
        if fuse:
            pred, dec = self.classifier(dec, tap=True)
            return pred, self.representation(dec)
        return self.classifier(dec), self.representation(dec)

    def stage(self, xs):
        """[Bi,3,H,W] fp32 images -> the staged network input of ``forward_nhwc``: NHWC in the compute dtype, or - bf16 and a stride-2 stem
        that conv_stem_s2d_kernel takes - the space-to-depth staging (ops.S2DInput)."""
        first = self.resnet_conv1 if hasattr(self.resnet_conv1, "kernel_size") else self.resnet_conv1[0]
        return ops.stage_inputs(list(xs), self.compute_dtype, s2d=ops.stem_s2d_ok(first, self.compute_dtype))

    def forward(self, x):
        pred, rep = self.forward_nhwc(self.stage([x]))
        return pred.permute(0, 3, 1, 2), rep.permute(0, 3, 1, 2)
