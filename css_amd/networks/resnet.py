"""ResNet-101 trunks built from HIP-backed layers.

``resnet101`` mirrors the reference's deep-stem constructor
(generalframeworks/networks/resnet.py:361-380 -> ResNet_Stem :142-291, Bottleneck :92-139);
``resnet101_tv`` is a torchvision-0.8.2-shaped ResNet-101 (the default backbone at
mix_label.py:68; torchvision is not installed in this image).  Both expose
``conv1 / bn1 / relu / maxpool / layer1..4`` -- all that ``DeepLabv3Plus_with_rep`` touches.
Activations are NHWC tensors between these modules.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..nn import HipBatchNorm2d, HipConv2d, HipMaxPool2d


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return HipConv2d(in_planes, out_planes, 3, stride, dilation, dilation, groups, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return HipConv2d(in_planes, out_planes, 1, stride, bias=False)


class Bottleneck(nn.Module):
    """1x1 -> BN -> ReLU -> 3x3(dil) -> BN -> ReLU -> 1x1 -> BN (+ downsample) -> add -> ReLU
    (reference resnet.py:119-139); BN + residual + ReLU run as one fused HIP kernel."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=HipBatchNorm2d):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        if torch.is_grad_enabled() and x.requires_grad:
            # x feeds conv1 AND the residual branch (identity, or the downsample convolution): take that branch's input from
            # conv1's tap so that the backward adds its gradient inside conv1's dgrad store (ops.conv2d)
            out, identity = self.conv1(x, tap=True)
            x = identity
        else:
            out = self.conv1(x)
        out = self.bn1(out, relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        if self.downsample is not None:
            identity = self.downsample[1](self.downsample[0](x))
        return self.bn3(self.conv3(out), res=identity, relu=True)


def _init_weights(model, zero_init_residual):
    for m in model.modules():
        if isinstance(m, HipConv2d):
            w = torch.empty_like(m.weight, memory_format=torch.contiguous_format)
            nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
            with torch.no_grad():
                m.weight.copy_(w)
        elif isinstance(m, HipBatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    if zero_init_residual:
        for m in model.modules():
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)


class _StemSequential(nn.Sequential):
    """conv-bn-relu-conv-bn-relu-conv with the reference's child indices 0,1,2,3,4,5,6."""

    def forward(self, x):
        x = self[1](self[0](x), relu=True)
        x = self[4](self[3](x), relu=True)
        return self[6](x)


class ResNet_Stem(nn.Module):
    def __init__(self, block, layers, zero_init_residual=True, groups=1, width_per_group=64,
                 replace_stride_with_dilation=(False, True, True), multi_grid=True, fpn=True):
        super().__init__()
        self._norm_layer = HipBatchNorm2d
        self.inplanes, self.dilation = 128, 1
        self.groups, self.base_width, self.fpn = groups, width_per_group, fpn
        self.conv1 = _StemSequential(conv3x3(3, 64, stride=2), HipBatchNorm2d(64), nn.ReLU(inplace=True),
                                     conv3x3(64, 64), HipBatchNorm2d(64), nn.ReLU(inplace=True),
                                     conv3x3(64, self.inplanes))
        self.bn1 = HipBatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(3, 2, 1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2],
                                       multi_grid=multi_grid)
        _init_weights(self, zero_init_residual)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False, multi_grid=False):
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       HipBatchNorm2d(planes * block.expansion))
        grids = [2, 2, 4] if multi_grid else [1] * blocks
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width,
                        previous_dilation * grids[0])]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation * grids[i]))
        return nn.Sequential(*layers)


class ResNet_TV(nn.Module):
    """torchvision-shaped ResNet (7x7/s2 stem, MaxPool(3,2,1), stride on conv2)."""

    def __init__(self, block, layers, zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = HipConv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = HipBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = HipMaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        _init_weights(self, zero_init_residual)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       HipBatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


def resnet101(pretrained=False, **kwargs):
    """Deep-stem ResNet-101 (reference resnet.py:361).  ``pretrained`` needs a checkpoint path in
    ``kwargs['checkpoint']``; there is no download in this environment."""
    ckpt = kwargs.pop("checkpoint", None)
    model = ResNet_Stem(Bottleneck, [3, 4, 23, 3], **kwargs)
    if pretrained:
        if ckpt is None:
            raise FileNotFoundError("resnet101(pretrained=True) needs checkpoint=<path to ImageNet state_dict>")
        missing, unexpected = model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=False)
        print(f"[Info] loaded ImageNet weights from {ckpt}; missing {missing}; unexpected {unexpected}")
    return model


def resnet101_tv(**kwargs):
    """torchvision.models.resnet101() stand-in (mix_label.py:68); load ``resnet101-63fe2227.pth`` with
    ``load_state_dict(..., strict=False)`` (its ``fc.*`` keys are unused here)."""
    return ResNet_TV(Bottleneck, [3, 4, 23, 3], **kwargs)
