"""HIP-backed layer modules with the attribute surface the reference code touches.

``HipConv2d`` exposes ``stride / kernel_size / dilation / padding`` tuples and a class name
containing "Conv", so the reference's ``_nostride_dilate`` (deeplabv3.py:135-149) rewrites it
exactly like an ``nn.Conv2d``.  ``HipBatchNorm2d`` keeps nn.BatchNorm2d's parameter / buffer
names (checkpoint compatibility) but is deliberately NOT a ``_BatchNorm`` subclass:
``nn.SyncBatchNorm.convert_sync_batchnorm`` (mix_label.py:76) then leaves it alone and the
module synchronises its own statistics across ranks whenever a process group is initialised.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class HipConv2d(nn.Module):
    fuse_bn_stats = True     # bias-free convolutions emit batch-norm statistics from their epilogue (ops.conv2d)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped convolution is not on the CSS hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        w = torch.empty(out_channels, in_channels, *self.kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        if bias:
            bound = 1 / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    @classmethod
    def from_torch(cls, m: nn.Conv2d):
        c = cls(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups, m.bias is not None)
        with torch.no_grad():
            c.weight.copy_(m.weight)
            if m.bias is not None:
                c.bias.copy_(m.bias)
        return c

    def forward(self, x, tap=False):   # x: NHWC internal tensor
        if isinstance(x, ops.S2DInput):    # the network input staged space-to-depth: the stride-2 stem (csrc/conv_stem.hip)
            assert not tap
            return ops.conv2d_stem_s2d(x, self.weight, bn_stats=self.bias is None and self.training and self.fuse_bn_stats)
        # a bias-free convolution of this network always feeds a batch norm: let its epilogue produce the statistics
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0],
                          bn_stats=self.bias is None and self.training and self.fuse_bn_stats, tap=tap)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, bias={self.bias is not None}")


class HipBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.sync = True   # cross-rank statistics when torch.distributed is initialised (SyncBatchNorm semantics)

    @classmethod
    def from_torch(cls, m):
        b = cls(m.num_features, m.eps, m.momentum if m.momentum is not None else 0.1)
        with torch.no_grad():
            b.weight.copy_(m.weight)
            b.bias.copy_(m.bias)
            b.running_mean.copy_(m.running_mean)
            b.running_var.copy_(m.running_var)
            b.num_batches_tracked.copy_(m.num_batches_tracked)
        return b

    def forward(self, x, res=None, relu=False, out_into=None, pool=None):
        """``pool``: a HipMaxPool2d(3, 2, 1) that follows this layer (the stem): applied inside the batch norm's apply pass
        (ops.bn_act(pool=...): the normalised activation is never written); any other pool configuration runs as a pass of its own."""
        if self.training:
            ops.count_bn_batch(self.num_batches_tracked)
        if pool is not None and not (ops.fused_stem_pool() and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1 and res is None
                                     and out_into is None):
            return pool(ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, res, relu, self.training, self.momentum, self.eps,
                                   self.sync, out_into=out_into))
        po = None
        if pool is not None:
            po = (ops.pool_out_size(x.shape[1], 3, 2, 1, pool.ceil_mode), ops.pool_out_size(x.shape[2], 3, 2, 1, pool.ceil_mode))
        return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, res, relu, self.training,
                          self.momentum, self.eps, self.sync, out_into=out_into, pool=po)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


class HipMaxPool2d(nn.Module):
    def __init__(self, kernel_size=3, stride=2, padding=1, ceil_mode=False):
        super().__init__()
        self.kernel_size, self.stride, self.padding, self.ceil_mode = kernel_size, stride, padding, ceil_mode

    @classmethod
    def from_torch(cls, m):
        def one(v):
            return v[0] if isinstance(v, (tuple, list)) else v
        return cls(one(m.kernel_size), one(m.stride), one(m.padding), m.ceil_mode)

    def forward(self, x):
        return ops.maxpool(x, self.kernel_size, self.stride, self.padding, self.ceil_mode)


class ConvBNReLU(nn.Sequential):
    """nn.Sequential(conv, bn, relu) with a fused forward; child names '0','1','2' as in the reference."""

    def __init__(self, conv, bn):
        super().__init__(conv, bn, nn.ReLU())

    def forward(self, x):
        return self[1](self[0](x), relu=True)


def convert_module(m: nn.Module) -> nn.Module:
    """Recursively replace nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d by their HIP-backed twins
    (same attribute names, parameters copied).  Used when a caller hands a torchvision-style
    ResNet to ``DeepLabv3Plus_with_rep`` (mix_label.py:68-75)."""
    if isinstance(m, nn.Conv2d):
        return HipConv2d.from_torch(m)
    if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
        return HipBatchNorm2d.from_torch(m)
    if isinstance(m, nn.MaxPool2d):
        return HipMaxPool2d.from_torch(m)
    for name, child in list(m.named_children()):
        setattr(m, name, convert_module(child))
    return m
