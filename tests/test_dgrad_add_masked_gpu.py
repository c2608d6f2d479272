"""css_conv2d_dgrad_add_masked (include/css_hip.h): the data gradient of a Bottleneck's conv1 plus the residual branch's gradient, with the
ReLU backward of the residual sum applied to that addend inside the store (/root/reference/generalframeworks/networks/resnet.py:119-139:
``out += identity; out = relu(out)`` - autograd's ReLU backward masks the gradient before it reaches both bn3 and the identity).

* through the C ABI, for every kernel family the dispatch can take (weight-stationary short-K kernel, the persistent 256x256 kernels with
  their leftover launches, the 128x64 kernels of the narrow layers, the fp32 path): bit-identical to css_conv2d_dgrad_add on an addend that
  was masked beforehand, and against torch-CPU fp32 ``conv_transpose`` arithmetic;
* through autograd: an identity Bottleneck's backward with the mask applied in the store (default) against CSS_BN_EAGER_DRES behaviour
  (bn_bwd_apply writes the masked copy): every gradient bit-identical, the side table of pending masks empty afterwards, and the masked
  entry point actually called.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import bf16_round, dev, rel_err  # noqa: E402


def _mask_bytes(keep, vec):
    """[M][C] bool -> [M][C / vec] uint8, bit e of byte (m, v) = keep[m, v * vec + e] (the layout css_bn_apply_mask writes)."""
    m, c = keep.shape
    w = (1 << torch.arange(vec, dtype=torch.int32))
    return (keep.reshape(m, c // vec, vec).to(torch.int32) * w).sum(-1).to(torch.uint8)


# forward conv Cin -> Cout (R x R, dilation d); the data gradient is the product K = R*R*Cout -> N = Cin: N, H, W, Cin, Cout, R, d, dtype
CASES = [
    (2, 19, 23, 1024, 256, 1, 1, torch.bfloat16),     # conv_ws_kernel (K = 256), ragged last tile
    (8, 65, 65, 1024, 256, 1, 1, torch.bfloat16),     # ... 4-5 tiles per stream (the mask bytes are part of the counted vmcnt)
    (16, 65, 65, 2048, 512, 1, 1, torch.bfloat16),    # K = 512 -> N = 2048: persistent 256x256 kernel + leftover launch (layer4 conv1)
    (4, 65, 65, 512, 128, 1, 1, torch.bfloat16),      # K = 128: conv_ws_kernel
    (2, 129, 129, 256, 64, 1, 1, torch.bfloat16),     # K = 64 (layer1 conv1)
    (3, 21, 21, 320, 264, 3, 2, torch.bfloat16),      # 3x3, ragged channels (Cin % 256 != 0)
    (2, 33, 33, 64, 24, 1, 1, torch.bfloat16),        # narrow: the 128x64 kernels
    (2, 17, 17, 64, 32, 1, 1, torch.float32),         # fp32 path: four elements per mask byte
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_masked_addend_matches_premasked_addend(case):
    from css_amd import ops
    from css_amd._lib import call, dev_stream, dtype_code
    n, h, w, cin, cout, r, dil, dt = case
    vec = ops.vec_of(dt)
    pad = dil * (r // 2)
    g = torch.Generator().manual_seed(991 + cin + cout + h)
    rnd = bf16_round if dt == torch.bfloat16 else (lambda t: t)
    wt = rnd(torch.randn(cout, cin, r, r, generator=g) / (cin * r * r) ** 0.5)
    dy = rnd(torch.randn(n, h, w, cout, generator=g))
    add = rnd(torch.randn(n, h, w, cin, generator=g))
    keep = torch.rand(n * h * w, cin, generator=g) < 0.6
    keep[:7] = True
    keep[7:13] = False
    mask = _mask_bytes(keep, vec)
    add_pre = torch.where(keep.reshape(n, h, w, cin), add, torch.zeros(()))

    wg = wt.to(dev()).contiguous(memory_format=torch.channels_last)
    wtg = ops.prepared_weight(wg, dt, cin, True)
    dyg, addg, preg, mg = dy.to(dev(), dt), add.to(dev(), dt), add_pre.to(dev(), dt), mask.to(dev())
    dx_m = torch.full((n, h, w, cin), float("nan"), dtype=dt, device=dev())
    dx_p = torch.full_like(dx_m, float("nan"))
    d, st = dev_stream(dyg)
    flops = 2.0 * n * h * w * cin * cout * r * r
    call("css_conv2d_dgrad_add_masked", dyg, wtg, dx_m, addg, cin, mg, n, h, w, cin, cin, h, w, cout, cout, r, r, 1, pad, dil, flops,
         dtype_code(dt), d, st)
    call("css_conv2d_dgrad_add", dyg, wtg, dx_p, preg, cin, n, h, w, cin, cin, h, w, cout, cout, r, r, 1, pad, dil, flops, dtype_code(dt), d, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx_m.float()).all())
    assert torch.equal(dx_m, dx_p), (dx_m.float() - dx_p.float()).abs().max().item()
    ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), wt, padding=pad, dilation=dil).permute(0, 2, 3, 1) + add_pre
    e = rel_err(dx_m.float().cpu(), ref)
    print(case, "vs torch-CPU fp32:", e)
    assert e < (2e-2 if dt == torch.bfloat16 else 2e-5)


def test_masked_entry_point_rejects_what_it_cannot_mask():
    from css_amd import _lib
    from css_amd._lib import dev_stream
    z = torch.zeros(64, dtype=torch.bfloat16, device=dev())
    d, st = dev_stream(z)
    lib = _lib.lib()
    # Cin = 12 is not a whole number of 16-byte vectors; a missing mask is an argument error too
    assert lib.css_conv2d_dgrad_add_masked(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 12, z.data_ptr(), 1, 2, 2, 12, 12, 2, 2, 8, 8, 1, 1, 1,
                                           0, 1, 0.0, 1, d, st) != 0
    assert lib.css_conv2d_dgrad_add_masked(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 16, None, 1, 2, 2, 16, 16, 2, 2, 8, 8, 1, 1, 1,
                                           0, 1, 0.0, 1, d, st) != 0


@pytest.mark.parametrize("shape", [(2, 33, 33, 256, 64), (4, 65, 65, 1024, 256)], ids=lambda s: "x".join(map(str, s)))
def test_identity_bottlenecks_backward_lazy_mask_equals_eager_copy(shape):
    """Two identity Bottlenecks in a row (resnet.py:119-139): the first one's conv1 receives the second one's residual gradient."""
    from css_amd import _lib, ops
    from css_amd.networks.resnet import Bottleneck
    n, h, w, c, planes = shape
    torch.manual_seed(5)
    blocks = torch.nn.Sequential(Bottleneck(c, planes), Bottleneck(c, planes)).to(dev()).train()
    for b in blocks:
        torch.nn.init.normal_(b.bn3.weight, 1.0, 0.2)
    x0 = torch.randn(n, h, w, c).to(dev(), torch.bfloat16)
    gout = torch.randn(n, h, w, c).to(dev(), torch.bfloat16)
    calls = []
    real_call = ops.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    def run(lazy):
        ops._lazy_dres = lazy
        ops.invalidate_weight_cache()
        for p in blocks.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        calls.clear()
        ops.call = spy
        try:
            y = blocks(x)
            (y.float() * gout.float()).sum().backward()
        finally:
            ops.call = real_call
        torch.cuda.synchronize()
        ops.assert_no_lazy_res_grads()
        return x.grad.clone(), [p.grad.clone() for p in blocks.parameters()], list(calls)

    prev = ops._lazy_dres
    try:
        dx_l, gp_l, c_l = run(True)
        dx_e, gp_e, c_e = run(False)
    finally:
        ops._lazy_dres = prev
    assert c_l.count("css_conv2d_dgrad_add_masked") == 2 and c_e.count("css_conv2d_dgrad_add_masked") == 0, (c_l, c_e)
    assert c_e.count("css_conv2d_dgrad_add") == 2 and c_l.count("css_conv2d_dgrad_add") == 0
    assert bool(torch.isfinite(dx_l.float()).all()) and float(dx_l.float().abs().max()) > 0
    assert torch.equal(dx_l, dx_e), (dx_l.float() - dx_e.float()).abs().max().item()
    for a, b in zip(gp_l, gp_e):
        assert rel_err(a.float().cpu(), b.float().cpu()) < 1e-5       # (weight gradients: fp32 slab sums in a fixed order; identical inputs)


def test_second_consumer_of_the_tap_raises_instead_of_adding_an_unmasked_gradient():
    """the lazily masked residual gradient is bound to its consumer.  If the tap tensor gets a second consumer, autograd hands the
    tapped convolution a SUM that contains the unmasked gradient: its backward must raise, never add it - under a plain loss.backward()
    with no trainer around; and an aborted backward must leave nothing behind for the next one."""
    from css_amd import _lib, ops
    from css_amd.nn import HipBatchNorm2d, HipConv2d
    torch.manual_seed(2)
    c = 64
    conv, bn = HipConv2d(c, c, 1, bias=False).to(dev()).train(), HipBatchNorm2d(c).to(dev()).train()
    x = torch.randn(2, 17, 17, c).to(dev(), torch.bfloat16).requires_grad_(True)

    def forward(second_consumer):
        y, tap = conv(x, tap=True)
        out = bn(y, res=tap, relu=True)
        return out.float().sum() + (tap.float().sum() * 0.5 if second_consumer else 0.0)

    if not ops._lazy_dres:
        pytest.skip("CSS_BN_EAGER_DRES=1: nothing is parked")
    with pytest.raises(_lib.CssHipError, match="second consumer"):
        forward(True).backward()
    # the aborted pass left its pair parked in a link of a dead graph: the next pass starts clean, runs, and its own check passes
    x.grad = None
    forward(False).backward()
    torch.cuda.synchronize()
    ops.assert_no_lazy_res_grads()
    assert x.grad is not None and bool(torch.isfinite(x.grad.float()).all())
