"""The convolution kernels that carry the bench, at the LAUNCH SHAPES of the bench (32 images of 65x65 / 129x129 for the 513^2
workload, 16 images of 97x97 for the 769^2 one), bf16, against torch-CPU fp32 ``F.conv2d`` on the same bf16-rounded inputs
(forward, data gradient, weight gradient) and - for the fused epilogue - against an fp64 reduction of the CPU output.

These are the shapes of /root/reference/generalframeworks/networks/resnet.py:119-139 (layer3 / layer4 Bottlenecks, dilated),
deeplabv3/aspp.py:36-38 (dilated 3x3 on 2048 channels) and deeplabv3/deeplabv3.py:151-169 (decoder 3x3 on 304 channels); the
small cases of test_ops_gpu.py never reach a multi-round grid, the XCD remap, the leftover split or the wgrad pixel split.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import bf16_round, dev, rel_err  # noqa: E402

# N, H, W, Cin, Cout, k, pad, dil
CASES = [
    (32, 65, 65, 1024, 256, 1, 0, 1),       # layer3 conv1: read-bound 1x1
    (32, 65, 65, 256, 1024, 1, 0, 1),       # layer3 conv3: write-bound short-K 1x1
    (32, 65, 65, 256, 256, 3, 2, 2),        # layer3 conv2
    (32, 65, 65, 512, 512, 3, 4, 4),        # layer4 conv2
    (32, 65, 65, 2048, 256, 3, 12, 12),     # ASPP branch, dilation 12
    (32, 65, 65, 2048, 256, 3, 36, 36),     # ASPP branch, dilation 36: most kernel rows are all padding
    (32, 129, 129, 304, 256, 3, 1, 1),      # decoder head: Cin not a multiple of the K tile
    (16, 97, 97, 256, 256, 3, 2, 2),        # 769^2 geometry, B = 8 + 8
    (16, 97, 97, 1024, 256, 1, 0, 1),
]


def prof_read():
    from css_amd import _lib
    out = {}
    for kind in range(8):
        ms, n, w = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.lib().css_prof_read(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        out[kind] = (ms.value, n.value, w.value)
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf16_bench_shapes_vs_cpu(case):
    from css_amd import _lib, ops
    n, h, w, cin, cout, k, pad, dil = case
    g = torch.Generator().manual_seed(1234 + cin + cout + k + dil)
    # NHWC generation (cheap), logical NCHW views for the CPU reference
    x = bf16_round(torch.randn(n, h, w, cin, generator=g) + 0.25)
    wt = bf16_round(torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5)
    gy = bf16_round(torch.randn(n, h, w, cout, generator=g))
    xr = x.permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, pad, dil)
    yr.backward(gy.permute(0, 3, 1, 2))
    y_ref = yr.detach().permute(0, 2, 3, 1)                     # [N,H,W,Cout] fp32, un-rounded

    xg = x.to(dev(), torch.bfloat16).requires_grad_(True)
    wg = wt.permute(0, 3, 1, 2).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lib = _lib.lib()
    lib.css_prof_reset()
    lib.css_prof_enable(1)
    with ops.bn_groups(2):
        y = ops.conv2d(xg, wg, None, 1, pad, dil, bn_stats=True)
    assert hasattr(y, "_css_bnstats"), "the bench path emits the BN statistics from the conv epilogue"
    y.backward(gy.to(dev(), torch.bfloat16))
    torch.cuda.synchronize()
    lib.css_prof_enable(0)
    prof = prof_read()
    lib.css_prof_reset()
    # dispatch: the 256x256 LDS-DMA kernels carry these shapes (kinds 5/6/7 of include/css_hip.h); forward and dgrad each cover
    # every output row exactly once between their main and leftover launches (shares of alg_work add up to the call's FLOPs)
    flops = 2.0 * n * h * w * cout * k * k * cin
    assert prof[5][1] >= 1 and prof[6][1] >= 1 and prof[7][1] >= 1, prof
    assert abs(prof[5][2] + prof[0][2] - flops) < 1e-6 * flops and abs(prof[6][2] + prof[1][2] - flops) < 1e-6 * flops, prof

    e_y = rel_err(y.detach().float().cpu(), y_ref)
    e_dx = rel_err(xg.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1))
    e_dw = rel_err(wg.grad.cpu(), wr.grad)
    print(f"{case}: fwd {e_y:.2e} dgrad {e_dx:.2e} wgrad {e_dw:.2e}; launches 256-kernel fwd/dgrad/wgrad "
          f"{prof[5][1]:.0f}/{prof[6][1]:.0f}/{prof[7][1]:.0f}, other fwd/dgrad/wgrad {prof[0][1]:.0f}/{prof[1][1]:.0f}/{prof[2][1]:.0f}")
    assert e_y < 2e-2 and e_dx < 2e-2 and e_dw < 2e-2

    # fused statistics against an fp64 reduction of the CPU output (two groups = the two batched forward passes)
    from css_amd._lib import call, dev_stream
    part, mg, groups, c_, bm = y._css_bnstats
    sums = torch.empty(groups * 2 * cout + groups, dtype=torch.float64, device=dev())   # [G][2][C] sums + [G] row counts
    d, st = dev_stream(y)
    call("css_bn_reduce_finalize_slabs", part, mg * groups, mg, groups, float(mg), None, None, None, None, 0.0, 0.0, None, None, None, None,
         sums, cout, y, cout, bm, d, st)
    yy = y_ref.double().reshape(groups, -1, cout)
    want = torch.stack([yy.sum(1), (yy * yy).sum(1)], 1)                   # [G][2][C]
    got = sums.cpu()[:groups * 2 * cout].reshape(groups, 2, cout)
    # the kernel sums the bf16-ROUNDED outputs: per element a relative rounding error <= 2^-9, random in sign
    scale_s = yy.abs().sum(1).max().item()
    assert ((got[:, 0] - want[:, 0]).abs().max().item() < 2e-3 * scale_s)
    assert rel_err(got[:, 1], want[:, 1]) < 2e-3


def test_wgrad_split_plan_at_bench_scale():
    """The weight-gradient launcher splits the pixels of a bench-size layer over more than one slice per XCD."""
    from css_amd import _lib
    n_cu = _lib.query("css_device_cu_count", 0)
    splits = _lib.query("css_wgrad_splits", 32 * 65 * 65, 256 * 9, 256, 1, n_cu)
    assert splits > 8, splits
    # round 4: the slices fill whole rounds of the CHIP (not of every XCD) with as few slabs as that allows
    for ktot, cd in ((256 * 9, 256), (2048 * 9, 256), (512 * 9, 512), (1024, 256)):
        s_ = _lib.query("css_wgrad_splits", 32 * 65 * 65, ktot, cd, 1, n_cu)
        wgs = -(-ktot // 256) * -(-cd // 256) * s_
        assert s_ >= 4 and wgs / n_cu - int(wgs / n_cu - 1e-9) >= 0.9, (ktot, cd, s_, wgs)
