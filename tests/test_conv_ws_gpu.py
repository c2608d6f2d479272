"""conv_ws_kernel (css_amd/csrc/conv_ws.hip): the weight-stationary kernel of the short-K 1x1 class - conv3 of a Bottleneck in the forward
pass (/root/reference/generalframeworks/networks/resnet.py:131-133, planes -> 4 planes, followed by bn3) and conv1 of a Bottleneck in its
data-gradient form (resnet.py:123-125; the residual gradient is added in the store: css_conv2d_dgrad_add) - against torch-CPU fp32
``F.conv2d`` on the same bf16-rounded inputs, through the C ABI, with the dispatch asserted (css_conv_ws_applies + launch counts).

Covers K = 64 / 128 / 256 / 512, 1 / 2 / 4 / 8 output panels, ragged row counts (M % 128 != 0), streams of 0, 1, 3+ tiles (the counted vmcnt has
its steady state from the third tile on), the statistics slabs across a group boundary, the addend variant at bench scale, and the
CSS_NO_WS_CONV switch."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import bf16_round, dev, rel_err  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prof_read():
    from css_amd import _lib
    out = {}
    for kind in range(8):
        ms, n, w = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.lib().css_prof_read(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        out[kind] = (ms.value, n.value, w.value)
    return out


def ws_applies(m, k, n, stats=0, addend=0):
    from css_amd import _lib
    n_cu = _lib.query("css_device_cu_count", 0)
    return _lib.query("css_conv_ws_applies", m, k, k, n, n, 1, 1, 1, 0, stats, addend, n, 0, 1, n_cu)


def run_case(n, h, w, cin, cout, tap, seed=0):
    """1x1 convolution cin -> cout with fused statistics forward; backward with (tap) or without the residual-gradient addend."""
    from css_amd import _lib, ops
    g = torch.Generator().manual_seed(4321 + cin + cout + n * h + seed)
    x = bf16_round(torch.randn(n, h, w, cin, generator=g) + 0.25)
    wt = bf16_round(torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5)
    gy = bf16_round(torch.randn(n, h, w, cout, generator=g))
    gt = bf16_round(torch.randn(n, h, w, cin, generator=g))
    xr = x.permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr)
    (yr * gy.permute(0, 3, 1, 2)).sum().backward()
    dx_ref = xr.grad.permute(0, 2, 3, 1) + (gt if tap else 0)
    y_ref = yr.detach().permute(0, 2, 3, 1)

    xg = x.to(dev(), torch.bfloat16).requires_grad_(True)
    wg = wt.permute(0, 3, 1, 2).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lib = _lib.lib()
    lib.css_prof_reset()
    lib.css_prof_enable(1)
    with ops.bn_groups(2):
        out = ops.conv2d(xg, wg, None, 1, 0, 1, bn_stats=True, tap=tap)
    y, xt = out if tap else (out, None)
    loss = (y.float() * gy.to(dev())).sum()
    if tap:
        loss = loss + (xt.float() * gt.to(dev())).sum()
    loss.backward()
    torch.cuda.synchronize()
    lib.css_prof_enable(0)
    prof = prof_read()
    lib.css_prof_reset()
    e_y = rel_err(y.detach().float().cpu(), y_ref)
    e_dx = rel_err(xg.grad.float().cpu(), dx_ref)
    e_dw = rel_err(wg.grad.cpu(), wr.grad)
    stats = getattr(y, "_css_bnstats", None)
    e_s = None
    if stats is not None:
        from css_amd._lib import call, dev_stream
        part, mg, groups, c_, bm = stats
        sums = torch.empty(groups * 2 * cout + groups, dtype=torch.float64, device=dev())
        d, st = dev_stream(y)
        call("css_bn_reduce_finalize_slabs", part, mg * groups, mg, groups, float(mg), None, None, None, None, 0.0, 0.0, None, None, None, None,
             sums, cout, y, cout, bm, d, st)
        yy = y_ref.double().reshape(groups, -1, cout)
        want = torch.stack([yy.sum(1), (yy * yy).sum(1)], 1)
        got = sums.cpu()[:groups * 2 * cout].reshape(groups, 2, cout)
        e_s = ((got[:, 0] - want[:, 0]).abs().max().item() / yy.abs().sum(1).max().item(), rel_err(got[:, 1], want[:, 1]))
    return e_y, e_dx, e_dw, e_s, prof


# forward GEMM K = cin -> N = cout on conv_ws (with statistics): N, H, W, Cin, Cout
FWD_CASES = [
    (2, 19, 23, 256, 1024),     # 7 tiles, 4 panels, ragged last tile, a group boundary inside a slab
    (4, 17, 17, 64, 256),       # K = 64: one stage per tile, one panel
    (2, 33, 33, 128, 2048),     # K = 128, 8 panels
    (3, 40, 40, 128, 512),      # whole tiles only (M = 4800 is not a multiple of 128 either: 37.5)
    (8, 65, 65, 256, 1024),     # 265 tiles on 64 streams: 4-5 tiles per stream (steady-state vmcnt counts)
    (16, 129, 129, 64, 256),    # K = 64 at depth: 2081 tiles on 256 streams
    (1, 9, 9, 256, 512),        # fewer rows than one tile: most workgroups have nothing to do
    (2, 33, 33, 512, 2048),     # K = 512 (layer4 conv3): eight stages per tile, one tile of look-ahead
    (8, 65, 65, 512, 2048),     # ... with 8-9 tiles per stream
]


@pytest.mark.parametrize("case", FWD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_ws_forward_stats_vs_cpu(case):
    n, h, w, cin, cout = case
    m = n * h * w
    groups_ok = m % 2 == 0 and m // 2 >= 128
    assert ws_applies(m, cin, cout, stats=1 if groups_ok else 0) == 1
    e_y, e_dx, e_dw, e_s, prof = run_case(*case, tap=False)
    print(f"{case}: fwd {e_y:.2e} dgrad {e_dx:.2e} wgrad {e_dw:.2e} stats {e_s}; launches fwd big/other {prof[5][1]:.0f}/{prof[0][1]:.0f}")
    assert e_y < 2e-2 and e_dx < 2e-2 and e_dw < 2e-2
    # conv_ws covers every row in ONE launch (the 256x256 kernels split a bench-size M into main + leftover launches)
    assert prof[5][1] == 1 and prof[0][1] == 0, prof
    if groups_ok:
        assert e_s is not None and e_s[0] < 2e-3 and e_s[1] < 2e-3, e_s


# backward: forward conv cin -> cout whose data gradient is the GEMM K = cout -> N = cin on conv_ws: N, H, W, Cin, Cout, tap
BWD_CASES = [
    (2, 19, 23, 1024, 256, False),
    (2, 19, 23, 1024, 256, True),
    (4, 17, 17, 256, 64, True),
    (2, 33, 33, 512, 128, True),
    (8, 65, 65, 1024, 256, True),      # 4-5 tiles per stream: the addend requests are part of the counted vmcnt
    (8, 65, 65, 512, 128, False),
    (32, 65, 65, 1024, 256, True),     # the bench's launch shape (layer3 conv1 backward with the residual gradient)
]


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_ws_dgrad_and_addend_vs_cpu(case):
    n, h, w, cin, cout, tap = case
    m = n * h * w
    assert ws_applies(m, cout, cin, addend=1 if tap else 0) == 1
    e_y, e_dx, e_dw, e_s, prof = run_case(n, h, w, cin, cout, tap=tap)
    print(f"{case}: fwd {e_y:.2e} dgrad {e_dx:.2e} wgrad {e_dw:.2e}; launches dgrad big/other {prof[6][1]:.0f}/{prof[1][1]:.0f}")
    assert e_y < 2e-2 and e_dx < 2e-2 and e_dw < 2e-2
    assert prof[6][1] == 1 and prof[1][1] == 0, prof


def test_ws_shape_rules():
    """What the kernel does NOT take stays on the 256x256 kernels."""
    assert ws_applies(135200, 256, 1024) == 1
    assert ws_applies(135200, 512, 2048) == 1 and ws_applies(135200, 512, 2048, stats=1) == 1
    assert ws_applies(135200, 512, 2048, addend=1) == 0        # K = 512: 128 registers of weights leave no room for the addend
    assert ws_applies(135200, 1024, 2048) == 0
    assert ws_applies(135200, 256, 304) == 0         # not whole panels
    assert ws_applies(135200, 256, 128) == 0
    assert ws_applies(135200, 192, 1024) == 0
    assert ws_applies(135200, 256, 1024, stats=1, addend=1) == 0
    from css_amd import _lib
    n_cu = _lib.query("css_device_cu_count", 0)
    assert _lib.query("css_conv_ws_applies", 135200, 256, 256, 1024, 1024, 3, 3, 1, 1, 0, 0, 0, 0, 1, n_cu) == 0    # 3x3
    assert _lib.query("css_conv_ws_applies", 135200, 256, 256, 1024, 1024, 1, 1, 2, 0, 0, 0, 0, 0, 1, n_cu) == 0    # stride 2
    assert _lib.query("css_conv_ws_applies", 135200, 256, 256, 1024, 1024, 1, 1, 1, 0, 0, 0, 0, 1, 1, n_cu) == 0    # bias
    assert _lib.query("css_conv_ws_applies", 135200, 256, 256, 1024, 1024, 1, 1, 1, 0, 0, 0, 0, 0, 0, n_cu) == 0    # fp32


WORKER = r'''
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_conv_ws_gpu as t
assert t.ws_applies(8 * 65 * 65, 256, 1024) == 0, "CSS_NO_WS_CONV must switch the kernel off"
e_y, e_dx, e_dw, e_s, prof = t.run_case(8, 65, 65, 256, 1024, tap=False)
print("ERRS", e_y, e_dx, e_dw, e_s)
assert e_y < 2e-2 and e_dx < 2e-2 and e_dw < 2e-2 and e_s[0] < 2e-3 and e_s[1] < 2e-3
'''


def test_ws_switch_off_is_a_shipped_configuration():
    e = dict(os.environ)
    e["CSS_NO_WS_CONV"] = "1"
    r = subprocess.run([sys.executable, "-c", WORKER % (ROOT, os.path.join(ROOT, "tests"))], env=e, capture_output=True, text=True, timeout=600)
    print(r.stdout[-400:], r.stderr[-800:])
    assert r.returncode == 0, r.stderr[-800:]
