"""conv3x3_c64_kernel (css_amd/csrc/conv_c64.hip): the patch-in-LDS kernel of the 3x3 stride-1 pad-1 convolutions on 64 input channels - conv2 of
the layer-1 Bottlenecks (/root/reference/generalframeworks/networks/resnet.py:126-129) forward and in its data-gradient form, and the second / third
convolution of the deep stem (resnet.py:177-190: 64 -> 64, 64 -> 128) - through the C ABI, against torch-CPU fp32 ``F.conv2d`` on the same
bf16-rounded inputs AND against the implicit-GEMM kernels it replaces (css_conv_c64_set_enabled(0)), with the dispatch asserted
(css_conv_c64_applies).

Covers images narrower / shorter than the kernel (W = 1, H = 1, 2 x 2), widths that are not multiples of anything, fewer rows than one tile, tiles
that cross image rows and images (the zero padding comes from a per-lane validity mask, not from the addresses), several tiles per workgroup (the
double-buffered patch, the counted vmcnt), the statistics slabs across a group boundary, both output widths and the bench's launch shapes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import bf16_round, dev, rel_err  # noqa: E402


def applies(n, h, w, cin, cout, r=3, stride=1, pad=1, dil=1, addend=0, bias=0, dtype=1):
    from css_amd import _lib
    return _lib.query("css_conv_c64_applies", n, h, w, cin, cin, cout, cout, r, r, stride, pad, dil, addend, bias, dtype)


def run_case(n, h, w, cout, seed=0, stats=True):
    from css_amd import ops
    g = torch.Generator().manual_seed(977 + cout + n * h * w + seed)
    x = bf16_round(torch.randn(n, h, w, 64, generator=g) + 0.25)
    wt = bf16_round(torch.randn(cout, 3, 3, 64, generator=g) / 24.0)
    gy = bf16_round(torch.randn(n, h, w, cout, generator=g))
    xr = x.permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    (yr * gy.permute(0, 3, 1, 2)).sum().backward()
    y_ref, dx_ref = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)

    def gpu():
        xg = x.to(dev(), torch.bfloat16).requires_grad_(True)
        wg = wt.permute(0, 3, 1, 2).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        m = n * h * w
        groups = 2 if (stats and m % 2 == 0 and m // 2 >= 128) else 1
        with ops.bn_groups(groups):
            y = ops.conv2d(xg, wg, None, 1, 1, 1, bn_stats=stats)
        (y.float() * gy.to(dev())).sum().backward()
        torch.cuda.synchronize()
        st = getattr(y, "_css_bnstats", None)
        sums = None
        if st is not None:
            from css_amd._lib import call, dev_stream
            part, mg, ngr, c_, bm = st
            sums = torch.empty(ngr * 2 * cout + ngr, dtype=torch.float64, device=dev())
            d, s_ = dev_stream(y)
            call("css_bn_reduce_finalize_slabs", part, mg * ngr, mg, ngr, float(mg), None, None, None, None, 0.0, 0.0, None, None, None, None,
                 sums, cout, y, cout, bm, d, s_)
            sums = sums.cpu()[:ngr * 2 * cout].reshape(ngr, 2, cout)
        return y.detach().float().cpu(), xg.grad.float().cpu(), wg.grad.cpu(), sums

    return y_ref, dx_ref, wr.grad, gpu


CASES = [
    (2, 19, 23, 64),        # 7 tiles, ragged last tile, tiles crossing image rows and the image boundary, a group boundary inside a slab
    (1, 9, 9, 64),          # fewer rows than one tile
    (1, 2, 2, 64),          # every tap row of every pixel touches the border
    (3, 1, 50, 64),         # H = 1: no vertical neighbour at all
    (2, 50, 1, 64),         # W = 1: no horizontal neighbour; the range of a tile is 128 different image rows
    (4, 17, 17, 128),       # the deep stem's 64 -> 128
    (3, 40, 40, 128),
    (8, 65, 65, 64),        # 265 tiles: one or two per workgroup
    (2, 129, 129, 64),      # layer 1 at the bench's resolution
    (4, 193, 193, 64),      # 1164 tiles on 256 workgroups: 4-5 tiles each (the double buffer, the counted wait in steady state)
    (1, 131, 257, 128),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_c64_forward_dgrad_stats_vs_cpu_and_vs_gather_kernels(case):
    from css_amd import _lib
    n, h, w, cout = case
    assert applies(n, h, w, 64, cout) == 1
    y_ref, dx_ref, dw_ref, gpu = run_case(*case)
    y, dx, dw, sums = gpu()
    was = _lib.query("css_conv_c64_set_enabled", 0)
    try:
        assert applies(n, h, w, 64, cout) == 0
        y0, dx0, dw0, sums0 = gpu()
    finally:
        _lib.query("css_conv_c64_set_enabled", was)
    e_y, e_dx, e_dw = rel_err(y, y_ref), rel_err(dx, dx_ref), rel_err(dw, dw_ref)
    d_y, d_dx = (y - y0).abs().max().item(), (dx - dx0).abs().max().item() if cout == 64 else 0.0
    print(f"{case}: vs cpu fwd {e_y:.2e} dgrad {e_dx:.2e} wgrad {e_dw:.2e}; vs gather kernels fwd max |d| {d_y:.3e} ({(y != y0).float().mean().item():.2e} of the "
          f"elements differ) dgrad {d_dx:.3e}")
    assert e_y < 2e-2 and e_dx < 2e-2 and e_dw < 2e-2
    # same K order, same instruction: bit-identical to the gather kernels, forward and data gradient (64 -> 128: its gradient gathers 128 channels
    # and stays on the gather kernels either way)
    assert torch.equal(y, y0) and torch.equal(dx, dx0)
    if sums is not None:
        yy = y_ref.double().reshape(sums.shape[0], -1, cout)
        want = torch.stack([yy.sum(1), (yy * yy).sum(1)], 1)
        e_s = ((sums[:, 0] - want[:, 0]).abs().max().item() / yy.abs().sum(1).max().item(), rel_err(sums[:, 1], want[:, 1]))
        e_s0 = ((sums[:, 0] - sums0[:, 0]).abs().max().item() / yy.abs().sum(1).max().item(), rel_err(sums[:, 1], sums0[:, 1]))
        print(f"   statistics vs cpu {e_s}, vs gather kernels {e_s0}")
        assert e_s[0] < 2e-3 and e_s[1] < 2e-3, e_s
        assert e_s0[0] < 1e-6 and e_s0[1] < 1e-6, e_s0


def test_c64_dispatch_rule():
    """What the kernel takes: 3x3, stride 1, pad 1, dilation 1, 64 gathered channels, 64 or 128 result channels, bf16, no addend, no bias."""
    assert applies(32, 129, 129, 64, 64) == 1 and applies(16, 385, 385, 64, 128) == 1
    assert applies(32, 129, 129, 64, 256) == 0 and applies(32, 129, 129, 128, 64) == 0 and applies(32, 129, 129, 64, 32) == 0
    assert applies(32, 129, 129, 64, 64, r=1, pad=0) == 0 and applies(32, 129, 129, 64, 64, stride=2) == 0
    assert applies(32, 129, 129, 64, 64, pad=2, dil=2) == 0 and applies(32, 129, 129, 64, 64, addend=1) == 0
    assert applies(32, 129, 129, 64, 64, bias=1) == 0 and applies(32, 129, 129, 64, 64, dtype=0) == 0


def test_c64_no_statistics_and_repeatability():
    """The plain form (the teacher's forward pass has no statistics epilogue) and run-to-run bit identity."""
    case = (4, 65, 65, 64)
    y_ref, dx_ref, dw_ref, gpu = run_case(*case, stats=False)
    y1, dx1, _, s1 = gpu()
    y2, dx2, _, s2 = gpu()
    assert s1 is None
    assert rel_err(y1, y_ref) < 2e-2 and rel_err(dx1, dx_ref) < 2e-2
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)


def test_c64_row_pitches_through_the_abi():
    """The gathered tensor and the result may be slices of wider rows (a concat buffer, a padded channel count): css_conv2d_forward straight
    through the C ABI with ld_src = 96 and ld_dst = 80 against the contiguous call - bit for bit - and nothing written outside the 64 channels."""
    from css_amd import _lib
    from css_amd._lib import call, dev_stream
    n, h, w = 2, 37, 41
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(n, h, w, 64, generator=g))
    wt = bf16_round(torch.randn(64, 3, 3, 64, generator=g) / 24.0)
    xw = torch.full((n, h, w, 96), 7.0).to(dev(), torch.bfloat16)
    xw[..., :64] = x.to(dev(), torch.bfloat16)
    xc = x.to(dev(), torch.bfloat16).contiguous()
    wg = wt.to(dev(), torch.bfloat16).contiguous()            # [Cout][R][S][Cin]: the forward layout
    yw = torch.full((n, h, w, 80), -3.0, device=dev(), dtype=torch.bfloat16)
    yc = torch.empty((n, h, w, 64), device=dev(), dtype=torch.bfloat16)
    d, st = dev_stream(xc)
    assert _lib.query("css_conv_c64_applies", n, h, w, 64, 96, 64, 80, 3, 3, 1, 1, 1, 0, 0, 1) == 1
    call("css_conv2d_forward", xw, wg, None, yw, n, h, w, 64, 96, h, w, 64, 80, 3, 3, 1, 1, 1, 0.0, 1, d, st)
    call("css_conv2d_forward", xc, wg, None, yc, n, h, w, 64, 64, h, w, 64, 64, 3, 3, 1, 1, 1, 0.0, 1, d, st)
    torch.cuda.synchronize()
    assert torch.equal(yw[..., :64], yc)
    assert bool((yw[..., 64:] == -3.0).all())
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_err(yc.float().cpu(), ref) < 2e-2
