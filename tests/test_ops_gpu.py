"""Every HIP op through the C ABI against plain torch-CPU fp32 math on the same seeded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import TOL, bf16_round, dev, rel_err, to_nchw_cpu, to_nhwc  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]

CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, bias
    (2, 17, 17, 64, 64, 1, 1, 0, 1, False),
    (2, 17, 17, 256, 128, 1, 1, 0, 1, False),
    (1, 33, 33, 64, 64, 3, 1, 1, 1, False),
    (2, 17, 17, 64, 128, 3, 2, 1, 1, False),
    (2, 9, 9, 256, 256, 3, 1, 2, 2, False),
    (2, 9, 9, 128, 64, 3, 1, 4, 4, False),
    (2, 9, 9, 256, 256, 3, 1, 12, 12, False),
    (2, 17, 17, 256, 512, 1, 2, 0, 1, False),
    (2, 65, 65, 3, 64, 7, 2, 3, 1, False),
    (2, 33, 33, 3, 64, 3, 2, 1, 1, False),
    (2, 17, 17, 304, 256, 3, 1, 1, 1, False),
    (2, 17, 17, 256, 21, 1, 1, 0, 1, True),
    (2, 17, 17, 256, 48, 1, 1, 0, 1, False),
    (3, 13, 11, 64, 320, 3, 1, 1, 1, True),
    (1, 1, 1, 2048, 256, 1, 1, 0, 1, False),
    (1, 65, 65, 128, 128, 3, 1, 24, 24, False),   # ASPP-like: tiles near the top / bottom skip all-padding kernel rows
    (2, 33, 33, 64, 128, 3, 1, 36, 36, False),    # dilation larger than the map: only the centre row of the kernel survives
    (3, 20, 12, 128, 192, 3, 1, 6, 6, False),     # tiles spanning two images
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case, dtype):
    from css_amd import ops
    n, h, w, cin, cout, k, stride, pad, dil, bias = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    if dtype == torch.bfloat16:
        x, wt = bf16_round(x), bf16_round(wt)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = bf16_round(gy)
    yr.backward(gy)

    v = ops.vec_of(dtype)
    xg = to_nhwc(x, dtype, ops.pad_to(cin, v)).requires_grad_(cin % v == 0)
    wg = wt.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(dev()).requires_grad_(True) if bias else None
    y = ops.conv2d(xg, wg, bg, stride, pad, dil)
    tol = TOL[dtype]
    assert rel_err(to_nchw_cpu(y), yr.detach()) < tol
    y.backward(to_nhwc(gy, dtype))
    if xg.requires_grad:
        assert rel_err(to_nchw_cpu(xg.grad), xr.grad) < tol
    assert rel_err(wg.grad.cpu(), wr.grad) < tol
    if bias:
        assert rel_err(bg.grad.cpu(), br.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,res,relu", [((2, 17, 17, 64), False, True), ((2, 9, 9, 256), True, True),
                                            ((2, 1, 1, 256), False, True), ((3, 13, 7, 48), False, False),
                                            ((2, 33, 33, 2048), True, True), ((4, 5, 5, 304), False, True)])
def test_bn_act_train(shape, res, relu, dtype):
    from css_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    r = torch.randn(n, c, h, w, generator=g) if res else None
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    if dtype == torch.bfloat16:
        x = bf16_round(x)
        r = bf16_round(r) if res else None
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    o = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if res:
        o = o + rr
    if relu:
        o = F.relu(o)
    go = torch.randn(o.shape, generator=g)
    if dtype == torch.bfloat16:
        go = bf16_round(go)
    o.backward(go)

    xg = to_nhwc(x, dtype).requires_grad_(True)
    rg = to_nhwc(r, dtype).requires_grad_(True) if res else None
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    rmg, rvg = rm.to(dev()), rv.to(dev())
    og = ops.bn_act(xg, gg, bg, rmg, rvg, rg, relu, True, 0.1, 1e-5, False)
    tol = TOL[dtype]
    assert rel_err(to_nchw_cpu(og), o.detach()) < tol
    assert rel_err(rmg.cpu(), rm_r) < 1e-5 and rel_err(rvg.cpu(), rv_r) < 1e-5
    og.backward(to_nhwc(go, dtype))
    # two samples per channel: invstd up to 1/sqrt(eps) amplifies the rounding of the bracketed difference
    btol = max(1e-3, tol * 5) if n * h * w == 2 else tol * 5
    # N*H*W == 2 makes d/dx cancel analytically (only eps-sized residue left): compare on the incoming-gradient scale
    floor = 1e-3 * go.abs().max()
    assert ((to_nchw_cpu(xg.grad) - xr.grad).abs().max() / max(xr.grad.abs().max(), floor)).item() < btol
    assert rel_err(gg.grad.cpu(), gr.grad) < btol and rel_err(bg.grad.cpu(), br.grad) < btol
    if res:
        assert rel_err(to_nchw_cpu(rg.grad), rr.grad) < btol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,groups", [((8, 65, 65, 1024), 2), ((3, 11, 7, 264), 1)])
def test_bn_residual_bit_mask_equals_activation_mask(shape, groups, dtype):
    """Residual layers (bn3 + identity + ReLU, /root/reference/generalframeworks/networks/resnet.py:133-137) hand the ReLU mask to their
    backward passes as one byte per 16-byte vector (css_bn_apply_mask / css_bn_bwd_*_mask); the round-2 form (CSS_BN_NO_MASK=1) re-reads the
    activation tensor.  Same arithmetic: bit-identical outputs and gradients, here at a bench-scale layer (8 x 65^2 x 1024, two statistics
    groups) and at a ragged one (C = 264: 33 vectors per row)."""
    from css_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(11 + c)
    x = (torch.randn(n, h, w, c, generator=g) * 2 + 0.5).to(dev(), dtype)
    r = torch.randn(n, h, w, c, generator=g).to(dev(), dtype)
    go = torch.randn(n, h, w, c, generator=g).to(dev(), dtype)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev()), (torch.randn(c, generator=g) * 0.1).to(dev())
    outs = []
    prev = ops._bn_bit_mask
    try:
        for use_mask in (True, False):
            ops._bn_bit_mask = use_mask
            xg, rg = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
            gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            rm, rv = torch.zeros(c, device=dev()), torch.ones(c, device=dev())
            o = ops.bn_act(xg, gg, bg, rm, rv, rg, True, True, 0.1, 1e-5, False, groups=groups)
            o.backward(go)
            outs.append((o.detach(), xg.grad, rg.grad, gg.grad, bg.grad, rm, rv))
    finally:
        ops._bn_bit_mask = prev
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert (outs[0][2] == 0).float().mean().item() > 0.2      # the mask does something: a fair share of the residual gradient is zeroed


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_eval(dtype):
    from css_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 9, 9, generator=g)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    o = F.relu(F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5))
    og = ops.bn_act(to_nhwc(x, dtype), gamma.to(dev()), beta.to(dev()), rm.to(dev()), rv.to(dev()), None, True, False)
    assert rel_err(to_nchw_cpu(og), o) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ceil", [((2, 33, 33, 64), False), ((2, 33, 33, 128), True), ((1, 18, 20, 64), True),
                                        ((2, 17, 17, 64), False)])
def test_maxpool(shape, ceil, dtype):
    from css_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h)
    x = F.relu(torch.randn(n, c, h, w, generator=g))     # ReLU output: many exact ties at 0
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    xr = x.clone().requires_grad_(True)
    o = F.max_pool2d(xr, 3, 2, 1, ceil_mode=ceil)
    go = torch.randn(o.shape, generator=g)
    if dtype == torch.bfloat16:
        go = bf16_round(go)
    o.backward(go)
    xg = to_nhwc(x, dtype).requires_grad_(True)
    og = ops.maxpool(xg, 3, 2, 1, ceil)
    assert og.shape[1:3] == o.shape[2:]
    assert rel_err(to_nchw_cpu(og), o.detach()) == 0
    og.backward(to_nhwc(go, dtype))
    assert rel_err(to_nchw_cpu(xg.grad), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ceil,groups", [((4, 33, 33, 64), False, 2), ((2, 34, 31, 128), True, 1), ((2, 129, 129, 64), False, 2),
                                               ((2, 97, 97, 128), True, 2)])
def test_bn_relu_maxpool_fused_equals_the_two_passes(shape, ceil, groups, dtype):
    """The stem's batch norm + ReLU + 3x3 s2 p1 max pool in ONE pass (css_bn_apply_maxpool, round 5: the normalised activation is never written;
    /root/reference/generalframeworks/networks/resnet.py:186-190) against bn_act followed by maxpool: pooled values, running statistics and - through
    the arg-max bytes - every gradient BIT-IDENTICAL (ReLU output: many exact ties at 0, first maximum wins on both paths)."""
    from css_amd import ops
    from css_amd.nn import HipBatchNorm2d, HipMaxPool2d
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h * 7 + c)
    x = torch.randn(n, h, w, c, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    res = []
    for fused in (True, False):
        bn = HipBatchNorm2d(c).to(dev()).train()
        pool = HipMaxPool2d(3, 2, 1, ceil_mode=ceil)
        with torch.no_grad():
            bn.weight.copy_(gamma.to(dev())); bn.bias.copy_(beta.to(dev()))
        xg = x.to(dev()).to(dtype).requires_grad_(True)
        with ops.bn_groups(groups):
            out = bn(xg, relu=True, pool=pool) if fused else pool(bn(xg, relu=True))
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(dev()).to(dtype)
        out.backward(go)
        res.append((out.detach().float().cpu(), xg.grad.float().cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(), bn.running_mean.cpu(), bn.running_var.cpu()))
    assert res[0][0].shape == res[1][0].shape and res[0][0].shape[1] == ops.pool_out_size(h, 3, 2, 1, ceil)
    for a, b, name in zip(res[0], res[1], ("pooled", "dx", "dgamma", "dbeta", "running_mean", "running_var")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    # and the pooled tensor against torch-CPU on the same operands
    xc = x.to(dtype).float().permute(0, 3, 1, 2)
    parts = []
    for gi in range(groups):
        part = xc[gi * n // groups:(gi + 1) * n // groups]
        m_, v_ = part.mean((0, 2, 3), keepdim=True), part.var((0, 2, 3), unbiased=False, keepdim=True)
        parts.append(F.relu((part - m_) / torch.sqrt(v_ + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)))
    ref = F.max_pool2d(torch.cat(parts), 3, 2, 1, ceil_mode=ceil)
    assert rel_err(res[0][0].permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hs,hd,c", [(17, 65, 21), (9, 17, 256), (33, 129, 19), (5, 5, 8), (1, 4, 8), (7, 10, 16)])
def test_bilinear(hs, hd, c, dtype):
    from css_amd import ops
    g = torch.Generator().manual_seed(hs * hd)
    x = torch.randn(2, c, hs, hs + 2, generator=g)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    xr = x.clone().requires_grad_(True)
    o = F.interpolate(xr, size=(hd, hd + 3), mode="bilinear", align_corners=True)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    xg = to_nhwc(x, dtype).requires_grad_(True)
    og = ops.bilinear(xg, hd, hd + 3, torch.float32)
    assert rel_err(to_nchw_cpu(og), o.detach()) < 2e-5 if dtype == torch.float32 else 1e-2
    og.backward(to_nhwc(go, torch.float32))
    assert rel_err(to_nchw_cpu(xg.grad), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_broadcast_cat(dtype):
    from css_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 9, 11, generator=g)
    y = torch.randn(2, 48, 9, 11, generator=g)
    if dtype == torch.bfloat16:
        x, y = bf16_round(x), bf16_round(y)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    p = F.adaptive_avg_pool2d(xr, 1)
    bb = F.interpolate(p, size=(9, 11), mode="bilinear", align_corners=False)
    o = torch.cat([yr, bb, xr], 1)
    go = torch.randn(o.shape, generator=g)
    if dtype == torch.bfloat16:
        go = bf16_round(go)
    o.backward(go)
    xg, yg = to_nhwc(x, dtype).requires_grad_(True), to_nhwc(y, dtype).requires_grad_(True)
    pg = ops.global_avg_pool(xg)
    og = ops.cat_channels(yg, ops.broadcast_hw(pg, 9, 11), xg)
    tol = TOL[dtype]
    assert rel_err(to_nchw_cpu(og), o.detach()) < tol
    og.backward(to_nhwc(go, dtype))
    assert rel_err(to_nchw_cpu(xg.grad), xr.grad) < tol and rel_err(to_nchw_cpu(yg.grad), yr.grad) < tol


def test_stage_input():
    from css_amd import ops
    x = torch.randn(2, 3, 9, 7)
    for dt, cp in ((torch.float32, 4), (torch.bfloat16, 8)):
        o = ops.stage_input(x.to(dev()), dt)
        assert o.shape == (2, 9, 7, cp)
        ref = x.permute(0, 2, 3, 1)
        assert rel_err(o[..., :3].float().cpu(), ref if dt == torch.float32 else bf16_round(ref)) == 0
        assert o[..., 3:].abs().max() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_groups_equal_separate_passes(dtype):
    """G forward passes batched along dim 0 with per-group statistics == G separate calls (values, running stats, grads)."""
    from css_amd import ops
    g = torch.Generator().manual_seed(9)
    c = 64
    xs = [torch.randn(2, c, 9, 7, generator=g) * (1 + i) + i for i in range(2)]
    rs = [torch.randn(2, c, 9, 7, generator=g) for _ in range(2)]
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    gos = [torch.randn(2, c, 9, 7, generator=g) for _ in range(2)]
    if dtype == torch.bfloat16:
        xs, rs, gos = [bf16_round(t) for t in xs], [bf16_round(t) for t in rs], [bf16_round(t) for t in gos]
    # reference: two sequential F.batch_norm calls sharing parameters and running stats
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    xr = [t.clone().requires_grad_(True) for t in xs]
    outs = []
    for i in range(2):
        outs.append(F.relu(F.batch_norm(xr[i], rm_r, rv_r, gr, br, True, 0.1, 1e-5) + rs[i]))
    (sum((o * go).sum() for o, go in zip(outs, gos))).backward()
    xg = to_nhwc(torch.cat(xs), dtype).requires_grad_(True)
    rg = to_nhwc(torch.cat(rs), dtype)
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    rmg, rvg = rm.to(dev()), rv.to(dev())
    og = ops.bn_act(xg, gg, bg, rmg, rvg, rg, True, True, 0.1, 1e-5, False, groups=2)
    tol = TOL[dtype]
    assert rel_err(to_nchw_cpu(og), torch.cat(outs).detach()) < tol
    assert rel_err(rmg.cpu(), rm_r) < 1e-5 and rel_err(rvg.cpu(), rv_r) < 1e-5
    og.backward(to_nhwc(torch.cat(gos), dtype))
    assert rel_err(to_nchw_cpu(xg.grad), torch.cat([t.grad for t in xr])) < tol * 5
    assert rel_err(gg.grad.cpu(), gr.grad) < tol * 5 and rel_err(bg.grad.cpu(), br.grad) < tol * 5


# ---- fused epilogues --------------------------------------------------------------------------------------------------
FUSED_STAT_CASES = [
    # N, H, W, Cin, Cout, k, pad, dil, groups   (M/groups >= 128; group boundaries mostly NOT multiples of 128)
    (2, 17, 17, 64, 64, 1, 0, 1, 1),       # 128x64 kernel, one group, ragged last slab
    (2, 17, 17, 64, 64, 1, 0, 1, 2),       # boundary at row 289: slab 2 straddles
    (4, 33, 33, 128, 256, 1, 0, 1, 2),     # 128x128 kernel only (less than one round of big tiles)
    (32, 65, 65, 64, 256, 1, 0, 1, 2),     # 256x128 DMA kernel: whole rounds + 128x128 remainder, boundary at 67600
    (16, 65, 65, 256, 128, 3, 2, 2, 2),    # dilated 3x3 through the DMA kernel
    (6, 23, 19, 64, 48, 3, 1, 1, 3),       # three groups, Cout not a multiple of the tile
]


@pytest.mark.parametrize("case", FUSED_STAT_CASES)
def test_conv_epilogue_bn_statistics(case):
    """conv + train-mode BN with the statistics from the convolution epilogue == the same with a separate bn_stats pass
    (both read the bf16-rounded output; only the summation order differs)."""
    from css_amd import ops
    n, h, w, cin, cout, k, pad, dil, groups = case
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(n, h, w, cin, generator=g) + 0.5).to(dev(), torch.bfloat16)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev()).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(dev()), torch.randn(cout, generator=g).to(dev())
    outs = []
    for fused in (True, False):
        rm, rv = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev())
        with ops.bn_groups(groups):
            y = ops.conv2d(x, wt, None, 1, pad, dil, bn_stats=fused)
            assert hasattr(y, "_css_bnstats") == fused
            a = ops.bn_act(y, gamma, beta, rm, rv, None, True, True, 0.1, 1e-5, False)
        outs.append((y.float().cpu(), a.float().cpu(), rm.cpu(), rv.cpu()))
    (y1, a1, rm1, rv1), (y0, a0, rm0, rv0) = outs
    assert torch.equal(y1, y0)
    assert rel_err(rm1, rm0) < 1e-5 and rel_err(rv1, rv0) < 1e-5
    # bf16 outputs: the two paths' batch statistics differ in the last fp32 bit, which flips the rounding of the odd
    # element; one bf16 ulp at the largest value is 2^-8 of it
    assert rel_err(a1, a0) < 5e-3
    assert (a1 != a0).float().mean() < 1e-3
    # and against an fp64 restatement of per-group statistics
    yy = y0.double().reshape(groups, -1, cout)
    mean = yy.mean(1)
    var = yy.var(1, unbiased=True)
    rm_ref, rv_ref = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    for gi in range(groups):
        rm_ref = 0.9 * rm_ref + 0.1 * mean[gi]
        rv_ref = 0.9 * rv_ref + 0.1 * var[gi]
    assert rel_err(rm1.double(), rm_ref) < 1e-5 and rel_err(rv1.double(), rv_ref) < 1e-5
    # SyncBN form of stage 2: raw per-group (sum, sum of squares) in fp64, the numbers that get all-reduced across ranks
    from css_amd._lib import call, dev_stream
    with ops.bn_groups(groups):
        y = ops.conv2d(x, wt, None, 1, pad, dil, bn_stats=True)
    part, mg, g_, c_, bm = y._css_bnstats
    sums = torch.empty(groups * 2 * cout + groups, dtype=torch.float64, device=dev())   # [G][2][C] sums + [G] row counts
    d, st = dev_stream(y)
    call("css_bn_reduce_finalize_slabs", part, mg * groups, mg, groups, float(mg), None, None, None, None, 0.0, 0.0, None, None, None, None,
         sums, cout, y, cout, bm, d, st)
    want = torch.stack([yy.sum(1), (yy * yy).sum(1)], 1).reshape(-1)
    assert rel_err(sums.cpu()[:groups * 2 * cout], want) < 1e-6
    assert sums.cpu()[groups * 2 * cout:].tolist() == [float(mg)] * groups        # this rank's rows per group ride behind the sums


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 17, 17, 64, 256, 1, 0, 1), (8, 65, 65, 128, 128, 3, 2, 2), (2, 9, 9, 256, 64, 3, 1, 1)])
def test_conv_tap_folds_residual_gradient(shape, dtype):
    """y, x_id = conv2d(x, tap=True): the gradient reaching x is dgrad(dy) + d(x_id), the sum done in the dgrad store."""
    from css_amd import ops
    n, h, w, cin, cout, k, pad, dil = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, cin, generator=g).to(dev(), dtype)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev()).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, h, w, cout, generator=g).to(dev(), dtype)
    gid = torch.randn(n, h, w, cin, generator=g).to(dev(), dtype)
    grads = []
    for tap in (True, False):
        xg = x.clone().requires_grad_(True)
        wg = wt.clone().requires_grad_(True)
        if tap:
            y, xid = ops.conv2d(xg, wg, None, 1, pad, dil, tap=True)
            assert xid.data_ptr() == xg.data_ptr()
        else:
            y, xid = ops.conv2d(xg, wg, None, 1, pad, dil), xg
        torch.autograd.backward([y, xid], [gy, gid])
        grads.append((xg.grad.float().cpu(), wg.grad.float().cpu()))
    assert rel_err(grads[0][1], grads[1][1]) < 1e-6
    # same arithmetic (sum in fp32, one rounding) as autograd's separate add
    assert rel_err(grads[0][0], grads[1][0]) < (1e-6 if dtype == torch.float32 else 2e-3)


def test_bulk_weight_preparation_matches_per_layer():
    """ops.prepare_flat_weights (one cast + one batched transpose launch) == the per-layer css_weight_layout results."""
    import torch.nn as nn
    from css_amd import ops
    from css_amd.networks.ddp_model import flatten_parameters
    from css_amd.nn import HipBatchNorm2d, HipConv2d
    torch.manual_seed(3)
    net = nn.Sequential(HipConv2d(3, 64, 7, 2, 3, bias=False), HipBatchNorm2d(64), HipConv2d(64, 72, 3, 1, 1, bias=False),
                        HipConv2d(72, 256, 1, bias=False), HipConv2d(256, 21, 1, bias=True), HipConv2d(256, 40, 3, 1, 2, 2, bias=False)).to(dev())
    flatten_parameters(net)
    convs = [m for m in net if isinstance(m, HipConv2d)]
    ref = {}
    for i, m in enumerate(convs):
        cin = m.in_channels
        if cin % 8 == 0:
            ref[(i, False)] = ops.prepared_weight(m.weight, torch.bfloat16, cin, False).clone()
            if m.out_channels % 8 == 0:
                ref[(i, True)] = ops.prepared_weight(m.weight, torch.bfloat16, cin, True).clone()
    ops.invalidate_weight_cache()
    ops.prepare_flat_weights(net, torch.bfloat16, dgrad=True)
    for (i, dg), want in ref.items():
        m = convs[i]
        stamp, got = m.weight.__dict__["_css_wcache"][(torch.bfloat16, m.in_channels, dg)]
        assert got.data_ptr() != want.data_ptr() and got.shape == want.shape
        assert torch.equal(got, want), (i, dg)
        assert ops.prepared_weight(m.weight, torch.bfloat16, m.in_channels, dg).data_ptr() == got.data_ptr()   # cache hit

