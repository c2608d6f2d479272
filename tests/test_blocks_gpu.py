"""Block-level forward/backward parity at tight tolerance (shallow chains: the forward error stays at ~1e-6, so no
ReLU mask differs between the two implementations and every gradient must agree to fp32 rounding).  This is what pins
the autograd wiring (residual adds, shared inputs, concat/slice, resize adjoints); the whole-network test then only
has to allow for mask flips."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import assert_close_robust, dev, rel_err, robust_err, to_nchw_cpu, to_nhwc  # noqa: E402

TOL = 1e-4    # forward (max-norm)
GTOL = 4e-3   # gradients (90th percentile): one flipped ReLU reaches every element at the ~1e-4 level through the
              # per-channel means of the batch-norm backward; a wiring bug is O(1) on most elements


def _load(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    module.load_state_dict(sub, strict=True)
    return module.to(dev()).train()


def _check_grads(module, sd, prefix, names):
    worst = 0.0
    named = dict(module.named_parameters())
    for n in names:
        assert_close_robust(named[n[len(prefix):]].grad.cpu(), sd[n].grad, GTOL, n)
    return worst


@pytest.mark.parametrize("li,bi,size", [(1, 0, 17), (1, 1, 17), (2, 0, 17), (3, 0, 9), (3, 5, 9), (4, 0, 9), (4, 2, 9)])
def test_bottleneck_block(li, bi, size):
    from oracle import css_oracle as O
    from css_amd.networks import resnet
    from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
    sd = O.init_state("tv", 21, 256, 7)
    spec = O.backbone_spec("tv")["layers"][li - 1][bi]
    prefix = f"resnet_layer{li}.{bi}."
    names = [n for n in O.param_names("tv", 21, 256) if n.startswith(prefix)]
    for n in names:
        sd[n].requires_grad_(True)
    cin = spec["conv1"]["cin"]
    g = torch.Generator().manual_seed(li * 10 + bi)
    x = torch.randn(3, cin, size, size, generator=g)
    xr = x.clone().requires_grad_(True)
    o = O._bottleneck(sd, spec, xr, True)
    wl = torch.randn(o.shape, generator=g)
    (o * wl).sum().backward()
    net = DeepLabv3Plus_with_rep(resnet.resnet101_tv(), dilate_scale=8, num_classes=21)   # applies _nostride_dilate
    blk = _load(getattr(net, f"resnet_layer{li}")[bi], {k: v.detach() for k, v in sd.items()}, prefix)
    xg = to_nhwc(x, torch.float32).requires_grad_(True)
    og = blk(xg)
    assert rel_err(to_nchw_cpu(og), o.detach()) < TOL
    (og * to_nhwc(wl, torch.float32)).sum().backward()
    assert_close_robust(to_nchw_cpu(xg.grad), xr.grad, GTOL, "dx")
    _check_grads(blk, sd, prefix, names)


def test_stem_maxpool_tv_and_deepstem():
    from oracle import css_oracle as O
    from css_amd import ops
    from css_amd.networks import resnet
    for bb, ctor in (("tv", resnet.resnet101_tv), ("stem", resnet.resnet101)):
        sd = O.init_state(bb, 21, 256, 3)
        spec = O.backbone_spec(bb)
        names = [n for n in O.param_names(bb, 21, 256) if n.startswith("resnet_conv1") or n.startswith("resnet_bn1")]
        for n in names:
            sd[n].requires_grad_(True)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(2, 3, 33, 33, generator=g)
        h = x
        for L in spec["stem"]:
            h = O._apply_conv(sd, L, h) if L["kind"] == "conv" else F.relu(O._apply_bn(sd, L, h, True))
        h = F.relu(O._apply_bn(sd, spec["stem_bn"], h, True))
        o = F.max_pool2d(h, 3, 2, 1, ceil_mode=spec["maxpool_ceil"])
        wl = torch.randn(o.shape, generator=g)
        (o * wl).sum().backward()
        m = ctor()
        m.load_state_dict({k.replace("resnet_", "", 1): v.detach() for k, v in sd.items()
                           if k.startswith("resnet_conv1") or k.startswith("resnet_bn1")}, strict=False)
        m = m.to(dev()).train()
        xg = ops.stage_input(x.to(dev()), torch.float32)
        og = m.maxpool(m.bn1(m.conv1(xg), relu=True))
        assert rel_err(to_nchw_cpu(og), o.detach()) < TOL
        (og * to_nhwc(wl, torch.float32)).sum().backward()
        named = dict(m.named_parameters())
        for n in names:
            assert_close_robust(named[n.replace("resnet_", "", 1)].grad.cpu(), sd[n].grad, GTOL, (bb, n))


def test_aspp_decoder_heads():
    """ASPP (5 branches on a shared input, concat, project) + decoder (project, resize, concat, two heads)."""
    from oracle import css_oracle as O
    from css_amd import ops
    from css_amd.networks import resnet
    from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
    K = 19
    sd = O.init_state("tv", K, 256, 11)
    hs = O.head_spec(K, 256)
    prefixes = ("ASPP.", "project.", "classifier.", "representation.")
    names = [n for n in O.param_names("tv", K, 256) if n.startswith(prefixes)]
    for n in names:
        sd[n].requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    x4 = F.relu(torch.randn(3, 2048, 9, 9, generator=g))
    xl = F.relu(torch.randn(3, 256, 17, 17, generator=g))
    x4r, xlr = x4.clone().requires_grad_(True), xl.clone().requires_grad_(True)

    def cbr(pair, t):
        return F.relu(O._apply_bn(sd, pair[1], O._apply_conv(sd, pair[0], t), True))
    res = [cbr(hs["aspp0"], x4r)] + [cbr(p, x4r) for p in hs["aspp_d"]]
    p = cbr(hs["aspp_pool"], F.adaptive_avg_pool2d(x4r, 1))
    res.append(F.interpolate(p, size=(9, 9), mode="bilinear", align_corners=False))
    feat = cbr(hs["aspp_proj"], torch.cat(res, 1))
    low = cbr(hs["project"], xlr)
    dec = torch.cat([low, F.interpolate(feat, size=(17, 17), mode="bilinear", align_corners=True)], 1)
    outs = []
    for key in ("classifier", "representation"):
        c0, b0, c1 = hs[key]
        outs.append(O._apply_conv(sd, c1, F.relu(O._apply_bn(sd, b0, O._apply_conv(sd, c0, dec), True))))
    wp, wr = torch.randn(outs[0].shape, generator=g), torch.randn(outs[1].shape, generator=g)
    ((outs[0] * wp).sum() + (outs[1] * wr).sum()).backward()

    net = DeepLabv3Plus_with_rep(resnet.resnet101_tv(), dilate_scale=8, num_classes=K)
    net.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
    net = net.to(dev()).train()
    x4g, xlg = to_nhwc(x4, torch.float32).requires_grad_(True), to_nhwc(xl, torch.float32).requires_grad_(True)
    fg = net.ASPP(x4g)
    lg = net.project(xlg)
    dg = ops.cat_channels(lg, ops.bilinear(fg, 17, 17))
    pg, rg = net.classifier(dg), net.representation(dg)
    assert rel_err(to_nchw_cpu(pg), outs[0].detach()) < TOL and rel_err(to_nchw_cpu(rg), outs[1].detach()) < TOL
    ((pg.permute(0, 3, 1, 2) * wp.to(dev())).sum() + (rg.permute(0, 3, 1, 2) * wr.to(dev())).sum()).backward()
    # the pooled branch normalises over only B=3 samples per channel: invstd up to 1/sqrt(eps) amplifies fp32 rounding
    # in its backward (same effect as the N*H*W == 2 case of test_ops_gpu), and its input gradient is added to x4's
    errs = {"x4": robust_err(to_nchw_cpu(x4g.grad), x4r.grad)[0], "xl": robust_err(to_nchw_cpu(xlg.grad), xlr.grad)[0]}
    named = dict(net.named_parameters())
    for n in names:
        errs[n] = robust_err(named[n].grad.cpu(), sd[n].grad)[0]
    print("aspp/decoder grad errs: max", max(errs.values()), {k: f"{v:.1e}" for k, v in errs.items() if v > TOL})
    for n, e in errs.items():
        assert e < (2e-3 if (n == "x4" or "convs.4." in n) else TOL), (n, e)
