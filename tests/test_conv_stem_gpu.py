"""The stride-2 stem convolution on the space-to-depth image (css_amd/csrc/conv_stem.hip; VERDICT r02-r04 "a streaming kernel for the stems"):
torchvision's conv1 = Conv2d(3, 64, 7, 2, 3) behind models.resnet101() (/root/reference/mix_label.py:68) and ResNet_Stem.conv1[0] = conv3x3(3, 64,
stride 2) (/root/reference/generalframeworks/networks/resnet.py:177-190).  Against torch-CPU on the bf16-rounded operands: the staging (exact),
the rearranged weights (exact), the convolution at ragged / even / odd / bench sizes (bf16 rounding of the output), the batch-norm statistics the
epilogue emits (through the batch norm that consumes them, two statistics groups), the weight gradient computed in s2d space and folded back, and
the dispatch (what the network's `stage` returns; CSS_NO_STEM_S2D=1 in a process of its own)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev, rel_err  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("n,h,w", [(2, 65, 65), (3, 64, 66), (1, 7, 9), (2, 129, 131)])
def test_s2d_staging_is_exact(n, h, w):
    from css_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn(n, 3, h, w, generator=g)
    s = ops.stage_inputs([x.to(dev())], torch.bfloat16, s2d=True)
    hs, ws = (h + 1) // 2, (w + 1) // 2
    assert isinstance(s, ops.S2DInput) and tuple(s.t.shape) == (n, hs, ws, 16) and s.hw == (h, w)
    want = torch.zeros(n, hs, ws, 16)
    xp = F.pad(_bf(x), (0, 2 * ws - w, 0, 2 * hs - h))
    for py in range(2):
        for px in range(2):
            for c in range(3):
                want[..., (2 * py + px) * 3 + c] = xp[:, c, py::2, px::2]
    assert torch.equal(s.t.float().cpu(), want)


@pytest.mark.parametrize("r,n,h,w,groups", [(7, 2, 65, 65, 2), (7, 3, 64, 66, 1), (7, 2, 129, 131, 2), (7, 1, 23, 17, 1), (3, 2, 97, 97, 2),
                                            (3, 3, 66, 64, 1), (7, 4, 513, 513, 2), (3, 2, 769, 769, 2)])
def test_stem_s2d_forward_stats_and_wgrad_vs_cpu(r, n, h, w, groups):
    from css_amd import ops
    from css_amd.nn import HipBatchNorm2d, HipConv2d
    g = torch.Generator().manual_seed(r * 100 + h)
    x = torch.randn(n, 3, h, w, generator=g)
    conv = HipConv2d(3, 64, r, 2, r // 2, bias=False).to(dev())
    with torch.no_grad():
        conv.weight.copy_((torch.randn(64, 3, r, r, generator=g) * 0.1).to(dev()))
    assert ops.stem_s2d_ok(conv, torch.bfloat16) and not ops.stem_s2d_ok(conv, torch.float32)
    bn = HipBatchNorm2d(64).to(dev())
    conv.train(); bn.train()
    xs = ops.stage_inputs([x.to(dev())], torch.bfloat16, s2d=True)
    with ops.bn_groups(groups):
        y = conv(xs)
        out = bn(y, relu=True)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    assert tuple(y.shape) == (n, ho, wo, 64) and y.dtype == torch.bfloat16
    wq = _bf(conv.weight.detach().float().cpu())
    ref = F.conv2d(_bf(x), wq, None, 2, r // 2)
    assert ref.shape[2:] == (ho, wo)
    e = rel_err(y.float().cpu().permute(0, 3, 1, 2), ref)
    print(f"stem s2d {r}x{r} {n}x{h}x{w}: forward rel err {e:.2e}")
    assert e < 1e-2
    # batch norm on the bf16-rounded output, per statistics group (what css_bn_reduce_finalize_slabs made of the epilogue's slabs)
    yq = y.float().cpu().permute(0, 3, 1, 2)
    outs = []
    for gi in range(groups):
        part = yq[gi * n // groups:(gi + 1) * n // groups]
        m_, v_ = part.mean((0, 2, 3), keepdim=True), part.var((0, 2, 3), unbiased=False, keepdim=True)
        outs.append(torch.relu((part - m_) / torch.sqrt(v_ + 1e-5)))
    e_bn = rel_err(out.float().cpu().permute(0, 3, 1, 2), torch.cat(outs))
    print(f"   batch norm from the fused statistics ({groups} groups): rel err {e_bn:.2e}")
    assert e_bn < 2e-2
    # weight gradient (s2d space, folded back) against torch-CPU on the same bf16-rounded operands
    gy = torch.randn(y.shape, generator=g) * 0.1
    conv.weight.grad = None
    y.backward(gy.to(dev()).to(torch.bfloat16))
    wr = wq.clone().requires_grad_(True)
    F.conv2d(_bf(x), wr, None, 2, r // 2).backward(_bf(gy).permute(0, 3, 1, 2))
    e_w = rel_err(conv.weight.grad.float().cpu(), wr.grad)
    print(f"   weight gradient rel err {e_w:.2e}")
    assert e_w < 2e-2


WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from css_amd import ops
from css_amd.networks import resnet
from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
torch.manual_seed(3)
net = DeepLabv3Plus_with_rep(resnet.resnet101_tv() if sys.argv[2] == "tv" else resnet.resnet101(), dilate_scale=8, num_classes=21, output_dim=256)
net = net.to("cuda:0").train().set_compute_dtype(torch.bfloat16)
x = torch.randn(4, 3, 129, 129, generator=torch.Generator().manual_seed(5)).to("cuda:0")
staged = net.stage([x])
with torch.no_grad():
    with ops.bn_groups(2):
        feat = net.resnet_maxpool(net.resnet_bn1(net.resnet_conv1(staged), relu=True))      # stem (+ its batch norms) + max pool
        pred, rep = net.forward_nhwc(net.stage([x]))
    net.set_compute_dtype(torch.float32)
    with ops.bn_groups(2):
        pred32, _ = net.forward_nhwc(net.stage([x]))
json.dump(dict(kind=type(staged).__name__, feat=feat.float().cpu().flatten()[::11].tolist(), pred=pred.float().cpu().flatten()[::3].tolist(),
               pred32=pred32.float().cpu().flatten()[::3].tolist()), open(sys.argv[1], "w"))
'''


@pytest.mark.parametrize("backbone", ["tv", "stem"])
def test_network_takes_the_s2d_stem_and_agrees_with_the_gather_kernels(tmp_path, backbone):
    """bf16 network with the s2d stem (default) and with CSS_NO_STEM_S2D=1 (the gather kernels), same seeds, two statistics groups: the staging
    differs (S2DInput vs NHWC-8 tensor); the stem's output after its batch norm(s) and the max pool agrees to bf16 rounding (the epilogue's
    statistics slabs feed the batch norm on both paths); the logits of both paths sit equally far from the fp32 network's (printed)."""
    import json
    res = {}
    for tag, env in (("s2d", {}), ("gather", {"CSS_NO_STEM_S2D": "1"})):
        out = str(tmp_path / f"{tag}.json")
        e = dict(os.environ, **env)
        if tag == "s2d":
            e.pop("CSS_NO_STEM_S2D", None)
        p = subprocess.Popen([sys.executable, "-c", WORKER % ROOT, out, backbone], env=e)
        assert p.wait(timeout=600) == 0
        res[tag] = json.load(open(out))
    assert res["s2d"]["kind"] == "S2DInput" and res["gather"]["kind"] == "Tensor"
    cosf = lambda u, v: torch.nn.functional.cosine_similarity(torch.tensor(u), torch.tensor(v), dim=0).item()
    c_feat = cosf(res["s2d"]["feat"], res["gather"]["feat"])
    c_pred = cosf(res["s2d"]["pred"], res["gather"]["pred"])
    c32 = {t: cosf(res[t]["pred"], res[t]["pred32"]) for t in res}
    print(f"{backbone}: stem features cosine s2d vs gather {c_feat:.6f}; logits cosine s2d vs gather {c_pred:.5f}; vs the fp32 network: {c32}")
    # measured: features 1.00001; the LOGITS of this default-init network are chaotic in bf16 (both paths 0.65-0.66 against the fp32 network and
    # 0.74 against each other: test_network_gpu.py discusses the conditioning) - they are printed, the stem is what this test is about
    assert c_feat > 0.9995
    assert abs(c32["s2d"] - c32["gather"]) < 0.1, c32
