"""Run-to-run reproducibility of the training step ("deterministic reductions on the parity path"; SURVEY 7 "hard parts":
the fp32 parity path needs a deterministic reduction order).

Every sum of the step is ordered: loss statistics are integer accumulators, the adjoint of the fused up-sampling flushes in coloured
launches, class sums / bias gradients / weight-gradient slices go through per-workgroup rows that a second kernel adds in row order,
batch-norm statistics were two-stage from the start.  So two runs from the same seeds must agree BIT FOR BIT - losses, every weight,
momentum, the EMA teacher, the prototypes - in fp32 and in bf16, over ten steps at the training learning rate (the regime in which one
differing last bit is amplified to per cents within a few steps: r03 measured two fp32 runs 4.8 steps apart after 30).
Reference step: /root/reference/mix_label.py:162-196.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev  # noqa: E402
from test_bf16_trajectory_gpu import _batch, _trainer  # noqa: E402


def _run(dtype, steps, S=129, B=4, lr=6.4e-3):
    l_img, l_lab, u_img = _batch(S, B, 5, 16)
    tr = _trainer(S, 11, 0.25, dtype, lr, 256, 512)
    np.random.seed(0)
    torch.manual_seed(0)
    hist = []
    for _ in range(steps):
        r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()))
        hist.append(torch.stack([r[k].float().reshape(()) for k in ("sup", "unsup", "contrast", "total")]).cpu())
    out = dict(hist=torch.stack(hist), p=tr.flat_p.detach().cpu().clone(), m=tr.flat_m.detach().cpu().clone(),
               ema=tr.flat_ema.detach().cpu().clone(), proto=tr.prototypes.detach().cpu().clone())
    del tr
    torch.cuda.empty_cache()
    return out


def _same_bits(a, b):
    # (NaN-safe: unsup is NaN by definition while no pseudo label is confident)
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_ten_steps_twice_bit_identical(dtype):
    a, b = _run(dtype, 10), _run(dtype, 10)
    for i in range(a["hist"].shape[0]):
        assert _same_bits(a["hist"][i], b["hist"][i]), (i, a["hist"][i].tolist(), b["hist"][i].tolist())
    for key in ("p", "m", "ema", "proto"):
        assert _same_bits(a[key], b[key]), (key, int((a[key] != b[key]).sum()), float((a[key] - b[key]).abs().max()))
    # it trains (the assertion was dropped in round 5 with a justification that held for the opt-in 64 x 64 weight-gradient tiles only; on
    # the default build - measured in round 6, scripts/det_probe.py - the supervised loss of this run falls 7.03 -> 0.80 in fp32 and 7.06 -> 0.80
    # in bf16 over the ten steps, the total 12.8 -> 6.9 / 7.2): the supervised loss must at least halve
    assert float(a["hist"][-1, 0]) < 0.5 * float(a["hist"][0, 0]), a["hist"][:, 0].tolist()
    assert float(a["hist"][-1, 3]) < float(a["hist"][0, 3]), a["hist"][:, 3].tolist()
    assert torch.isfinite(a["hist"][:, [0, 2, 3]]).all()


def test_weight_gradient_kernels_are_reproducible_at_bench_scale():
    """The weight gradient of every kernel family twice on the same operands: bit-identical (slices are added in slice order)."""
    from css_amd import ops
    torch.manual_seed(3)
    cases = [  # (dtype, N, H, Cin, Cout, k, dil): small-tile bf16 (Cout < 256), big-tile bf16, fp32
        (torch.bfloat16, 32, 129, 64, 64, 3, 1), (torch.bfloat16, 32, 65, 256, 256, 3, 2), (torch.bfloat16, 32, 129, 64, 256, 1, 1),
        (torch.float32, 4, 65, 64, 64, 3, 1), (torch.float32, 4, 33, 256, 512, 1, 1)]
    for dt, n, h, cin, cout, k, dil in cases:
        x = torch.randn(n, h, h, cin, device=dev()).to(dt)
        w = (torch.randn(cout, cin, k, k, device=dev()) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        grads = []
        for _ in range(2):
            w.grad = None
            y = ops.conv2d(x, w, None, 1, dil * (k // 2), dil)
            y.backward(torch.ones_like(y) * 0.01 + y.detach() * 0.001)
            grads.append(w.grad.detach().clone())
        assert _same_bits(grads[0].float().contiguous(), grads[1].float().contiguous()), (dt, n, h, cin, cout, k)


def test_loss_kernels_are_reproducible_at_bench_scale():
    """CE / unsupervised loss from the low-resolution logits at the c2 launch shape (16 x 129^2 -> 513^2), forward and backward twice:
    identical bits (integer statistics, coloured flush); class sums and bias gradient likewise."""
    from css_amd import ops
    from css_amd._lib import call, dev_stream, dtype_code, query
    from css_amd.loss.loss import Attention_Threshold_Loss, CrossEntropyLoss
    torch.manual_seed(5)
    B, h, H, K = 8, 129, 513, 21
    small = torch.randn(B, h, h, K, device=dev()).to(torch.bfloat16)
    lab = torch.randint(-1, K, (B, H, H), device=dev())
    conf = torch.rand(B, H, H, device=dev())
    outs = []
    for _ in range(2):
        s1 = small.clone().requires_grad_(True)
        l1 = CrossEntropyLoss(-1).forward_small(s1, lab)
        l2 = Attention_Threshold_Loss(0.5).forward_small(s1, lab, conf)
        (l1 + 0.5 * l2).backward()
        outs.append((l1.detach().clone(), l2.detach().clone(), s1.grad.float().clone()))
    for a, b in zip(*outs):
        assert _same_bits(a.contiguous(), b.contiguous())
    # class sums (fp64 out) and column sums
    P, C = 32 * 129 * 129, 256
    rep = torch.randn(P, C, device=dev()).to(torch.bfloat16)
    cls = torch.randint(-1, K, (P,), device=dev(), dtype=torch.int32)
    d, st = dev_stream(rep)
    res = []
    for _ in range(2):
        sums = torch.empty(K * C + K, dtype=torch.float64, device=dev())
        ws = torch.empty(query("css_contrast_class_sums_ws_bytes", P, K, C) // 4, dtype=torch.float32, device=dev())
        call("css_contrast_class_sums", rep, C, cls, P, K, C, sums, ws, dtype_code(rep.dtype), d, st)
        col = torch.zeros(C, dtype=torch.float32, device=dev())
        cws = torch.empty(query("css_colsum_ws_bytes", P, C) // 4, dtype=torch.float32, device=dev())
        call("css_colsum", rep, C, P, C, col, cws, dtype_code(rep.dtype), d, st)
        res.append((sums.clone(), col.clone()))
    assert torch.equal(res[0][0].view(torch.int64), res[1][0].view(torch.int64)) and _same_bits(res[0][1], res[1][1])
    # and they are the right sums
    ref = torch.zeros(K, C, dtype=torch.float64, device=dev()).index_add_(0, cls[cls >= 0].long(), rep[cls >= 0].double())
    assert torch.allclose(res[0][0][:K * C].view(K, C), ref, rtol=1e-6, atol=1e-3)
    assert torch.allclose(res[0][1].double(), rep.double().sum(0), rtol=1e-4, atol=5e-2)
    # the other two column-sum kernels: C = 21 (class logits: dense lanes, 12 rows per pass) and a width above 256 without the vector layout,
    # both accumulating INTO out, fp32 and bf16, ragged row counts
    for C2, M2, dt in ((21, 32 * 129 * 129, torch.bfloat16), (19, 100003, torch.float32), (300, 70001, torch.bfloat16), (257, 513, torch.float32)):
        x = torch.randn(M2, C2, device=dev()).to(dt)
        outs = []
        for _ in range(2):
            o = torch.full((C2,), 0.5, dtype=torch.float32, device=dev())
            w = torch.empty(query("css_colsum_ws_bytes", M2, C2) // 4, dtype=torch.float32, device=dev())
            call("css_colsum", x, C2, M2, C2, o, w, dtype_code(dt), d, st)
            outs.append(o.clone())
        assert _same_bits(outs[0], outs[1]), (C2, M2)
        assert torch.allclose(outs[0].double(), x.double().sum(0) + 0.5, rtol=1e-4, atol=5e-2), (C2, M2, float((outs[0].double() - x.double().sum(0) - 0.5).abs().max()))
