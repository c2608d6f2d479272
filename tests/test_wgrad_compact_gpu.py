"""conv_wgrad_p8_kernel (round 6): the 256 x 256 weight-gradient kernel with LIVE-ROW COMPACTION - for a dilated 3x3
convolution the pixel loop of a k-column tile of kernel row r runs only over the output rows whose source row is inside the image for r
(profiles/r06_aspp_zero_tap_share.txt: 12 / 24 / 40 % of the pixel steps of the ASPP branches multiply all-padding rows otherwise).
Geometry edge cases against torch-CPU fp32 on the same bf16-rounded operands: dilation >= map height (a kernel row entirely in the padding:
computed as zeros over all rows), maps narrower than one 32-pixel step, non-square maps, slices that end inside an image, one image, stride of
the compacted walk across several images per step (tiny maps), and dilation 1 (only the border row is dead).
Reference: the weight gradients of /root/reference/generalframeworks/networks/deeplabv3/aspp.py:17-24 and resnet.py:126-129 (dilated conv2)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import bf16_round, dev, rel_err  # noqa: E402

# N, H, W, Cin, Cout, dil
CASES = [(3, 17, 17, 256, 256, 12), (2, 9, 9, 256, 256, 12), (5, 33, 29, 512, 256, 6), (4, 65, 65, 256, 256, 36), (1, 65, 65, 256, 512, 24),
         (7, 5, 5, 256, 256, 2), (2, 40, 70, 256, 256, 1), (6, 21, 33, 768, 256, 18)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_dilated_weight_gradient_with_live_row_compaction(case):
    from css_amd import ops
    n, h, w, cin, cout, dil = case
    g = torch.Generator().manual_seed(77 + h + w + dil)
    x = bf16_round(torch.randn(n, h, w, cin, generator=g) + 0.25)
    wt = bf16_round(torch.randn(cout, 3, 3, cin, generator=g) / (cin * 9) ** 0.5)
    gy = bf16_round(torch.randn(n, h, w, cout, generator=g))
    xr = x.permute(0, 3, 1, 2)
    wr = wt.permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(xr, wr, None, 1, dil, dil).backward(gy.permute(0, 3, 1, 2))
    xg = x.to(dev(), torch.bfloat16)
    wg = wt.permute(0, 3, 1, 2).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    grads = []
    for _ in range(2):
        wg.grad = None
        ops.conv2d(xg, wg, None, 1, dil, dil).backward(gy.to(dev(), torch.bfloat16))
        grads.append(wg.grad.detach().clone())
    e = rel_err(grads[0].cpu(), wr.grad)
    # per kernel row: a wrong live-row range shows up in ONE row of taps only
    per_row = [rel_err(grads[0].cpu()[:, :, r], wr.grad[:, :, r]) if float(wr.grad[:, :, r].abs().max()) > 0 else
               float(grads[0].cpu()[:, :, r].abs().max()) for r in range(3)]
    print(f"{case}: wgrad rel err {e:.2e}, per kernel row {[f'{v:.1e}' for v in per_row]}")
    assert e < 5e-3 and max(per_row) < 5e-3, (e, per_row)
    assert torch.equal(grads[0].view(torch.int32), grads[1].view(torch.int32))            # ordered slab reduction: same bits run to run
