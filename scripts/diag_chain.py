import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from css_amd import ops
from gpu_util import to_nhwc, to_nchw_cpu, rel_err, dev
g = torch.Generator().manual_seed(0)
B, H, C0, C1, K = 3, 17, 304, 256, 21
x = torch.randn(B, C0, H, H, generator=g)
w0 = torch.randn(C1, C0, 3, 3, generator=g) * 0.02
gam, bet = torch.rand(C1, generator=g) + 0.5, torch.randn(C1, generator=g) * 0.1
w1 = torch.randn(K, C1, 1, 1, generator=g) * 0.06
b1 = torch.randn(K, generator=g) * 0.1
wl = torch.randn(B, K, H, H, generator=g)
# CPU
P = [t.clone().requires_grad_(True) for t in (x, w0, gam, bet, w1, b1)]
y0 = F.conv2d(P[0], P[1], None, 1, 1)
a = F.relu(F.batch_norm(y0, torch.zeros(C1), torch.ones(C1), P[2], P[3], True, 0.1, 1e-5))
o = F.conv2d(a, P[4], P[5])
(o * wl).sum().backward()
# GPU
xg = to_nhwc(x, torch.float32).requires_grad_(True)
w0g = w0.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
gg, bg = gam.to(dev()).requires_grad_(True), bet.to(dev()).requires_grad_(True)
w1g = w1.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
b1g = b1.to(dev()).requires_grad_(True)
y0g = ops.conv2d(xg, w0g, None, 1, 1, 1)
ag = ops.bn_act(y0g, gg, bg, torch.zeros(C1, device=dev()), torch.ones(C1, device=dev()), None, True, True, 0.1, 1e-5, False)
ag.retain_grad(); y0g.retain_grad()
og = ops.conv2d(ag, w1g, b1g, 1, 0, 1)
print("fwd", rel_err(to_nchw_cpu(og), o.detach()))
pred = og.permute(0, 3, 1, 2)
(pred * wl.to(dev())).sum().backward()
y0.retain_grad
print("dx", rel_err(to_nchw_cpu(xg.grad), P[0].grad))
print("dw0", rel_err(w0g.grad.cpu(), P[1].grad))
print("dgamma", rel_err(gg.grad.cpu(), P[2].grad), "dbeta", rel_err(bg.grad.cpu(), P[3].grad))
print("dw1", rel_err(w1g.grad.cpu(), P[4].grad), "db1", rel_err(b1g.grad.cpu(), P[5].grad))
# da check: CPU da = conv_transpose
da_ref = torch.autograd.grad((F.conv2d(a.detach().requires_grad_(True), w1, b1) * wl).sum(), [], allow_unused=True) if False else None
a2 = a.detach().clone().requires_grad_(True)
(F.conv2d(a2, w1, b1) * wl).sum().backward()
print("da", rel_err(to_nchw_cpu(ag.grad), a2.grad))
