import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from css_amd import ops
from gpu_util import to_nhwc, to_nchw_cpu, rel_err, dev
for (n,h,w,c,res,relu) in [(3,9,9,2048,True,True),(3,9,9,2048,False,False),(3,9,9,512,False,True),(3,9,9,1024,False,True),(16,65,65,256,False,True),(2,17,17,64,True,True)]:
    g = torch.Generator().manual_seed(c+h)
    x = torch.randn(n,c,h,w,generator=g)*2+0.5
    r = torch.randn(n,c,h,w,generator=g) if res else None
    gamma, beta = torch.rand(c,generator=g)+0.5, torch.randn(c,generator=g)*0.1
    xr = x.clone().requires_grad_(True); rr = r.clone().requires_grad_(True) if res else None
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    o = F.batch_norm(xr, torch.zeros(c), torch.ones(c), gr, br, True, 0.1, 1e-5)
    if res: o = o + rr
    if relu: o = F.relu(o)
    go = torch.randn(o.shape, generator=g); o.backward(go)
    xg = to_nhwc(x, torch.float32).requires_grad_(True); rg = to_nhwc(r, torch.float32).requires_grad_(True) if res else None
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    og = ops.bn_act(xg, gg, bg, torch.zeros(c,device=dev()), torch.ones(c,device=dev()), rg, relu, True, 0.1, 1e-5, False)
    og.backward(to_nhwc(go, torch.float32))
    print((n,h,w,c,res,relu), "fwd %.1e dx %.1e dgamma %.1e dbeta %.1e" % (rel_err(to_nchw_cpu(og), o.detach()), rel_err(to_nchw_cpu(xg.grad), xr.grad), rel_err(gg.grad.cpu(), gr.grad), rel_err(bg.grad.cpu(), br.grad)))
