"""Time of the stem's max pool (forward, backward) at the bench's launch shape.  python scripts/pool_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from css_amd import ops
dev = torch.device("cuda:0")
for n, h, ceil in ((32, 257, False), (16, 385, True)):
    x = torch.randn(n, h, h, 64, device=dev).to(torch.bfloat16).requires_grad_(True)
    y = ops.maxpool(x, 3, 2, 1, ceil)
    g = torch.randn_like(y)
    def t(f):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): f()
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    def fwd():
        with torch.no_grad(): ops.maxpool(x, 3, 2, 1, ceil)
    def bwd():
        x.grad = None
        y.backward(g, retain_graph=True)
    print(f"maxpool {n}x{h}^2x64 ceil={ceil}: forward {t(fwd):7.1f} us   backward {t(bwd):7.1f} us")
