"""Time of the stem convolution (+ the batch norm that follows: statistics from the epilogue on both paths) at the bench's launch shape, the
space-to-depth kernel against the gather kernel - torch events on the launch stream, 20 launches each.  python scripts/stem_time.py [tv|stem]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from css_amd import ops
from css_amd.nn import HipBatchNorm2d, HipConv2d

kind = sys.argv[1] if len(sys.argv) > 1 else "tv"
r, n, h = (7, 32, 513) if kind == "tv" else (3, 16, 769)
dev = torch.device("cuda:0")
x = torch.randn(n, 3, h, h, device=dev)
conv = HipConv2d(3, 64, r, 2, r // 2, bias=False).to(dev).train()
bn = HipBatchNorm2d(64).to(dev).train()


def run(s2d, what):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad(), ops.bn_groups(2):
        st = ops.stage_inputs([x], torch.bfloat16, s2d=s2d)
        f = {"conv": lambda: conv(st), "conv+bn": lambda: bn(conv(st), relu=True), "stage": lambda: ops.stage_inputs([x], torch.bfloat16, s2d=s2d)}[what]
        for _ in range(3):
            f()
        e0.record()
        for _ in range(20):
            f()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for what in ("stage", "conv", "conv+bn"):
    print(f"{kind} {r}x{r} s2 {n}x{h}^2 {what:8s}: gather {run(False, what):8.1f} us   s2d {run(True, what):8.1f} us")
