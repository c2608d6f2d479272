"""bn_apply on cold data: rotate over many buffers (> 256 MB Infinity Cache) and compare with torch copy."""
import sys, torch
sys.path.insert(0, '.')
from css_amd._lib import call, dev_stream
from css_amd.ops import dtype_code
dev = torch.device('cuda:0')
def timeit(f, n):
    for i in range(n): f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(n): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, C) in [(34848, 256), (34848, 1024), (135200, 128)]:
    NB = 40
    xs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(NB)]
    outs = [torch.empty_like(xs[0]) for _ in range(NB)]
    G, Mg = 2, M // 2
    scale = torch.rand(G * C, device=dev); shift = torch.rand(G * C, device=dev)
    dc = dtype_code(torch.bfloat16); d, st = dev_stream(xs[0])
    mb = M * C * 2 / 1e6
    t_copy = timeit(lambda i: outs[i % NB].copy_(xs[i % NB]), NB * 3)
    t_apply = timeit(lambda i: call("css_bn_apply", xs[i % NB], C, None, C, outs[i % NB], C, scale, shift, M, C, 1, Mg, dc, d, st), NB * 3)
    # producer -> consumer chain: copy writes a buffer, apply reads it right away
    def chain(i):
        outs[i % NB].copy_(xs[i % NB])
        call("css_bn_apply", outs[i % NB], C, None, C, xs[(i + 7) % NB], C, scale, shift, M, C, 1, Mg, dc, d, st)
    t_chain = timeit(chain, NB * 3)
    print(f"M={M} C={C} {mb:.1f} MB cold: copy {t_copy:.1f} us, apply {t_apply:.1f} us, copy+apply chain {t_chain:.1f} us", flush=True)
