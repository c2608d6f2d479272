"""Micro-benchmark of the BN elementwise / reduction kernels against torch copy / add at the same sizes."""
import sys, torch
sys.path.insert(0, '.')
from css_amd._lib import call, dev_stream, query
from css_amd.ops import dtype_code

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us

dev = torch.device('cuda:0')
for (M, C) in [(34848, 256), (34848, 1024), (34848, 2048), (135200, 128), (135200, 512), (532512, 64), (532512, 256)]:
    x = torch.randn(M, C, device=dev).bfloat16(); r = torch.randn(M, C, device=dev).bfloat16()
    out = torch.empty_like(x); dy = torch.empty_like(x); dres = torch.empty_like(x)
    G, Mg = 2, M // 2
    scale = torch.rand(G * C, device=dev); shift = torch.rand(G * C, device=dev)
    mean = torch.rand(G * C, device=dev); invstd = torch.rand(G * C, device=dev); gamma = torch.rand(C, device=dev)
    dc = dtype_code(torch.bfloat16); d, st = dev_stream(x)
    nrb = query("css_bn_nrb", Mg, G, C, dc)
    partial = torch.empty(G * nrb * 2 * C, dtype=torch.float64, device=dev)
    sums = torch.zeros(G * 2 * C, dtype=torch.float64, device=dev)
    mb = M * C * 2 / 1e6
    t_copy = timeit(lambda: out.copy_(x))
    t_add = timeit(lambda: torch.add(x, r, out=out))
    t_apply = timeit(lambda: call("css_bn_apply", x, C, None, C, out, C, scale, shift, M, C, 1, Mg, dc, d, st))
    t_applyr = timeit(lambda: call("css_bn_apply", x, C, r, C, out, C, scale, shift, M, C, 1, Mg, dc, d, st))
    t_stats = timeit(lambda: call("css_bn_stats", x, Mg, G, C, C, partial, dc, d, st))
    t_bred = timeit(lambda: call("css_bn_bwd_reduce", r, C, None, C, x, C, mean, invstd, scale, shift, Mg, G, C, 1, partial, dc, d, st))
    t_bapp = timeit(lambda: call("css_bn_bwd_apply", r, C, None, C, x, C, dy, C, None, C, mean, invstd, gamma, sums, scale, shift, float(Mg), M, C, 1, Mg, dc, d, st))
    t_bappr = timeit(lambda: call("css_bn_bwd_apply", r, C, out, C, x, C, dy, C, dres, C, mean, invstd, gamma, sums, scale, shift, float(Mg), M, C, 1, Mg, dc, d, st))
    def bw(t, passes): return passes * mb / t   # TB/s: MB/us
    print(f"M={M:7d} C={C:5d} {mb:7.1f} MB | copy {t_copy:6.1f}us {bw(t_copy,2):.2f} | add {t_add:6.1f} {bw(t_add,3):.2f} | apply {t_apply:6.1f} {bw(t_apply,2):.2f}"
          f" | apply+res {t_applyr:6.1f} {bw(t_applyr,3):.2f} | stats {t_stats:6.1f} {bw(t_stats,1):.2f} | bred {t_bred:6.1f} {bw(t_bred,2):.2f}"
          f" | bapp {t_bapp:6.1f} {bw(t_bapp,3):.2f} | bapp+res {t_bappr:6.1f} {bw(t_bappr,5):.2f}  TB/s", flush=True)
