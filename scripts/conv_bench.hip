// Standalone timing harness for the conv kernels (ablations via -DCSS_ABLATE_*): hipcc -O3 --offload-arch=gfx950
#include "../css_amd/csrc/conv.hip"
#include "../css_amd/csrc/conv_wgrad.hip"
#include "../css_amd/csrc/conv_pp.hip"
#include "proto/conv_pp64.hip"
#include "../css_amd/csrc/conv_p8.hip"
#include "../css_amd/csrc/conv_ws.hip"
#include "../css_amd/csrc/conv_c64.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>
struct Shape { const char* name; int N, H, W, Cin, Cout, R, stride, pad, dil; };
int main() {
  std::vector<Shape> shapes = {
      {"l3 3x3 d2 256->256", 32, 65, 65, 256, 256, 3, 1, 2, 2},
      {"l3 1x1 1024->256", 32, 65, 65, 1024, 256, 1, 1, 0, 1},
      {"l3 1x1 256->1024", 32, 65, 65, 256, 1024, 1, 1, 0, 1},
      {"l4 3x3 d4 512->512", 32, 65, 65, 512, 512, 3, 1, 4, 4},
      {"l4 1x1 512->2048", 32, 65, 65, 512, 2048, 1, 1, 0, 1},
      {"aspp 3x3 d12 2048->256", 32, 65, 65, 2048, 256, 3, 1, 12, 12},
      {"head 3x3 304->256", 32, 129, 129, 304, 256, 3, 1, 1, 1},
      {"l1 1x1 64->256", 32, 129, 129, 64, 256, 1, 1, 0, 1},
  };
  const int only = getenv("CB_ONLY") ? atoi(getenv("CB_ONLY")) : -1;   // run a single shape (index into the table)
  // CB_SHAPE="N,H,W,Cin,Cout,R,stride,pad,dil[;...]": run these shapes instead of the table (a 1x1 shape is a plain GEMM M = N*H*W, K = Cin:
  // the comparison with scripts/gemm8p.hip on the same M / N / K)
  static char names[32][64];
  if (const char* cs = getenv("CB_SHAPE")) {
    shapes.clear();
    int v[9], n = 0;
    while (*cs && sscanf(cs, "%d,%d,%d,%d,%d,%d,%d,%d,%d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8) == 9 && n < 32) {
      snprintf(names[n], 64, "%dx%d^2 %dx%d %d->%d", v[0], v[1], v[5], v[5], v[3], v[4]);
      shapes.push_back({names[n], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]});
      ++n;
      while (*cs && *cs != ';') ++cs;
      if (*cs == ';') ++cs;
    }
  }
  const bool uniform = getenv("CB_UNIFORM") != nullptr;                 // operands uniform in [-1, 1) like the yardstick GEMM's
  auto f2bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); };
  for (auto& s : shapes) {
    if (only >= 0 && &s - shapes.data() != only) continue;
    const int Ho = (s.H + 2 * s.pad - s.dil * (s.R - 1) - 1) / s.stride + 1, Wo = Ho;
    size_t nx = (size_t)s.N * s.H * s.W * s.Cin, nw = (size_t)s.Cout * s.R * s.R * s.Cin, ny = (size_t)s.N * Ho * Wo * s.Cout;
    std::vector<unsigned short> hx(nx), hw(nw);
    for (auto& v : hx) v = 0x3C00 + (rand() & 0x3FF) - ((rand() & 1) << 15);   // random bf16-ish bit patterns around +-1
    for (auto& v : hw) v = 0x3800 + (rand() & 0x3FF) - ((rand() & 1) << 15);
    if (uniform) {
      for (auto& v : hx) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
      for (auto& v : hw) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    }
    void *dx, *dw, *dy;
    float* dwg;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, ny * 2); hipMalloc(&dwg, nw * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(dwg, 0, nw * 4);
    ConvArgs a{};
    a.src = dx; a.wt = dw; a.dst = dy; a.bias = nullptr;
    a.N = s.N; a.Hs = s.H; a.Ws = s.W; a.Cs = s.Cin; a.lds = s.Cin;
    a.Hd = Ho; a.Wd = Wo; a.Cd = s.Cout; a.ldd = s.Cout;
    a.R = s.R; a.S = s.R; a.stride = s.stride; a.pad = s.pad; a.dil = s.dil; a.mode = 0;
    a.M = s.N * Ho * Wo; a.Ktot = s.R * s.R * s.Cin;
#ifdef CB_NEW_EPILOGUE
    float* dstat = nullptr; void* dadd = nullptr;
    if (getenv("CB_STATS")) { hipMalloc(&dstat, ((size_t)(a.M + 127) / 128 + 2) * 2 * s.Cout * 4); a.stats = dstat; a.stat_Mg = a.M / 2; }
    if (getenv("CB_ADD")) { hipMalloc(&dadd, ny * 2); hipMemset(dadd, 0, ny * 2); a.addend = dadd; a.ld_add = s.Cout; }
#endif
    WgradArgs g{};
    g.x = dx; g.dy = dy; g.dw = dwg; g.N = s.N; g.Hs = s.H; g.Ws = s.W; g.Cs = s.Cin; g.ldx = s.Cin; g.Hd = Ho; g.Wd = Wo; g.Cd = s.Cout;
    g.ldy = s.Cout; g.R = s.R; g.S = s.R; g.stride = s.stride; g.pad = s.pad; g.dil = s.dil; g.M = a.M; g.Ktot = a.Ktot; g.m_per_split = a.M;
    // the slab workspace the product passes (ops.py): partial tiles + ordered reduction instead of atomics (CB_NO_WS=1: the atomics path)
    if (!getenv("CB_NO_WS")) {
      g.ws_bytes = css_wgrad_ws_bytes_(a.M, a.Ktot, s.Cout, CSS_BF16, 256);
      if (g.ws_bytes) hipMalloc((void**)&g.ws, g.ws_bytes);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2.0 * a.M * s.Cout * a.Ktot;
    for (int which = 0; which < (getenv("CB_NOWGRAD") ? 1 : 2); ++which) {
      for (int i = 0; i < 3; ++i) which ? css_launch_wgrad(g, CSS_BF16, 256, 0) : css_launch_conv(a, CSS_BF16, 256, 0);
      hipDeviceSynchronize();
      const int reps = 20;
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; ++i) which ? css_launch_wgrad(g, CSS_BF16, 256, 0) : css_launch_conv(a, CSS_BF16, 256, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-24s %s  M=%d K=%d N=%d  %8.1f us  %7.1f TFLOP/s\n", s.name, which ? "wgrad" : "fwd  ", a.M, a.Ktot, s.Cout, ms / reps * 1e3,
             flops / (ms / reps * 1e-3) / 1e12);
    }
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dwg);
  }
  return 0;
}
