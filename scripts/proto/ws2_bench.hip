// PROTOTYPE harness for scripts/proto/conv_ws2.hip (next round): conv_ws_kernel vs conv_ws2_kernel on the 256 -> N shapes, plain epilogue;
// outputs compared element by element (same MFMA instruction on the same K blocks: they must be identical), 20 launches timed each.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/proto/ws2_bench.hip -o build/ws2_bench && ./build/ws2_bench
#include "../../css_amd/csrc/conv.hip"
#include "../../css_amd/csrc/conv_pp.hip"
#include "../../css_amd/csrc/conv_pp64.hip"
#include "../../css_amd/csrc/conv_ws.hip"
#include "conv_ws2.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Shape { const char* name; int N, H, W, Cin, Cout; };
int main() {
  std::vector<Shape> shapes = {{"l3 conv3 256->1024", 32, 65, 65, 256, 1024}, {"256->512", 32, 65, 65, 256, 512}, {"c4 256->1024 (16 x 97^2)", 16, 97, 97, 256, 1024},
                               {"tiny 256->1024 (2 x 9^2)", 2, 9, 9, 256, 1024}, {"ragged 256->2048 (3 x 21^2)", 3, 21, 21, 256, 2048}};
  int bad = 0;
  for (auto& s : shapes) {
    const int M = s.N * s.H * s.W;
    const size_t nx = (size_t)M * s.Cin, nw = (size_t)s.Cout * s.Cin, ny = (size_t)M * s.Cout;
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(99);
    for (auto& v : hx) v = 0x3C00 + (rand() & 0x3FF) - ((rand() & 1) << 15);
    for (auto& v : hw) v = 0x3800 + (rand() & 0x3FF) - ((rand() & 1) << 15);
    void *dx, *dw, *dy[2];
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy[0], ny * 2); hipMalloc(&dy[1], ny * 2);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ep = 0; ep < 2; ++ep) {      // 0: plain, 1: BN statistics (one group: 128-row slabs of ws against pairs of 64-row slabs of ws2)
      float us[2];
      float* dst[2] = {nullptr, nullptr};
      const size_t n128 = (size_t)2 * ((M + 255) / 256), n64 = (size_t)(M + 63) / 64;
      hipMalloc(&dst[0], n128 * 2 * s.Cout * 4); hipMalloc(&dst[1], n64 * 2 * s.Cout * 4);
      hipMemset(dst[0], 0, n128 * 2 * s.Cout * 4); hipMemset(dst[1], 0, n64 * 2 * s.Cout * 4);
      for (int v = 0; v < 2; ++v) {
        ConvArgs a{};
        a.src = dx; a.wt = dw; a.dst = dy[v];
        a.N = s.N; a.Hs = a.Hd = s.H; a.Ws = a.Wd = s.W; a.Cs = a.lds = s.Cin; a.Cd = a.ldd = s.Cout;
        a.R = a.S = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.M = M; a.Ktot = s.Cin;
        a.src_bytes = (unsigned)(nx * 2); a.wt_bytes = (unsigned)(nw * 2);
        if (ep) { a.stats = dst[v]; a.stat_Mg = M; }
        if (!(v ? css_conv_ws2_supported(a, 256) : css_conv_ws_supported(a, 256))) { printf("%s: variant %d not supported\n", s.name, v); return 2; }
        hipMemset(dy[v], 0xFF, ny * 2);
        auto go = [&]() { v ? css_launch_conv_ws2(a, 256, 0) : css_launch_conv_ws(a, 256, 0); };
        for (int i = 0; i < 3; ++i) go();
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) go();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        us[v] = ms / 20 * 1e3f;
      }
      std::vector<unsigned short> y0(ny), y1(ny);
      hipMemcpy(y0.data(), dy[0], ny * 2, hipMemcpyDeviceToHost);
      hipMemcpy(y1.data(), dy[1], ny * 2, hipMemcpyDeviceToHost);
      size_t nd = 0;
      for (size_t i = 0; i < ny; ++i) nd += y0[i] != y1[i];
      double smax = 0;
      if (ep) {
        std::vector<float> s0(n128 * 2 * s.Cout), s1(n64 * 2 * s.Cout);
        hipMemcpy(s0.data(), dst[0], s0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(s1.data(), dst[1], s1.size() * 4, hipMemcpyDeviceToHost);
        for (size_t p = 0; p < (size_t)(M + 127) / 128; ++p)
          for (int c = 0; c < 2 * s.Cout; ++c) {
            const double want = s0[p * 2 * s.Cout + c];
            const double got = (double)s1[(2 * p) * 2 * s.Cout + c] + (2 * p + 1 < n64 ? (double)s1[(2 * p + 1) * 2 * s.Cout + c] : 0.0);
            const double d = fabs(want - got) / (fabs(want) + 1.0);
            if (!(d <= smax)) smax = d;
          }
      }
      const bool ok = nd == 0 && smax < 1e-5;
      if (!ok) ++bad;
      printf("%-30s %s M=%-6d ws %7.1f us  ws2 %7.1f us (%.2fx, %.2f TB/s)  mismatching elements %zu / %zu  stats rel %.2g  %s\n", s.name, ep ? "stats" : "plain", M,
             us[0], us[1], us[0] / us[1], 2.0 * (nx + ny + nw) / (us[1] * 1e-6) / 1e12, nd, ny, smax, ok ? "OK" : "MISMATCH");
      hipFree(dst[0]); hipFree(dst[1]);
    }
    hipFree(dx); hipFree(dw); hipFree(dy[0]); hipFree(dy[1]);
  }
  return bad ? 1 : 0;
}
