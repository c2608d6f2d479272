// conv_igemm_p8_kernel (conv_p8.hip) against conv_igemm_pp64_kernel (conv_pp64.hip) on the bench shapes, same launch schedule:
//   * parity: outputs (and the addend form) must be BIT-IDENTICAL (same MFMA instruction on the same K blocks in the same order), compared
//     element by element; the statistics slabs to 1e-5 of their magnitude (round 5: summed on the matrix pipe; -DP8_NO_MFMA_STATS: bit-identical);
//   * race screen: P8_RACE (default 50) repeated launches of the new kernel must reproduce its first result bit for bit;
//   * timing: both kernels in interleaved rounds in one process (min / median over rounds), uniform random operands in [-1, 1).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/p8_bench.hip -o build/p8_bench && ./build/p8_bench
#include "../css_amd/csrc/conv.hip"
#include "../css_amd/csrc/conv_wgrad.hip"
#include "../css_amd/csrc/conv_pp.hip"
#include "proto/conv_pp64.hip"
#include "../css_amd/csrc/conv_p8.hip"
#include "../css_amd/csrc/conv_ws.hip"
#include "../css_amd/csrc/conv_c64.hip"
#define G8_NO_MAIN
#include "gemm8p.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Shape { const char* name; int N, H, W, Cin, Cout, R, pad, dil, mode; };
int main() {
  std::vector<Shape> shapes = {
      {"l3 3x3 d2 256->256", 32, 65, 65, 256, 256, 3, 2, 2, 0},
      {"l3 1x1 1024->256", 32, 65, 65, 1024, 256, 1, 0, 1, 0},
      {"l4 3x3 d4 512->512", 32, 65, 65, 512, 512, 3, 4, 4, 0},
      {"l4 1x1 2048->512", 32, 65, 65, 2048, 512, 1, 0, 1, 0},
      {"aspp 3x3 d12 2048->256", 32, 65, 65, 2048, 256, 3, 12, 12, 0},
      {"aspp 3x3 d36 2048->256", 32, 65, 65, 2048, 256, 3, 36, 36, 0},
      {"head 3x3 304->256", 32, 129, 129, 304, 256, 3, 1, 1, 0},
      {"l3 3x3 d2 dgrad 256->256", 32, 65, 65, 256, 256, 3, 2, 2, 1},
      {"c4 l3 3x3 d2 (16 x 97^2)", 16, 97, 97, 256, 256, 3, 2, 2, 0},
      {"ragged 3x3 320->264 (3 x 21^2)", 3, 21, 21, 320, 264, 3, 1, 1, 0},
      {"gemm 4608->512 (32 x 64^2)", 32, 64, 64, 4608, 512, 1, 0, 1, 0},
      {"gemm 2304->256 (32 x 64^2)", 32, 64, 64, 2304, 256, 1, 0, 1, 0},
      {"gemm 4608->256 (32 x 64^2)", 32, 64, 64, 4608, 256, 1, 0, 1, 0},
      {"gemm 1024->256 (32 x 64^2)", 32, 64, 64, 1024, 256, 1, 0, 1, 0},
      {"gemm 4096->4096 (16 x 64^2)", 16, 64, 64, 4096, 4096, 1, 0, 1, 0},
  };
  const int only = getenv("P8_ONLY") ? atoi(getenv("P8_ONLY")) : -1;
  const int from = getenv("P8_FROM") ? atoi(getenv("P8_FROM")) : 0;
  const int nrace = getenv("P8_RACE") ? atoi(getenv("P8_RACE")) : 50;
  const int rounds = getenv("P8_ROUNDS") ? atoi(getenv("P8_ROUNDS")) : 5;
  auto f2bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); };
  int bad_total = 0;
  for (auto& s : shapes) {
    if ((only >= 0 && &s - shapes.data() != only) || &s - shapes.data() < from) continue;
    const int Ho = s.H + 2 * s.pad - s.dil * (s.R - 1), Wo = Ho;        // stride 1
    const int M = s.N * Ho * Wo;
    const size_t nx = (size_t)s.N * s.H * s.W * s.Cin, nw = (size_t)s.Cout * s.R * s.R * s.Cin, ny = (size_t)M * s.Cout;
    std::vector<unsigned short> hx(nx), hw(nw), hadd(ny);
    srand(1234);
    for (auto& v : hx) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    for (auto& v : hw) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    for (auto& v : hadd) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    void *dx, *dw, *dy[2], *dadd;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy[0], ny * 2); hipMalloc(&dy[1], ny * 2); hipMalloc(&dadd, ny * 2);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(dadd, hadd.data(), ny * 2, hipMemcpyHostToDevice);
    const size_t nstat = (size_t)2 * ((M + 255) / 256) * 2 * s.Cout;
    float* dstat[2];
    hipMalloc(&dstat[0], nstat * 4); hipMalloc(&dstat[1], nstat * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ep = 0; ep < 3; ++ep) {        // 0: plain, 1: statistics (two groups), 2: addend
      ConvArgs a{};
      a.src = dx; a.wt = dw; a.bias = nullptr;
      a.N = s.N; a.Hs = s.H; a.Ws = s.W; a.Cs = s.Cin; a.lds = s.Cin;
      a.Hd = Ho; a.Wd = Wo; a.Cd = s.Cout; a.ldd = s.Cout;
      a.R = s.R; a.S = s.R; a.stride = 1; a.pad = s.pad; a.dil = s.dil; a.mode = s.mode;
      a.M = M; a.Ktot = s.R * s.R * s.Cin;
      a.src_bytes = (unsigned)(nx * 2); a.wt_bytes = (unsigned)(nw * 2); a.dst_bytes = (unsigned)(ny * 2);
      a.m_begin = 0;
      a.stat_nslab = (M + 127) / 128;
      if (ep == 2) { a.addend = dadd; a.ld_add = s.Cout; }
      if (!css_conv_pp64_supported(a)) { printf("%-32s not a pp64 shape\n", s.name); break; }
      const int tiles = ((M + 255) / 256) * ((s.Cout + 255) / 256), grid = tiles < 256 ? tiles : 256;
      auto run = [&](int v) {
        ConvArgs b = a;
        const int ob = (v == 0 || v == 4) ? 0 : 1;       // output buffer: pp64 variants -> 0, the others -> 1
        b.dst = dy[ob];
        if (ep == 1) { b.stats = dstat[ob]; b.stat_Mg = M / 2; b.stat_G = 2; }
        if (v == 0) css_launch_conv_pp64(b, grid, 0);
        else if (v == 1) css_launch_conv_p8(b, grid, 0);
        else if (v == 3) css_launch_conv_p8(b, tiles, 0);             // the same kernel, one workgroup per tile (not persistent)
        else if (v == 4) css_launch_conv_pp64(b, tiles, 0);
        else hipLaunchKernelGGL(gemm8p_kernel, dim3(tiles), dim3(512), 0, 0, dx, dw, dy[1], M, s.Cout, s.Cin);   // the yardstick on the same GEMM
      };
      const bool gemm = s.R == 1 && ep == 0 && s.Cout % 256 == 0 && s.Cin % 64 == 0;
      hipMemset(dy[0], 0xFF, ny * 2); hipMemset(dy[1], 0xFF, ny * 2);
      hipMemset(dstat[0], 0, nstat * 4); hipMemset(dstat[1], 0, nstat * 4);
      run(0); run(1);
      hipDeviceSynchronize();
      std::vector<unsigned short> h0(ny), h1(ny), h2(ny);
      hipMemcpy(h0.data(), dy[0], ny * 2, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), dy[1], ny * 2, hipMemcpyDeviceToHost);
      size_t mism = 0;
      for (size_t i = 0; i < ny; ++i) mism += h0[i] != h1[i];
      size_t smism = 0;
      double srel = 0;         // largest difference of a statistics entry relative to the largest entry (P8_MFMA_STATS: another summation order)
      if (ep == 1) {
        std::vector<float> s0(nstat), s1(nstat);
        hipMemcpy(s0.data(), dstat[0], nstat * 4, hipMemcpyDeviceToHost);
        hipMemcpy(s1.data(), dstat[1], nstat * 4, hipMemcpyDeviceToHost);
        double smax = 0, sd = 0;
        for (size_t i = 0; i < nstat; ++i) {
          smism += memcmp(&s0[i], &s1[i], 4) != 0;
          if (fabs((double)s0[i]) > smax) smax = fabs((double)s0[i]);
          if (!(fabs((double)s0[i] - (double)s1[i]) <= sd)) sd = fabs((double)s0[i] - (double)s1[i]);
        }
        srel = sd / (smax + 1e-30);
#ifndef P8_NO_MFMA_STATS     // (the shipped kernel sums a slab on the matrix pipe: the same quantity in another order than pp64's VALU sums)
        if (srel <= 1e-5) smism = 0;
#endif
      }
      int racebad = 0;
      for (int r = 0; r < (ep == 0 ? nrace : 5); ++r) {
        hipMemset(dy[1], 0xFF, ny * 2);
        run(1);
        hipMemcpy(h2.data(), dy[1], ny * 2, hipMemcpyDeviceToHost);
        if (memcmp(h1.data(), h2.data(), ny * 2)) ++racebad;
      }
      size_t gmism = 0;
      if (gemm) {
        hipMemset(dy[1], 0xFF, ny * 2);
        run(2);
        hipMemcpy(h2.data(), dy[1], ny * 2, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < ny; ++i) gmism += h0[i] != h2[i];
      }
      std::vector<float> us[5];
      for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < 5; ++v) {
          if (v == 2 && !gemm) continue;
          for (int i = 0; i < 2; ++i) run(v);
          const int reps = 10;
          hipEventRecord(e0, 0);
          for (int i = 0; i < reps; ++i) run(v);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          us[v].push_back(ms / reps * 1e3f);
        }
      std::sort(us[0].begin(), us[0].end()); std::sort(us[1].begin(), us[1].end()); std::sort(us[2].begin(), us[2].end()); std::sort(us[3].begin(), us[3].end()); std::sort(us[4].begin(), us[4].end());
      const double flops = 2.0 * M * s.Cout * a.Ktot;
      const bool ok = mism == 0 && smism == 0 && racebad == 0;
      bad_total += !ok;
      printf("%-32s %-6s M=%d K=%d N=%d  pp64 %8.1f us (%6.1f TF)  p8 %8.1f us (%6.1f TF)  median p8/pp64 %.3f  mismatch %zu stats %zu (rel %.2g) racebad %d  %s\n", s.name,
             ep == 0 ? "plain" : ep == 1 ? "stats" : "addend", M, a.Ktot, s.Cout, us[0][0], flops / us[0][0] * 1e-6, us[1][0], flops / us[1][0] * 1e-6,
             us[1][rounds / 2] / us[0][rounds / 2], mism, smism, srel, racebad, ok ? "OK" : "FAIL");
      printf("%-32s        one workgroup per tile: p8 %8.1f us (%6.1f TF) median / pp64 persistent %.3f;  pp64 %8.1f us (%6.1f TF) median / pp64 persistent %.3f\n", "",
             us[3][0], flops / us[3][0] * 1e-6, us[3][rounds / 2] / us[0][rounds / 2], us[4][0], flops / us[4][0] * 1e-6, us[4][rounds / 2] / us[0][rounds / 2]);
      if (gemm) printf("%-32s        gemm8p (yardstick, one workgroup per tile) %8.1f us (%6.1f TF)  median gemm8p/pp64 %.3f  mismatch vs pp64 %zu\n", "", us[2][0],
                       flops / us[2][0] * 1e-6, us[2][rounds / 2] / us[0][rounds / 2], gmism);
      fflush(stdout);
    }
    hipFree(dx); hipFree(dw); hipFree(dy[0]); hipFree(dy[1]); hipFree(dadd); hipFree(dstat[0]); hipFree(dstat[1]);
  }
  printf("p8_bench: %s\n", bad_total ? "FAILED" : "all OK");
  return bad_total != 0;
}
