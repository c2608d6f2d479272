// Weight-gradient harness (round 6): the Cout >= 256 weight-gradient launches of one c2 / c4 step on random data, for the MFMA-shape A/B of
// the 256 x 256 kernel (conv_wgrad_p8_kernel: v_mfma_f32_32x32x16_bf16, conv_wgrad_p16_kernel: v_mfma_f32_16x16x32_bf16).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/wgrad_bench.hip -o build/wgrad_bench          (timing)
//   ... -DWG_STAMP -o build/wgrad_stamp                                                                                    (in-kernel clock; never timed)
//   WB_MFMA=16|32 build/wgrad_bench          time that shape (A/B: alternate processes);  WB_CHECK=1: run BOTH, compare against an fp64 host
//   reference on a sub-sample of dW and against each other;  WB_WL=c4: the 769^2 deep-stem shapes;  WB_REPS (20)
#include "../css_amd/csrc/conv_wgrad.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern int css_wgrad_mfma_override_, css_wgrad_no_compact_override_;
struct Shape { const char* name; int N, H, Cin, Cout, R, dil, count; };
static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const bool c4 = getenv("WB_WL") && !strcmp(getenv("WB_WL"), "c4");
  const int NB = c4 ? 16 : 32, H8 = c4 ? 97 : 65, H4 = c4 ? 193 : 129;
  // (count = launches of this shape per step of the tv-R101 student; only the shapes the 256 x 256 kernel takes: Cout >= 256 and K >= 256)
  std::vector<Shape> shapes = {
      {"l3 1x1 1024->256", NB, H8, 1024, 256, 1, 1, 22}, {"l3 3x3 d2 256->256", NB, H8, 256, 256, 3, 2, 22}, {"l3 1x1 256->1024", NB, H8, 256, 1024, 1, 1, 23},
      {"l4 1x1 2048->512", NB, H8, 2048, 512, 1, 1, 2},  {"l4 3x3 d4 512->512", NB, H8, 512, 512, 3, 4, 2},   {"l4 1x1 512->2048", NB, H8, 512, 2048, 1, 1, 3},
      {"l4 ds 1024->2048", NB, H8, 1024, 2048, 1, 1, 1}, {"aspp 1x1 2048->256", NB, H8, 2048, 256, 1, 1, 1},  {"aspp 3x3 d12 2048->256", NB, H8, 2048, 256, 3, 12, 1},
      {"aspp 3x3 d24 2048->256", NB, H8, 2048, 256, 3, 24, 1}, {"aspp 3x3 d36 2048->256", NB, H8, 2048, 256, 3, 36, 1},
      {"aspp proj 1280->256", NB, H8, 1280, 256, 1, 1, 1}, {"head 3x3 304->256", NB, H4, 304, 256, 3, 1, 2},  {"l2 1x1 512->... l3 ds 512->1024", NB, H8, 512, 1024, 1, 1, 1},
  };
  const int reps = getenv("WB_REPS") ? atoi(getenv("WB_REPS")) : 20;
  const bool check = getenv("WB_CHECK") != nullptr;
  const int mf = getenv("WB_MFMA") ? atoi(getenv("WB_MFMA")) : 16;
  const int only = getenv("WB_ONLY") ? atoi(getenv("WB_ONLY")) : -1;
  double tot_us = 0, tot_fl = 0;
  unsigned long long* stamps = nullptr;
#ifdef WG_STAMP
  hipMalloc((void**)&stamps, 4096 * 4 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(wg_stamp_buf), &stamps, sizeof(stamps));
#endif
  for (auto& s : shapes) {
    if (only >= 0 && &s - shapes.data() != only) continue;
    const int pad = s.dil * (s.R / 2), Ho = s.H, M = s.N * Ho * Ho, K = s.R * s.R * s.Cin;
    const size_t nx = (size_t)s.N * s.H * s.H * s.Cin, ny = (size_t)M * s.Cout, nw = (size_t)s.Cout * K;
    std::vector<unsigned short> hx(nx), hy(ny);
    unsigned rs = 1234567u;                                                        // xorshift32: rand() took ~10 s per shape on 277 M elements
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; };
    for (auto& v : hx) { const unsigned r = rnd(); v = 0x3C00 + (r & 0x3FF) - (((r >> 20) & 1) << 15); }     // random bf16 bit patterns around +-1 (as conv_bench.hip)
    for (auto& v : hy) { const unsigned r = rnd(); v = 0x3800 + (r & 0x3FF) - (((r >> 20) & 1) << 15); }
    void *dx, *dy;
    float* dw;
    hipMalloc(&dx, nx * 2); hipMalloc(&dy, ny * 2); hipMalloc((void**)&dw, nw * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice);
    WgradArgs g{};
    g.x = dx; g.dy = dy; g.dw = dw; g.N = s.N; g.Hs = s.H; g.Ws = s.H; g.Cs = s.Cin; g.ldx = s.Cin; g.Hd = Ho; g.Wd = Ho; g.Cd = s.Cout;
    g.ldy = s.Cout; g.R = s.R; g.S = s.R; g.stride = 1; g.pad = pad; g.dil = s.dil; g.M = M; g.Ktot = K; g.m_per_split = M;
    g.ws_bytes = css_wgrad_ws_bytes_(M, K, s.Cout, CSS_BF16, 256);
    if (g.ws_bytes) hipMalloc((void**)&g.ws, g.ws_bytes);
    const double flops = 2.0 * M * s.Cout * K;
    if (check) {
      std::vector<float> out[2];
      const bool ckc = getenv("WB_CHECK_COMPACT") != nullptr;        // compare live-row compaction off / on (32x32x16) instead of the two MFMA shapes
      for (int v = 0; v < 2; ++v) {
        css_wgrad_mfma_override_ = ckc ? 32 : (v ? 16 : 32);
        css_wgrad_no_compact_override_ = ckc ? (v ? 0 : 1) : 0;
        hipMemset(dw, 0, nw * 4);
        css_launch_wgrad(g, CSS_BF16, 256, 0);
        out[v].resize(nw);
        hipMemcpy(out[v].data(), dw, nw * 4, hipMemcpyDeviceToHost);
      }
      double d2 = 0, n2 = 0, dmax = 0;
      for (size_t i = 0; i < nw; ++i) { const double d = (double)out[0][i] - out[1][i]; d2 += d * d; n2 += (double)out[0][i] * out[0][i]; dmax = std::max(dmax, std::fabs(d)); }
      // fp64 host reference on 64 sampled (cout, k) entries
      double worst[2] = {0, 0}, scale = 0;
      for (int smp = 0; smp < 64; ++smp) {
        const int n = (smp * 7919 + 13) % s.Cout, k = (int)(((size_t)smp * 104729 + 7) % K);
        const int tap = k / s.Cin, c = k % s.Cin, tr = tap / s.R, ts = tap % s.R;
        double acc = 0;
        for (int m = 0; m < M; ++m) {
          const int img = m / (Ho * Ho), rem = m % (Ho * Ho), ho = rem / Ho, wo = rem % Ho;
          const int hs = ho + tr * s.dil - pad, ws = wo + ts * s.dil - pad;
          if (hs < 0 || hs >= s.H || ws < 0 || ws >= s.H) continue;
          acc += (double)bf2f(hy[(size_t)m * s.Cout + n]) * bf2f(hx[(((size_t)img * s.H + hs) * s.H + ws) * s.Cin + c]);
        }
        scale = std::max(scale, std::fabs(acc));
        for (int v = 0; v < 2; ++v) worst[v] = std::max(worst[v], std::fabs(out[v][(size_t)n * K + k] - acc));
      }
      printf("%-34s CHECK %s: rel-L2 %.2e max %.2e | vs fp64 (64 samples, scale %.1f): first %.2e  second %.2e\n", s.name,
             ckc ? "compaction off vs on (32x32x16)" : "32x32x16 vs 16x16x32", std::sqrt(d2 / n2), dmax, scale, worst[0] / scale, worst[1] / scale);
    } else {
      css_wgrad_mfma_override_ = mf;
      css_wgrad_no_compact_override_ = getenv("WB_NOCOMPACT") ? 1 : 0;
      for (int i = 0; i < 3; ++i) css_launch_wgrad(g, CSS_BF16, 256, 0);
      hipDeviceSynchronize();
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; ++i) css_launch_wgrad(g, CSS_BF16, 256, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms / reps * 1e3;
      int splits, mps;
      css_wgrad_plan_(M, K, s.Cout, CSS_BF16, 256, &splits, &mps);
      printf("%-34s mfma %d%s  M=%d K=%d N=%d  slices %d  %8.1f us  %7.1f TFLOP/s  x%d", s.name, mf, getenv("WB_NOCOMPACT") ? " nocompact" : "", M, K, s.Cout, splits, us, flops / us / 1e6, s.count);
      tot_us += us * s.count; tot_fl += flops * s.count;
#ifdef WG_STAMP
      {
        const int nwg = std::min(4096, (int)(cdiv(K, 256) * cdiv(s.Cout, 256) * splits));
        std::vector<unsigned long long> hs(nwg * 4);
        hipMemcpy(hs.data(), stamps, nwg * 32, hipMemcpyDeviceToHost);
        std::vector<double> clk, cyc;
        for (int i = 0; i < nwg; ++i) {
          const double dc = (double)(hs[4 * i + 2] - hs[4 * i]), dr = (double)(hs[4 * i + 3] - hs[4 * i + 1]);
          if (dr > 0) { clk.push_back(dc / dr * 0.1); cyc.push_back(dc); }
        }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        if (!clk.empty()) printf("  | K loop: median clock %.3f GHz, median %.0f cycles (%d workgroups)", clk[clk.size() / 2], cyc[cyc.size() / 2], (int)clk.size());
      }
#endif
      printf("\n");
    }
    hipFree(dx); hipFree(dy); hipFree(dw);
    if (g.ws) hipFree(g.ws);
  }
  if (!check) printf("TOTAL mfma %d%s: %.1f us per step over these launches, %.1f TFLOP/s\n", mf, getenv("WB_NOCOMPACT") ? " nocompact" : "", tot_us, tot_fl / tot_us / 1e6);
  return 0;
}
