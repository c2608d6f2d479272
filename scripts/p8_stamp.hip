// (-DPSTAMPS=1|2 with -DP8_PSTAMPS=<same> -DG8_PSTAMPS=<same>: per-PHASE stamps of K steps 8 and 9 of a workgroup's first tile instead.)
// Diagnostic build (never timed as a whole): where conv_igemm_p8_kernel and the yardstick GEMM spend their cycles per segment - s_memtime
// stamps at the segment boundaries of a workgroup's first two tiles (cdna_hip_programming.md section 7, In-kernel stamps).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DP8_STAMP -DG8_STAMP scripts/p8_stamp.hip -o build/p8_stamp
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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double med(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  struct Shape { const char* name; int N, H, W, Cin, Cout; };
  std::vector<Shape> shapes = {{"gemm 4608->512 (32 x 64^2)", 32, 64, 64, 4608, 512}, {"gemm 2304->256 (32 x 64^2)", 32, 64, 64, 2304, 256},
                               {"gemm 1024->256 (32 x 64^2)", 32, 64, 64, 1024, 256}};
  auto f2bf = [](float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); };
  for (auto& s : shapes) {
    const int M = s.N * s.H * s.W;
    const size_t nx = (size_t)M * s.Cin, nw = (size_t)s.Cout * s.Cin, ny = (size_t)M * s.Cout;
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1234);
    for (auto& v : hx) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    for (auto& v : hw) v = f2bf((float)(rand() & 0xFFFFFF) / 8388608.0f - 1.0f);
    void *dx, *dw, *dy;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, ny * 2);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    const int tiles = ((M + 255) / 256) * (s.Cout / 256), nk = s.Cin / 64;
    unsigned long long* dbg;
    const size_t ndbg = (size_t)tiles * 2 * 16;
    hipMalloc(&dbg, ndbg * 8);
    ConvArgs a{};
    a.src = dx; a.wt = dw; a.dst = dy; a.bias = (const float*)dbg;
    a.N = s.N; a.Hs = s.H; a.Ws = s.W; a.Cs = s.Cin; a.lds = s.Cin;
    a.Hd = s.H; a.Wd = s.W; a.Cd = s.Cout; a.ldd = s.Cout;
    a.R = 1; a.S = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.mode = 0;
    a.M = M; a.Ktot = s.Cin;
    a.src_bytes = (unsigned)(nx * 2); a.wt_bytes = (unsigned)(nw * 2); a.dst_bytes = (unsigned)(ny * 2);
    for (int v = 0; v < 3; ++v) {      // 0: p8 persistent, 1: p8 one workgroup per tile, 2: gemm8p
      const int grid = v == 0 ? (tiles < 256 ? tiles : 256) : tiles;
      auto run = [&]() {
        if (v < 2) css_launch_conv_p8(a, grid, 0);
        else hipLaunchKernelGGL(gemm8p_kernel, dim3(tiles), dim3(512), 0, 0, dx, dw, dy, M, s.Cout, s.Cin, dbg);
      };
      for (int i = 0; i < 20; ++i) run();      // warm (clock / cache state of a steady stream of launches)
      hipMemset(dbg, 0, ndbg * 8);
      run();
      hipDeviceSynchronize();
#ifdef PSTAMPS
      std::vector<unsigned> h(ndbg * 2);
      hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
      for (int g = 0; g < 2; ++g)
        for (int ks = 0; ks < 2; ++ks) {
          std::vector<double> ph[4], ld[4], mf[4], tl[4];
          for (int b = 0; b < grid; ++b) {
            const unsigned* t = &h[((size_t)b * 2 + g) * 32 + ks * 16];
            if (!t[0]) continue;
            for (int p = 0; p < 4; ++p) {
              ph[p].push_back((double)(unsigned)(t[3 * p + 3] - t[3 * p]));
              if (PSTAMPS >= 2) {
                ld[p].push_back((double)(unsigned)(t[3 * p + 1] - t[3 * p]));
                mf[p].push_back((double)(unsigned)(t[3 * p + 2] - t[3 * p + 1]));
                tl[p].push_back((double)(unsigned)(t[3 * p + 3] - t[3 * p + 2]));
              }
            }
          }
          printf("%-28s %-26s group %d K step %d: phases %5.0f %5.0f %5.0f %5.0f = %6.0f", s.name, v == 0 ? "p8 persistent" : v == 1 ? "p8 one workgroup per tile" : "gemm8p",
                 g, 8 + ks, med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[0]) + med(ph[1]) + med(ph[2]) + med(ph[3]));
          if (PSTAMPS >= 2)
            printf("   load|mfma|tail: %4.0f|%4.0f|%4.0f  %4.0f|%4.0f|%4.0f  %4.0f|%4.0f|%4.0f  %4.0f|%4.0f|%4.0f", med(ld[0]), med(mf[0]), med(tl[0]), med(ld[1]), med(mf[1]),
                   med(tl[1]), med(ld[2]), med(mf[2]), med(tl[2]), med(ld[3]), med(mf[3]), med(tl[3]));
          printf("\n");
        }
#else
      std::vector<unsigned long long> h(ndbg);
      hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
      for (int g = 0; g < 2; ++g) {
        std::vector<double> pro, wait, loop0, epi0, loop1, epi1, total;
        for (int b = 0; b < grid; ++b) {
          const unsigned long long* t = &h[((size_t)b * 2 + g) * 8];
          if (!t[0]) continue;
          if (v < 2) {
            pro.push_back((double)(t[1] - t[0])); wait.push_back((double)(t[2] - t[1])); loop0.push_back((double)(t[3] - t[2]));
            epi0.push_back((double)(t[4] - t[3]));
            if (t[5]) { loop1.push_back((double)(t[5] - t[4])); epi1.push_back((double)(t[6] - t[5])); }
            total.push_back((double)(t[7] - t[0]));
          } else {
            pro.push_back((double)(t[1] - t[0])); wait.push_back((double)(t[2] - t[1])); loop0.push_back((double)(t[3] - t[2]));
            epi0.push_back((double)(t[4] - t[3])); total.push_back((double)(t[4] - t[0]));
          }
        }
        printf("%-28s %-26s wave group %d: prologue issue %7.0f  first wait %7.0f  K loop tile0 %8.0f (%6.1f / K step)  epilogue0 %7.0f", s.name,
               v == 0 ? "p8 persistent" : v == 1 ? "p8 one workgroup per tile" : "gemm8p", g, med(pro), med(wait), med(loop0), med(loop0) / nk, med(epi0));
        if (!loop1.empty()) printf("  K loop tile1 %8.0f (%6.1f / K step)  epilogue1 %7.0f", med(loop1), med(loop1) / nk, med(epi1));
        printf("  whole %9.0f ticks\n", med(total));
      }
#endif
    }
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dbg);
  }
  return 0;
}
