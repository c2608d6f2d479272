// Round 5, VERDICT r04 item 2 ("measure first"): can the MFMA-bound and the HBM-bound kernels of backward use the chip SIDE BY SIDE when each chain
// gets a CU partition of its own (two streams from hipExtStreamCreateWithCUMask)?  Measurement infrastructure, not a product path.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/partition_bench.hip -o build/partition_bench && ./build/partition_bench
// The unit is the backward of ONE layer-3 Bottleneck at the bench's launch shapes (32 images of 65^2: M = 135200 rows, planes 256,
// /root/reference/generalframeworks/networks/resnet.py:119-139) with the library's own launchers, in the order ops.py queues it:
//   bn3 backward (reduce + stage 2 + apply, [M][1024], mask form)   conv3 dgrad (1x1 1024 -> 256)   conv3 wgrad
//   bn2 backward ([M][256])                                          conv2 dgrad (3x3 d2 256 -> 256)  conv2 wgrad
//   bn1 backward ([M][256])                                          conv1 dgrad (1x1 256 -> 1024 + residual addend: conv_ws_kernel)   conv1 wgrad
// A: everything on one full-chip stream (what ships).  B: two PLAIN streams, weight gradients on the second (round 4's experiment).  C: two
// CU-masked streams - batch norm + data gradients on X CUs, weight gradients on 256 - X, the persistent kernels' grids sized to their partition.
// CU mask bit b belongs to XCD b % 8 (the KFD spreads the bits round-robin over the XCCs), so the first X bits (X a multiple of 8) give every
// XCD X / 8 CUs: the kernels' blockIdx & 7 = XCD assumption holds inside a partition (printed: the XCC ids the workgroups of a probe kernel saw).
#include "../css_amd/csrc/conv.hip"
#include "../css_amd/csrc/conv_wgrad.hip"
#include "../css_amd/csrc/conv_pp.hip"
#include "../css_amd/csrc/conv_p8.hip"
#include "../css_amd/csrc/conv_ws.hip"
#include "../css_amd/csrc/conv_c64.hip"
#include "../css_amd/csrc/bn.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void xcc_probe_kernel(unsigned* out) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
}

struct Block {
  int M, P;               // rows, planes
  void *a0, *y1, *a1, *y2, *a2, *y3, *da3, *dy3, *da2, *dy2, *da1, *dy1, *dx;      // activations / gradients (bf16)
  unsigned char* mask3;
  void *w1t, *w2t, *w3t;  // dgrad-layout weights
  float *dw1, *dw2, *dw3, *ws1, *ws2, *ws3;
  size_t wsb1, wsb2, wsb3;
  float *mean, *invstd, *scale, *shift, *gamma, *dgamma, *dbeta;
  double *partial, *sums;
};

static void bn_bwd(const Block& b, const void* da, const void* a, const void* y, void* dy, int C, const unsigned char* mask, hipStream_t st) {
  const int G = 2, Mg = b.M / G;
  const int nrb = css_bn_nrb_(Mg, G, C, CSS_BF16);
  css_launch_bn_bwd_reduce(da, C, mask ? nullptr : a, mask ? 0 : C, y, C, b.mean, b.invstd, mask ? nullptr : b.scale, mask ? nullptr : b.shift, Mg, G, C, 1,
                           b.partial, mask, CSS_BF16, st);
  css_launch_bn_reduce(b.partial, nrb, C, G, b.sums, b.dgamma, b.dbeta, 1, 0.0, st);
  css_launch_bn_bwd_apply(da, C, mask ? nullptr : a, mask ? 0 : C, y, C, dy, C, nullptr, 0, b.mean, b.invstd, b.gamma, b.sums, mask ? nullptr : b.scale,
                          mask ? nullptr : b.shift, (double)Mg, nullptr, b.M, C, 1, Mg, mask, CSS_BF16, st);
}
static void dgrad(const Block& b, const void* dy, int Cout, const void* wt, void* dx, int Cin, int R, int dil, const void* addend, int n_cu, hipStream_t st) {
  ConvArgs a{};
  a.src = dy; a.wt = wt; a.dst = dx;
  a.N = 32; a.Hs = 65; a.Ws = 65; a.Cs = Cout; a.lds = Cout;
  a.Hd = 65; a.Wd = 65; a.Cd = Cin; a.ldd = Cin;
  a.R = R; a.S = R; a.stride = 1; a.pad = R == 3 ? dil : 0; a.dil = dil; a.mode = 1;
  a.M = b.M; a.Ktot = R * R * Cout;
  if (addend) { a.addend = addend; a.ld_add = Cin; }
  css_launch_conv(a, CSS_BF16, n_cu, st);
}
static void wgrad(const Block& b, const void* x, int Cin, const void* dy, int Cout, float* dw, float* ws, size_t wsb, int R, int dil, int n_cu, hipStream_t st) {
  WgradArgs g{};
  g.x = x; g.dy = dy; g.dw = dw; g.N = 32; g.Hs = 65; g.Ws = 65; g.Cs = Cin; g.ldx = Cin; g.Hd = 65; g.Wd = 65; g.Cd = Cout; g.ldy = Cout;
  g.R = R; g.S = R; g.stride = 1; g.pad = R == 3 ? dil : 0; g.dil = dil; g.M = b.M; g.Ktot = R * R * Cin; g.m_per_split = b.M;
  g.ws = ws; g.ws_bytes = wsb;
  css_launch_wgrad(g, CSS_BF16, n_cu, st);
}

// one Bottleneck backward.  sm: stream of batch norm + data gradients, sw: stream of the weight gradients (may be the same); n_m / n_w: the CU
// counts the persistent kernels of each stream size their grids for; ev: three events for the hand-over main -> side
static void block_bwd(const Block& b, hipStream_t sm, hipStream_t sw, int n_m, int n_w, hipEvent_t* ev) {
  const bool two = sm != sw;
  bn_bwd(b, b.da3, nullptr, b.y3, b.dy3, 4 * b.P, b.mask3, sm);
  if (two) { CK(hipEventRecord(ev[0], sm)); CK(hipStreamWaitEvent(sw, ev[0], 0)); }
  dgrad(b, b.dy3, 4 * b.P, b.w3t, b.da2, b.P, 1, 1, nullptr, n_m, sm);
  wgrad(b, b.a2, b.P, b.dy3, 4 * b.P, b.dw3, b.ws3, b.wsb3, 1, 1, n_w, sw);
  bn_bwd(b, b.da2, b.a2, b.y2, b.dy2, b.P, nullptr, sm);
  if (two) { CK(hipEventRecord(ev[1], sm)); CK(hipStreamWaitEvent(sw, ev[1], 0)); }
  dgrad(b, b.dy2, b.P, b.w2t, b.da1, b.P, 3, 2, nullptr, n_m, sm);
  wgrad(b, b.a1, b.P, b.dy2, b.P, b.dw2, b.ws2, b.wsb2, 3, 2, n_w, sw);
  bn_bwd(b, b.da1, b.a1, b.y1, b.dy1, b.P, nullptr, sm);
  if (two) { CK(hipEventRecord(ev[2], sm)); CK(hipStreamWaitEvent(sw, ev[2], 0)); }
  dgrad(b, b.dy1, b.P, b.w1t, b.dx, 4 * b.P, 1, 1, b.da3, n_m, sm);
  wgrad(b, b.a0, 4 * b.P, b.dy1, b.P, b.dw1, b.ws1, b.wsb1, 1, 1, n_w, sw);
}

static void fill_bf16(void* d, size_t n, unsigned seed, unsigned short base) {
  std::vector<unsigned short> h(n);
  unsigned s = seed;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = base + ((s >> 10) & 0x3FF) - (((s >> 25) & 1) << 15); }
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}

int main() {
  const int M = 32 * 65 * 65, P = 256;
  Block b{};
  b.M = M; b.P = P;
  auto A = [&](void** p, size_t elems, unsigned seed) { CK(hipMalloc(p, elems * 2)); fill_bf16(*p, elems, seed, 0x3C00); };
  A(&b.a0, (size_t)M * 4 * P, 1); A(&b.y1, (size_t)M * P, 2); A(&b.a1, (size_t)M * P, 3); A(&b.y2, (size_t)M * P, 4); A(&b.a2, (size_t)M * P, 5);
  A(&b.y3, (size_t)M * 4 * P, 6); A(&b.da3, (size_t)M * 4 * P, 7); A(&b.dy3, (size_t)M * 4 * P, 8); A(&b.da2, (size_t)M * P, 9);
  A(&b.dy2, (size_t)M * P, 10); A(&b.da1, (size_t)M * P, 11); A(&b.dy1, (size_t)M * P, 12); A(&b.dx, (size_t)M * 4 * P, 13);
  CK(hipMalloc((void**)&b.mask3, (size_t)M * 4 * P / 8)); CK(hipMemset(b.mask3, 0xA5, (size_t)M * 4 * P / 8));
  CK(hipMalloc(&b.w1t, (size_t)4 * P * P * 2)); fill_bf16(b.w1t, (size_t)4 * P * P, 21, 0x3800);
  CK(hipMalloc(&b.w2t, (size_t)9 * P * P * 2)); fill_bf16(b.w2t, (size_t)9 * P * P, 22, 0x3800);
  CK(hipMalloc(&b.w3t, (size_t)4 * P * P * 2)); fill_bf16(b.w3t, (size_t)4 * P * P, 23, 0x3800);
  CK(hipMalloc((void**)&b.dw1, (size_t)4 * P * P * 4)); CK(hipMalloc((void**)&b.dw2, (size_t)9 * P * P * 4)); CK(hipMalloc((void**)&b.dw3, (size_t)4 * P * P * 4));
  CK(hipMemset(b.dw1, 0, (size_t)4 * P * P * 4)); CK(hipMemset(b.dw2, 0, (size_t)9 * P * P * 4)); CK(hipMemset(b.dw3, 0, (size_t)4 * P * P * 4));
  float* fl;
  CK(hipMalloc((void**)&fl, 7 * 2 * 1024 * 4));
  std::vector<float> hf(7 * 2048);
  for (size_t i = 0; i < hf.size(); ++i) hf[i] = 0.5f + 0.001f * (float)(i % 97);
  CK(hipMemcpy(fl, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  b.mean = fl; b.invstd = fl + 2048; b.scale = fl + 4096; b.shift = fl + 6144; b.gamma = fl + 8192; b.dgamma = fl + 10240; b.dbeta = fl + 12288;
  CK(hipMalloc((void**)&b.partial, (size_t)2 * 4096 * 2 * 1024 * 8)); CK(hipMalloc((void**)&b.sums, 2 * 2 * 1024 * 8));

  hipEvent_t e0, e1, ev[3], join;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&join));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int reps = getenv("PB_REPS") ? atoi(getenv("PB_REPS")) : 12;

  auto set_ws = [&](int n_w) {      // the slab workspaces depend on the CU count the weight-gradient plan is made for
    for (float** p : {&b.ws1, &b.ws2, &b.ws3}) if (*p) { CK(hipFree(*p)); *p = nullptr; }
    b.wsb1 = css_wgrad_ws_bytes_(M, 4 * P, P, CSS_BF16, n_w); b.wsb2 = css_wgrad_ws_bytes_(M, 9 * P, P, CSS_BF16, n_w); b.wsb3 = css_wgrad_ws_bytes_(M, P, 4 * P, CSS_BF16, n_w);
    if (b.wsb1) CK(hipMalloc((void**)&b.ws1, b.wsb1));
    if (b.wsb2) CK(hipMalloc((void**)&b.ws2, b.wsb2));
    if (b.wsb3) CK(hipMalloc((void**)&b.ws3, b.wsb3));
  };
  auto run = [&](const char* name, hipStream_t sm, hipStream_t sw, int n_m, int n_w) {
    set_ws(n_w);
    double best = 1e30, sum = 0;
    for (int round = 0; round < 4; ++round) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) block_bwd(b, sm, sw, n_m, n_w, ev);
      if (sm != sw) { CK(hipEventRecord(join, sw)); CK(hipStreamWaitEvent(sm, join, 0)); }
      CK(hipStreamSynchronize(sm));
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      if (round) { best = us < best ? us : best; sum += us; }
    }
    printf("%-64s %8.1f us per block backward (best of 3; mean %8.1f)\n", name, best, sum / 3);
    fflush(stdout);
  };
  // per-kernel-class times on the full chip (one stream), for the record
  auto time_one = [&](const char* name, auto fn) {
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    for (int i = 0; i < 3; ++i) fn(s0);
    CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(e0, s0));
    for (int i = 0; i < 10; ++i) fn(s0);
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-40s %8.1f us\n", name, ms * 100.f);
    CK(hipStreamDestroy(s0));
  };
  set_ws(256);
  printf("== the parts on the full chip ==\n");
  time_one("bn3 backward [M][1024] (mask)", [&](hipStream_t s) { bn_bwd(b, b.da3, nullptr, b.y3, b.dy3, 4 * P, b.mask3, s); });
  time_one("bn2 backward [M][256]", [&](hipStream_t s) { bn_bwd(b, b.da2, b.a2, b.y2, b.dy2, P, nullptr, s); });
  time_one("conv3 dgrad 1x1 1024->256", [&](hipStream_t s) { dgrad(b, b.dy3, 4 * P, b.w3t, b.da2, P, 1, 1, nullptr, 256, s); });
  time_one("conv2 dgrad 3x3 d2", [&](hipStream_t s) { dgrad(b, b.dy2, P, b.w2t, b.da1, P, 3, 2, nullptr, 256, s); });
  time_one("conv1 dgrad 1x1 256->1024 + addend (ws)", [&](hipStream_t s) { dgrad(b, b.dy1, P, b.w1t, b.dx, 4 * P, 1, 1, b.da3, 256, s); });
  time_one("conv3 wgrad", [&](hipStream_t s) { wgrad(b, b.a2, P, b.dy3, 4 * P, b.dw3, b.ws3, b.wsb3, 1, 1, 256, s); });
  time_one("conv2 wgrad", [&](hipStream_t s) { wgrad(b, b.a1, P, b.dy2, P, b.dw2, b.ws2, b.wsb2, 3, 2, 256, s); });
  time_one("conv1 wgrad", [&](hipStream_t s) { wgrad(b, b.a0, 4 * P, b.dy1, P, b.dw1, b.ws1, b.wsb1, 1, 1, 256, s); });

  printf("== one layer-3 Bottleneck backward, %d repetitions per timing ==\n", reps);
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  run("A  one stream, full chip (ships)", s1, s1, 256, 256);
  run("B  two plain streams (weight gradients on the second)", s1, s2, 256, 256);
  run("A  one stream, full chip (again)", s1, s1, 256, 256);
  unsigned* probe;
  CK(hipMalloc((void**)&probe, 4096 * 4));
  // PB_X=<CUs of the main partition>: ONE split per process (the first run of this harness did not come back from its second split - the
  // streams of a split are created once and never destroyed here, and the caller bounds every process with `timeout`)
  const int X = getenv("PB_X") ? atoi(getenv("PB_X")) : 192;     // (conv_ws_kernel needs (X / 8) % 4 == 0 for its four panels)
  {
    unsigned mm[8] = {0}, mw[8] = {0};
    for (int bit = 0; bit < 256; ++bit) (bit < X ? mm : mw)[bit >> 5] |= 1u << (bit & 31);
    hipStream_t sm, sw;
    CK(hipExtStreamCreateWithCUMask(&sm, 8, mm));
    CK(hipExtStreamCreateWithCUMask(&sw, 8, mw));
    printf("   masked streams created (X = %d)\n", X); fflush(stdout);
    // which XCDs does each partition reach?
    for (int w = 0; w < 2; ++w) {
      CK(hipMemset(probe, 0xFF, 4096 * 4));
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(2048), dim3(64), 0, w ? sw : sm, probe);
      CK(hipDeviceSynchronize());
      unsigned h[2048], cnt[16] = {0}, rr = 0;
      CK(hipMemcpy(h, probe, 2048 * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < 2048; ++i) { cnt[h[i] & 15]++; if (i < 64 && (h[i] & 7) == (unsigned)(i & 7)) ++rr; }
      printf("   X = %d, %s partition: workgroups per XCC", X, w ? "side" : "main");
      for (int i = 0; i < 8; ++i) printf(" %u", cnt[i]);
      printf("; blockIdx & 7 == XCC for %u of the first 64\n", rr);
      fflush(stdout);
    }
    char name[128];
    snprintf(name, sizeof name, "C  masked streams: BN + dgrad on %d CUs, wgrad on %d CUs", X, 256 - X);
    run(name, sm, sw, X, 256 - X);
    snprintf(name, sizeof name, "C' masked, but everything on the %d-CU stream", X);
    run(name, sm, sm, X, X);
    snprintf(name, sizeof name, "C\" masked, but everything on the %d-CU stream", 256 - X);
    run(name, sw, sw, 256 - X, 256 - X);
  }
  run("A  one stream, full chip (last)", s1, s1, 256, 256);
  return 0;
}
