// Prototype (NOT product): C[M][N] = A[M][K] * B[N][K]^T in bf16 on gfx950 with FOUR waves per workgroup, 128x128 outputs per wave
// (256 accumulator registers, one wave per SIMD) - the structure of the vendor GEMMs - to measure its ceiling against the shipped
// 8-wave ping-pong kernel (conv_pp64.hip) on the GEMM-shaped (1x1) layers.  hipcc -O3 --offload-arch=gfx950 scripts/proto/gemm128.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;
#ifndef NSTAGE
#define NSTAGE 2
#endif
#ifndef USE32
#define USE32 0
#endif
#ifndef DMA_PER_GROUP
#define DMA_PER_GROUP 2      // LDS-DMA pieces issued behind every group of 8 MFMAs (16 pieces per stage and wave)
#endif
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, void* lds, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, (int)off, 0, 0, 0);
}
// tiles: 256 rows x 128 B (K = 64), chunk c of row r stored at c ^ ((r >> 1) & 7)
__global__ __launch_bounds__(256) void gemm128_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int M, int N,
                                                     int K, int tiles_m, int tiles_n) {
  constexpr int BUF = 256 * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * 2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((size_t)N * K * 2), 0x00020000);
  // DMA: wave w fills rows w*64 + 8 i + (lane >> 3), i = 0..7, of each operand tile; chunk position lane & 7
  const int prow = wave * 64 + (lane >> 3);
  const int cch0 = (lane & 7) ^ ((lane >> 4) & 3);
  const int sw = (l15 >> 1) & 7;
  const int koff[2] = {((lg ^ sw) << 4), (((4 + lg) ^ sw) << 4)};
  const int nk = K / 64;
  for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
    unsigned aoff[8], boff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cch = cch0 ^ ((i & 1) << 2);
      const int r = prow + 8 * i;
      aoff[i] = (m0 + r) < M ? (unsigned)(m0 + r) * (unsigned)K * 2u + cch * 16u : 0x80000000u;
      boff[i] = (n0 + r) < N ? (unsigned)(n0 + r) * (unsigned)K * 2u + cch * 16u : 0x80000000u;
    }
    auto piece = [&](int stage, int p, unsigned kb) {     // p = 0..15: A pieces then B pieces
      unsigned char* base = smem + stage * 2 * BUF + (p >= 8 ? BUF : 0) + wave * (64 * 128) + (p & 7) * 1024;
      const unsigned off = p >= 8 ? boff[p & 7] : aoff[p & 7];
      dma16(p >= 8 ? rs_b : rs_a, base, ((off | kb) & 0x80000000u) ? 0x80000000u : off + kb);
    };
#if USE32
    // 32x32x16: A/B fragment = 32 rows x 8 k per half-wave (k chunk = 2 kk + (lane >> 5)), D[32x32]: col = lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3)
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const int sw32 = (l31 >> 1) & 7;
#else
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) piece(0, p, 0);
    for (int ks = 0; ks < nk; ++ks) {
      const int st = ks % NSTAGE, nst = (ks + 1) % NSTAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned kb = ks + 1 < nk ? (unsigned)(ks + 1) * 128u : 0x80000000u;
#if USE32
      const unsigned char* ab = smem + st * 2 * BUF + (wm * 128 + l31) * 128;
      const unsigned char* bb = smem + st * 2 * BUF + BUF + (wn * 128 + l31) * 128;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 fa[4], fb[4];
        const int ko = (((2 * kk + lh) ^ sw32) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bb + j * 4096 + ko);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ab + i * 4096 + ko);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#ifndef NOLOAD
          piece(nst, kk * 4 + i, kb);
#endif
        }
      }
#else
      const unsigned char* ab = smem + st * 2 * BUF + (wm * 128 + l15) * 128;
      const unsigned char* bb = smem + st * 2 * BUF + BUF + (wn * 128 + l15) * 128;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 fa[8], fb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bb + j * 2048 + koff[h]);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + koff[h]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#ifndef NOLOAD
#pragma unroll
          for (int q = 0; q < DMA_PER_GROUP; ++q) {
            const int p = (h * 8 + i) * DMA_PER_GROUP + q;
            if (p < 16) piece(nst, p, kb);       // (past the last K step: offsets beyond the buffer = zeros into a stage nobody reads)
          }
#endif
        }
      }
#endif
    }
#if USE32
    // D[channel n = row][pixel m = col]: row = 8 (r >> 2) + 4 lh + (r & 3)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 128 + 32 * i + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 128 + 32 * j + 8 * q + 4 * lh;
          if (m < M && n < N) {
            union { __bf16 h[4]; uint2 u; } pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk.h[r] = (__bf16)acc[i][j][4 * q + r];
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk.u;
          }
        }
    }
#else
    // epilogue: D col = lane & 15 (pixel), row = 4 (lane >> 4) + reg (channel): 8-byte stores (prototype: not tuned)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + 16 * i + l15;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = n0 + wn * 128 + 16 * j + 4 * lg;
        if (m < M && n < N) {
          union { __bf16 h[4]; uint2 u; } pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk.h[r] = (__bf16)acc[i][j][r];
          *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk.u;
        }
      }
    }
#endif
  }
}
int main() {
  struct S { const char* name; int M, N, K; } shapes[] = {{"K2304 N256", 135168, 256, 2304}, {"K1024 N256", 135168, 256, 1024},
                                                          {"K256 N1024", 135168, 1024, 256}, {"K18432 N256", 135168, 256, 18432}};
  for (auto& s : shapes) {
    size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
    if (na * 2 >= 0x7FFFFFF0ull) { printf("%s: A too large for 32-bit offsets, skipped\n", s.name); continue; }
    std::vector<unsigned short> ha(na), hb(nb);
    for (auto& v : ha) v = 0x3C00 + (rand() & 0x3FF) - ((rand() & 1) << 15);
    for (auto& v : hb) v = 0x3800 + (rand() & 0x3FF) - ((rand() & 1) << 15);
    void *da, *db, *dc;
    hipMalloc(&da, na * 2); hipMalloc(&db, nb * 2); hipMalloc(&dc, nc * 2);
    hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice);
    const int tm = (s.M + 255) / 256, tn = (s.N + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm128_kernel, dim3(256), dim3(256), 0, 0, (const __bf16*)da, (const __bf16*)db, (__bf16*)dc, s.M, s.N, s.K, tm, tn);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm128_kernel, dim3(256), dim3(256), 0, 0, (const __bf16*)da, (const __bf16*)db, (__bf16*)dc, s.M, s.N, s.K, tm, tn);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // spot check of 64 outputs against a host dot product
    std::vector<unsigned short> hc(nc);
    hipMemcpy(hc.data(), dc, nc * 2, hipMemcpyDeviceToHost);
    auto f = [](unsigned short v) { union { unsigned u; float x; } c; c.u = (unsigned)v << 16; return c.x; };
    double maxrel = 0;
    for (int q = 0; q < 64; ++q) {
      const size_t m = (size_t)(rand() % s.M), n = (size_t)(rand() % s.N);
      double ref = 0;
      for (int k = 0; k < s.K; ++k) ref += (double)f(ha[m * s.K + k]) * f(hb[n * s.K + k]);
      const double got = f(hc[m * s.N + n]);
      const double rel = fabs(got - ref) / (fabs(ref) + 1e-3 * sqrt((double)s.K));
      if (rel > maxrel) maxrel = rel;
    }
    printf("%-12s %8.1f us  %7.1f TFLOP/s   spot-check max rel err %.3g\n", s.name, ms / reps * 1e3, 2.0 * s.M * s.N * s.K / (ms / reps * 1e-3) / 1e12, maxrel);
    hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
