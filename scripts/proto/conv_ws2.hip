// PROTOTYPE for the next round (not part of libcss_hip.so, not measured yet: written after this round's GPU budget was spent).
//
// conv_ws2_kernel: conv_ws_kernel (css_amd/csrc/conv_ws.hip) with 64-pixel tiles under a 512-channel panel, K = 256.
// Why (profiles/r02_ws_kernel.txt section 2): the 256 -> 1024 call of conv_ws_kernel is bound by the CU's vector-memory
// path, which fill and stores share (64 KiB in + 64 KiB out per 128 x 256 tile in 5.6 us = 23 GB/s per CU; stores + fill alone 92 us of
// the call's 111).  A wave that owns 64 output channels instead of 32
//   * halves the fill per output byte (a 64-pixel stage of 8 KiB feeds 512 channels: 32 KiB in per 64 KiB out),
//   * writes whole 128-byte lines (64 channels x 2 bytes per pixel) instead of half lines,
//   * halves the LDS fragment reads per MFMA (8 ds_read_b128 per 32 MFMAs instead of 16),
// at the price of 128 registers of weights per wave (K = 256 only; no room for the addend form) and of statistics slabs of 64 rows
// (stage 2 would have to learn them: the STATS instance below writes them, the harness checks pairs of them against conv_ws_kernel's
// 128-row slabs; 254 VGPRs with statistics, 238 without, no scratch).
// Expected from the byte model: stores + fill 92 -> ~69 us, the 256 -> 1024 call 111 -> 85-90 us.
//
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics scripts/proto/ws2_bench.hip -o build/ws2_bench
#include "../../css_amd/csrc/common.h"
#include "../../css_amd/csrc/launchers.h"

namespace {
typedef __attribute__((address_space(3))) void w2_lds_void;
constexpr unsigned W2_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int w2_u32x4;
typedef __attribute__((ext_vector_type(4))) float w2_f32x4;
typedef __attribute__((ext_vector_type(2))) float w2_f32x2;
__device__ __forceinline__ float w2_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}
__device__ __forceinline__ float w2_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float w2_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void w2_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (w2_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ void w2_swap16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ unsigned w2_pack2(float lo, float hi) {
  union { bf16_t h[2]; unsigned u; } t;
  t.h[0] = (bf16_t)lo;
  t.h[1] = (bf16_t)hi;
  return t.u;
}
template <int N> __device__ __forceinline__ void w2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

// grid = n_cu workgroups of 512 threads, (n_cu / 8) % (Cd / 512) == 0; K = Cs = 256.
// STATS: batch-norm statistics slabs of 64 ROWS ([ceil(M / 64)][2][Cd] fp32: slab p = rows 64 p .. 64 p + 63, holding the rows that lie in the
// statistics group of its first row) - css_bn_reduce_finalize_slabs would have to learn that slab height (tile_rows = 128).
template <bool STATS>
__global__ __launch_bounds__(512) void conv_ws2_kernel(const ConvArgs a) {
  constexpr int KS = 4, BM = 64, BN = 512, NT = 3, LA = NT * KS, NS = LA + 2;
  constexpr int STG = BM * 128;                                  // one stage: 64 pixels x 128 bytes (64 channels) = 8 KiB, one piece per wave
  constexpr int NST = 8 + (STATS ? 8 : 0);                       // stores of a tile per wave
  constexpr int W0 = LA - 1, W1 = W0 + NST, W2 = W1 + NST, W3 = W2 + NST;   // vmcnt of the stage wait in tiles 0, 1, 2, later ones
  static_assert(W3 <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int G = gridDim.x, c8 = G >> 3;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int np = a.Cd / BN, spx = c8 / np;
  const int panel = idx8 % np, stream = xcd * spx + idx8 / np, nstreams = 8 * spx;
  const int mt_total = (a.M + BM - 1) / BM;
  const int nmy = stream < mt_total ? (mt_total - stream + nstreams - 1) / nstreams : 0;
  if (nmy == 0) return;
  const int n0w = panel * BN + wave * 64;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  bf16x8 fw[2 * KS][4];                                          // 128 VGPRs: channels n0w + 16 j + (lane & 15), k = 32 q + 8 (lane >> 4) .. + 7
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = (unsigned)(n0w + 16 * j + l15) * (unsigned)a.Ktot * 2u + (unsigned)(32 * q + 8 * lg) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }

  // issue side: stage = 8 pieces of 1 KiB, ONE per wave: rows 8 wave + (lane >> 3); chunk at position lane & 7 of row r = source chunk
  // (lane & 7) ^ ((r >> 1) & 7), (r >> 1) & 7 = (4 wave + (lane >> 4)) & 7
  const int prow = wave * 8 + (lane >> 3);
  const int cch = (lane & 7) ^ ((((wave & 1) << 2) | ((lane >> 4) & 3)));
  const unsigned lds2 = (unsigned)a.lds * 2u;
  int it_ti = 0, it_k = 0, islot = 0;
  unsigned rowoff;
  auto issue_rows = [&]() {
    const int m = (stream + it_ti * nstreams) * BM + prow;
    rowoff = (it_ti < nmy && m < a.M) ? (unsigned)m * lds2 + (unsigned)(cch * 16) : W2_OOB;
  };
  auto issue_stage = [&]() {
    w2_dma16(rs_a, smem + islot * STG + wave * 1024, rowoff != W2_OOB ? rowoff + (unsigned)(it_k * 128) : W2_OOB);
    islot = islot == NS - 1 ? 0 : islot + 1;
    if (++it_k == KS) {
      it_k = 0;
      ++it_ti;
      issue_rows();
    }
  };
  issue_rows();
#pragma unroll
  for (int s = 0; s < LA; ++s) issue_stage();
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fw[q][j]));

  f32x4 acc[4][4];        // [pixel tile i: pixels 16 i + (lane & 15)][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int sw = (l15 >> 1) & 7;
  int cslot = 0;
  const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);

  for (int ti = 0; ti < nmy; ++ti) {
    const int m0 = (stream + ti * nstreams) * BM;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      if (ti >= 3) w2_wait_vm<W3>();
      else if (ti == 2) w2_wait_vm<W2>();
      else if (ti == 1) w2_wait_vm<W1>();
      else w2_wait_vm<W0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_stage();
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* ab = smem + cslot * STG + l15 * 128;
      bf16x8 fa[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[h][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + (((4 * h + lg) ^ sw) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (k == 0 && h == 0) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2 * k + h][j], fa[h][i], z, 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[2 * k + h][j], fa[h][i], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      cslot = cslot == NS - 1 ? 0 : cslot + 1;
    }
    // epilogue: 128 contiguous bytes per pixel and wave (two 16-byte stores per lane and pixel tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 16 * i + l15;
      const unsigned rowb = (unsigned)m * (unsigned)a.ldd * 2u;
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        unsigned lo0 = w2_pack2(acc[i][jp][0], acc[i][jp][1]), hi0 = w2_pack2(acc[i][jp][2], acc[i][jp][3]);
        unsigned lo1 = w2_pack2(acc[i][jp + 1][0], acc[i][jp + 1][1]), hi1 = w2_pack2(acc[i][jp + 1][2], acc[i][jp + 1][3]);
        w2_swap16(lo0, lo1);
        w2_swap16(hi0, hi1);
        const w2_u32x4 v = {lo0, hi0, lo1, hi1};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m < a.M ? rowb + (unsigned)(nl + 16 * jp) * 2u : W2_OOB), 0, 0);
      }
    }
    if (STATS) {
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(m0 >> 6) * 2u * (unsigned)a.Cd * 4u;
      const int bnd = (m0 / a.stat_Mg + 1) * a.stat_Mg;
      const bool whole = m0 + BM <= bnd;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w2_f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        auto accum = [&](bool test) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 t = acc[i][j];
            asm volatile("" : "+v"(t));
            const unsigned lo = w2_pack2(t[0], t[1]), hi = w2_pack2(t[2], t[3]);
            w2_f32x2 v01 = {w2_lo(lo), w2_hi(lo)}, v23 = {w2_lo(hi), w2_hi(hi)};
            if (test && !(m0 + 16 * i + l15 < bnd)) { v01 = w2_f32x2{0.f, 0.f}; v23 = w2_f32x2{0.f, 0.f}; }
            s01 += v01; s23 += v23;
            q01 += v01 * v01; q23 += v23 * v23;
          }
        };
        if (whole) accum(false);
        else accum(true);
        const w2_f32x4 os = {w2_row16_sum(s01[0]), w2_row16_sum(s01[1]), w2_row16_sum(s23[0]), w2_row16_sum(s23[1])};
        const w2_f32x4 oq = {w2_row16_sum(q01[0]), w2_row16_sum(q01[1]), w2_row16_sum(q23[0]), w2_row16_sum(q23[1])};
        const int n = n0w + 16 * j + 4 * lg;
        const bool lane_ok = l15 == 0;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w2_u32x4, os), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : W2_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w2_u32x4, oq), rs_s, (int)(lane_ok ? base + (unsigned)(a.Cd + n) * 4u : W2_OOB), 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool css_conv_ws2_supported(const ConvArgs& a, int n_cu) {
  if (a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0 || a.Hs != a.Hd || a.Ws != a.Wd || a.bias || a.addend) return false;
  if (a.Ktot != a.Cs || a.Cs != 256 || a.Cd < 512 || a.Cd % 512 || a.lds % 8 || a.ldd % 8) return false;
  const int np = a.Cd / 512;
  if (n_cu < 8 || n_cu % 8 || (n_cu / 8) % np) return false;
  return a.M > 0 && (size_t)a.M * a.ldd * 2 < 0x7FFFFFF0ull;
}
void css_launch_conv_ws2(ConvArgs a, int n_cu, hipStream_t st) {
  a.dst_bytes = (unsigned)((size_t)a.M * a.ldd * 2);
  if (a.stats) {
    a.stat_bytes = (unsigned)((size_t)cdiv(a.M, 64) * 2 * a.Cd * 4);          // 64-row slabs
    hipLaunchKernelGGL((conv_ws2_kernel<true>), dim3(n_cu), dim3(512), 0, st, a);
  } else {
    hipLaunchKernelGGL((conv_ws2_kernel<false>), dim3(n_cu), dim3(512), 0, st, a);
  }
}
