// conv3x3_c64_kernel: the 3x3, stride-1, pad-1 convolutions with 64 INPUT channels in bf16 - conv2 of every layer-1 Bottleneck
// (generalframeworks/networks/resnet.py:126-129: planes = 64) forward and in its data-gradient form, and the second and third convolution of
// the deep stem (resnet.py:177-190: 64 -> 64 and 64 -> 128 at half the image size) - with the unique input patch of a tile staged in LDS ONCE
// (VERDICT r02-r04 item "a streaming kernel for the Cout <= 128 layers").
//
// Why: the implicit-GEMM kernels gather the im2col matrix tap by tap - with K = 9 x 64 every input pixel travels nine times through the CU's
// vector-memory path for 64 channels of output, and conv_igemm_dma_kernel<128,64> runs these layers at 0.17 of the MFMA peak and 1.5 TB/s (neither
// roof: the fill path).  Here
//   * a workgroup (eight waves) walks tiles of 128 CONSECUTIVE output pixels (global row-major order: a tile may cross image rows and images, so
//     the 128-row statistics slab is the tile).  For tap row a the input pixels a tile needs are ONE contiguous range of 130 pixels (start = tile
//     start + (a - 1) W - 1): 3 x 17 LDS-DMA instructions of 8 pixels x 128 bytes per tile (whole cache lines), double-buffered (2 x 56 KiB); the
//     16-byte chunk c of range position p sits at chunk c ^ ((p >> 1) & 7) of LDS row p (conv_ws_kernel's swizzle: the 16 lanes of a fragment read
//     cover the 64 banks once, for every column tap b - the swizzle of p + b depends on the lane and b only: three lane registers);
//   * what the contiguous range gets wrong - the left / right zero padding of an image row and the rows above / below an image, which linear
//     addressing fills with the neighbouring row / image - is zeroed in the fragment registers by a per-lane 9-bit validity mask (skipped by waves
//     whose pixels are all interior);
//   * weight-stationary as conv_ws_kernel: a wave holds 32 output channels x all of K = 576 as MFMA operands (18 K blocks of 32 = (tap, channel
//     half): 144 VGPRs, loaded once per launch); Cout / 32 channel groups x 8 / (Cout / 32) pixel groups of waves; v_mfma_f32_16x16x32_bf16,
//     D[channel][pixel]: the register epilogue of conv_ws_kernel's round-4 form (bf16 -> v_permlane16_swap -> 16-byte stores);
//   * BN statistics: a wave owns 32 or 64 rows x 32 channels; the pixel groups park their 16-lane sums in LDS and the first adds them in group
//     order (one extra barrier per tile) - slabs of 128 rows as everywhere ([2 ceil(M / 256)][2][Cout] fp32).
// The data-gradient form is the same kernel on the dgrad weight layout [Cin][tap][Cout]: weight tap (r, s) reads the patch at (2 - r, 2 - s)
// (hs = hd + 1 - r).
// Numerics: bf16 products, fp32 accumulation over K in the order of the weights ((r, s, channel) ascending), one v_mfma_f32_16x16x32_bf16 per 32
// values of K - exactly the sequence of the gather kernels, so outputs and statistics slabs are BIT-IDENTICAL to theirs in both forms
// (tests/test_conv_c64_gpu.py compares with them - css_conv_c64_set_enabled(0) - and with torch-CPU).
#include "common.h"
#include "launchers.h"
#include <cstdlib>

namespace {
typedef __attribute__((address_space(3))) void c6_lds_void;
constexpr unsigned C6_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int c6_u32x4;
typedef __attribute__((ext_vector_type(4))) float c6_f32x4;
typedef __attribute__((ext_vector_type(2))) float c6_f32x2;

__device__ __forceinline__ void c6_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (c6_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ float c6_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ void c6_swap16(unsigned& a, unsigned& b) {      // (the builtin returns one register for both results: DESIGN.md 3)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float c6_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float c6_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <int N> __device__ __forceinline__ void c6_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

// NWC = Cout / 32 channel groups of waves (2 or 4).  grid = min(tiles, n_cu) workgroups of 512 threads.
template <int NWC, bool STATS, bool DGRAD>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const ConvArgs a) {
  constexpr int BT = 128, NWP = 8 / NWC, PXW = BT / NWP, NI = PXW / 16;        // pixel groups, pixels per wave, 16-pixel tiles per wave
  constexpr int NPI = 17, RB = NPI * 1024, NPIECE = 3 * NPI, NPWV = 7;      // 8-pixel pieces per tap row; a tap row in bytes; pieces per wave
  constexpr int BUF = 8 * NPWV * 1024;       // the patch (51 pieces) + 5 pieces of padding: every wave issues NPWV pieces per tile (a constant for the vmcnt
                                             // bookkeeping of the compiler and of this file), the five past the patch out of range into the padding
  constexpr int KB = 18;                                                        // K blocks of 32: (tap, channel half)
  constexpr int NST = NI + (STATS ? 4 : 0);                                     // stores of a tile and wave
  static_assert(NPWV * 8 >= NPIECE && 3 * RB <= BUF, "pieces per wave");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF + (STATS ? 8 * 64 * 4 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % NWC, pg = wave / NWC;
  const int l15 = lane & 15, lg = lane >> 4;
  const int W = a.Wd, H = a.Hd, M = a.M, Cd = 32 * NWC;
  const int ntiles = (M + BT - 1) / BT;
  // tile schedule: the workgroups of one XCD take CONSECUTIVE tiles (their tap rows overlap: the halo comes from that XCD's L2)
  const int G = gridDim.x, q8 = G >> 3, r8 = G & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int pos0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  if (pos0 >= ntiles) return;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  // ---- my weights: MFMA operand fragments for channels 32 cg + 16 j + (lane & 15), K block q = (weight tap q >> 1, channel half q & 1),
  // k = 8 (lane >> 4) .. + 7 of the block.  The K blocks are walked in the order of the weights (= the order of the gather kernels: results
  // bit-identical to theirs); in the data-gradient form weight tap (r, s) reads the patch at tap (2 - r, 2 - s) ----
  bf16x8 fw[KB][2];
#pragma unroll
  for (int q = 0; q < KB; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = ((unsigned)(32 * cg + 16 * j + l15) * 576u + (unsigned)(32 * q + 8 * lg)) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }

  // ---- issue side: piece p = 7 wave + i < 51 -> tap row p / 17, pixels 8 (p % 17) + (lane >> 3) of its range, source chunk (lane & 7) ^ ((pos >> 1) & 7) ----
  const unsigned lds2 = (unsigned)a.lds * 2u;
  auto issue_patch = [&](int t, int buf) {
    unsigned char* const base = smem + buf * BUF;
    const int m0 = t * BT;
#pragma unroll
    for (int i = 0; i < NPWV; ++i) {
      const int p = wave * NPWV + i;
      const int ar = p / NPI, c = p - NPI * ar;
      const int pos = 8 * c + (lane >> 3);
      const int P = m0 + (ar - 1) * W - 1 + pos;
      const bool ok = p < NPIECE && t < ntiles && P >= 0 && P < M;
#ifndef C6_ABL_NODMA      // (timing ablations for scripts/conv_bench.hip: C6_ABL_NODMA / _NOMFMA / _NOSTORE; results are garbage)
      c6_dma16(rs_a, base + p * 1024, ok ? (unsigned)P * lds2 + (unsigned)(((lane & 7) ^ ((pos >> 1) & 7)) * 16) : C6_OOB);
#else
      (void)base; (void)ok;
#endif
    }
  };
  int t = pos0;
  __builtin_amdgcn_sched_barrier(0);            // the weights first: their wait below then leaves the patch in flight
  issue_patch(t, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < KB; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fw[q][j]));      // the weights are needed from here on (one counted wait, see conv_ws.hip)

  // lane part of a fragment's LDS address for column tap b and channel half hh: range position (my pixel) + b, chunk 4 hh + lg swizzled
  int lb[3];              // (channel half 1: chunk + 4 = the same address with bit 6 flipped)
#pragma unroll
  for (int b = 0; b < 3; ++b) lb[b] = (l15 + b) * 128 + ((lg ^ (((l15 + b) >> 1) & 7)) << 4);
  f32x4 acc[NI][2];       // [pixel tile i: pixels 16 i + (lane & 15) of my PXW][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int nl = 32 * cg + 16 * (lg & 1) + 8 * (lg >> 1);           // first of my 8 channels in a store (after the lane swap)
  float* const carry = reinterpret_cast<float*>(smem + 2 * BUF);    // [wave][j][sums | squares][lg][4] floats
  int cur = 0;
  bool first = true;

  for (; t < ntiles; t += G) {
    const int m0 = t * BT, mw = m0 + PXW * pg;
    // ---- validity of my pixels: bit 3 a + b set <=> tap (a, b) reads inside the image ----
    unsigned vm[NI];
    bool interior = true;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = mw + 16 * i + l15;
      const int n = (int)fdiv((uint32_t)m, a.fd_hw), r = m - n * (H * W);
      const int y = (int)fdiv((uint32_t)r, a.fd_w), x = r - y * W;
      unsigned yb = 0, xb = 0;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        yb |= (unsigned)(y - 1 + e >= 0 && y - 1 + e < H) << e;
        xb |= (unsigned)(x - 1 + e >= 0 && x - 1 + e < W) << e;
      }
      unsigned v = 0;
#pragma unroll
      for (int e = 0; e < 3; ++e) v |= ((yb >> e) & 1u) ? (xb << (3 * e)) : 0u;
      if (m >= M) v = 0;
      interior = interior && v == 0x1FFu;
      vm[i] = v;
    }
    const bool all_in = __builtin_amdgcn_ballot_w64(!interior) == 0;
    __builtin_amdgcn_sched_barrier(0);
    if (first) c6_wait_vm<0>();                 // my pieces of this tile have landed (issued a tile ago: younger are that tile's NST stores)
    else c6_wait_vm<NST>();
    first = false;
    __builtin_amdgcn_s_barrier();               // everybody's pieces have landed; everybody is done with the other buffer
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_patch(t + G, cur ^ 1);                // the next tile's patch (past my last tile: out of range = zeros, no traffic)
    __builtin_amdgcn_sched_barrier(0);

    const unsigned char* const pb = smem + cur * BUF + (PXW * pg) * 128;
    // the fragments of K block q + 1 are requested BEFORE the MFMAs of block q (two sets of NI registers): with 4 or 8 MFMAs per block and
    // wave the LDS latency would otherwise be exposed once per block
    auto read_frags = [&](int q, bf16x8 (&fa)[NI]) {
      const int tp = DGRAD ? 8 - (q >> 1) : (q >> 1), ar = tp / 3, b = tp - 3 * ar, hh = q & 1;
#pragma unroll
      for (int i = 0; i < NI; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pb + (lb[b] ^ (hh << 6)) + ar * RB + i * 2048);
    };
    auto mask_frags = [&](int q, bf16x8 (&fa)[NI]) {
      const int tp = DGRAD ? 8 - (q >> 1) : (q >> 1);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bool ok = (vm[i] >> tp) & 1u;
        c6_u32x4 u = __builtin_bit_cast(c6_u32x4, fa[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = ok ? u[e] : 0u;
        fa[i] = __builtin_bit_cast(bf16x8, u);
      }
    };
    auto mfma_block = [&](int q, const bf16x8 (&fa)[NI]) {
#ifdef C6_ABL_NOMFMA
      if (q == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fa[i]));
      return;
#endif
      if (q == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], z, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], acc[i][j], 0, 0, 0);
      }
    };
    if constexpr (NI <= 2) {
      bf16x8 fa0[NI], fa1[NI];
      read_frags(0, fa0);
#pragma unroll
      for (int q = 0; q < KB; q += 2) {
        __builtin_amdgcn_sched_barrier(0);
        read_frags(q + 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        if (!all_in) mask_frags(q, fa0);
        mfma_block(q, fa0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < KB) read_frags(q + 2, fa0);
        __builtin_amdgcn_sched_barrier(0);
        if (!all_in) mask_frags(q + 1, fa1);
        mfma_block(q + 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {        // (four pixel tiles per wave: a second fragment set would spill; eight MFMAs per block cover more of the latency)
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        bf16x8 fa[NI];
        read_frags(q, fa);
        if (!all_in) mask_frags(q, fa);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(q, fa);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur ^= 1;

    // ---- epilogue: registers -> bf16 -> lane swap -> 16-byte stores (the statistics see the packed values) ----
    const int bnd = STATS ? (m0 / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;      // rows >= bnd: next statistics group (stage 2 sums them from the tensor)
    const bool whole = m0 + BT <= bnd;
    c6_f32x2 s01[2], s23[2], q01[2], q23[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s01[j] = s23[j] = q01[j] = q23[j] = c6_f32x2{0.f, 0.f};
    auto out_tiles = [&](bool test) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = mw + 16 * i + l15;
        unsigned lo[2], hi[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          lo[j] = pack2_bf16(acc[i][j][0], acc[i][j][1]);
          hi[j] = pack2_bf16(acc[i][j][2], acc[i][j][3]);
          if (STATS) {
            c6_f32x2 v01 = {c6_lo(lo[j]), c6_hi(lo[j])}, v23 = {c6_lo(hi[j]), c6_hi(hi[j])};
            if (test && !(m < bnd)) { v01 = c6_f32x2{0.f, 0.f}; v23 = c6_f32x2{0.f, 0.f}; }      // (rows >= M hold zeros already)
            s01[j] += v01; s23[j] += v23;
            q01[j] += v01 * v01; q23[j] += v23 * v23;
          }
        }
        c6_swap16(lo[0], lo[1]);
        c6_swap16(hi[0], hi[1]);
        const c6_u32x4 vv = {lo[0], hi[0], lo[1], hi[1]};
#ifdef C6_ABL_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b128(vv, rs_d, (int)C6_OOB, 0, 0);
#else
        __builtin_amdgcn_raw_buffer_store_b128(vv, rs_d, (int)(m < M ? ((unsigned)m * (unsigned)a.ldd + (unsigned)nl) * 2u : C6_OOB), 0, 0);
#endif
      }
    };
    if (!STATS || whole) out_tiles(false);
    else out_tiles(true);
    if (STATS) {
      // 16-lane sums of my PXW rows x 32 channels -> LDS; the first pixel group adds the groups in order and stores the slab's entry
      c6_f32x4 o[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        o[j][0] = c6_f32x4{c6_row16_sum(s01[j][0]), c6_row16_sum(s01[j][1]), c6_row16_sum(s23[j][0]), c6_row16_sum(s23[j][1])};
        o[j][1] = c6_f32x4{c6_row16_sum(q01[j][0]), c6_row16_sum(q01[j][1]), c6_row16_sum(q23[j][0]), c6_row16_sum(q23[j][1])};
      }
      // carry layout per wave: [j][sq][lg][4] floats = 64 floats: lane (l15 == 0, lg) writes its four channels
      float* const cw = carry + wave * 64;
      if (l15 == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int sq = 0; sq < 2; ++sq) *reinterpret_cast<c6_f32x4*>(cw + ((j * 2 + sq) * 4 + lg) * 4) = o[j][sq];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads(): its fence also waits for the next tile's LDS-DMA pieces)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(m0 >> 7) * 2u * (unsigned)Cd * 4u;
      const bool lane_ok = pg == 0 && l15 == 0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
          c6_f32x4 tsum = {0.f, 0.f, 0.f, 0.f};
          if (lane_ok) {
            tsum = *reinterpret_cast<const c6_f32x4*>(carry + cg * 64 + ((j * 2 + sq) * 4 + lg) * 4);
#pragma unroll
            for (int g2 = 1; g2 < NWP; ++g2)
              tsum += *reinterpret_cast<const c6_f32x4*>(carry + (g2 * NWC + cg) * 64 + ((j * 2 + sq) * 4 + lg) * 4);
          }
          const int n = sq * Cd + 32 * cg + 16 * j + 4 * lg;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c6_u32x4, tsum), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : C6_OOB), 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the ghost patch past my last tile must have landed before the LDS is released
}

// Shapes this kernel takes (everything else stays on the implicit-GEMM kernels); CSS_NO_C64_CONV=1 switches it off (A/B, tests/test_kernel_switches_gpu.py)
static int g_c64_off = -1;
static void c64_read_env() {
  if (g_c64_off < 0) {
    const char* e = getenv("CSS_NO_C64_CONV");
    g_c64_off = (e && *e && !(e[0] == '0' && !e[1])) ? 1 : 0;
  }
}
int css_conv_c64_set_enabled_(int on) {
  c64_read_env();
  const int was = g_c64_off ? 0 : 1;
  g_c64_off = on ? 0 : 1;
  return was;
}
bool css_conv_c64_supported(const ConvArgs& a, int dtype) {
  c64_read_env();
  if (g_c64_off || dtype != CSS_BF16) return false;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.dil != 1 || a.Hs != a.Hd || a.Ws != a.Wd) return false;
  if (a.Cs != 64 || a.Ktot != 576 || (a.Cd != 64 && a.Cd != 128) || a.addend || a.bias || a.m_begin != 0) return false;
  if (a.lds % 8 || a.ldd % 8 || (reinterpret_cast<uintptr_t>(a.src) & 15) || (reinterpret_cast<uintptr_t>(a.wt) & 15) || (reinterpret_cast<uintptr_t>(a.dst) & 15)) return false;
  if (a.M != a.N * a.Hd * a.Wd || a.M <= 0) return false;
  if ((size_t)a.M * a.lds * 2 >= 0x7FFFFFF0ull || (size_t)a.M * a.ldd * 2 >= 0x7FFFFFF0ull) return false;
  if (a.stats && a.stat_Mg < 128) return false;
  if (a.mode && (a.Cd != 64 || a.stats)) return false;      // the data-gradient form: 64 -> 64 only (the gradient of 64 -> 128 gathers 128 channels)
  return true;
}
void css_launch_conv_c64(ConvArgs a, int n_cu, hipStream_t st) {
  a.dst_bytes = (unsigned)((size_t)a.M * a.ldd * 2);
  if (a.stats) a.stat_bytes = (unsigned)((size_t)2 * cdiv(a.M, 256) * 2 * a.Cd * 4);
  a.fd_hw = make_fastdiv((uint32_t)(a.Hd * a.Wd));
  a.fd_w = make_fastdiv((uint32_t)a.Wd);
  const int ntiles = cdiv(a.M, 128);
  const dim3 g(ntiles < n_cu ? ntiles : n_cu), b(512);
  if (a.mode) hipLaunchKernelGGL((conv3x3_c64_kernel<2, false, true>), g, b, 0, st, a);
  else if (a.Cd == 64) {
    if (a.stats) hipLaunchKernelGGL((conv3x3_c64_kernel<2, true, false>), g, b, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<2, false, false>), g, b, 0, st, a);
  } else {
    if (a.stats) hipLaunchKernelGGL((conv3x3_c64_kernel<4, true, false>), g, b, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<4, false, false>), g, b, 0, st, a);
  }
}
