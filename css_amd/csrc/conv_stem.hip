// conv_stem_s2d_kernel: the stride-2 stem convolution of the backbone - torchvision's 7x7 s2 p3 3 -> 64 (mix_label.py:68: models.resnet101();
// SURVEY N4) and the first 3x3 s2 p1 3 -> 64 of the deep stem (generalframeworks/networks/resnet.py:177-190) - as a stride-1 convolution on the
// SPACE-TO-DEPTH image, with the unique input patch of a tile staged in LDS ONCE ("a streaming kernel for the stems").
//
// Why: the generic implicit-GEMM kernels gather the im2col matrix tap by tap.  With 3 input channels (one 16-byte vector per pixel, 5 of 8 lanes
// padding) the 7x7 stem is K = 49 x 8 = 392 of which 147 are real, every input pixel travels ~12 times through the vector-memory path, and the
// launch takes 451 us against a ~70 us HBM floor (135 MB in, 270 MB out at 513^2, 32 images).  Here
//   * the image is staged as [N][Hs][Ws][16] bf16, Hs = ceil(H / 2): s2d pixel (ys, xs) holds the 2 x 2 x 3 values of image pixels
//     (2 ys + py, 2 xs + px) at channel (2 py + px) 3 + c, channels 12..15 zero - HALF the bytes of the 8-channel staging;
//   * out(y, x) = sum over a, b < TA of  w2[a][b][.] . s2d[y - TA/2 + a][x - TA/2 + b][.]   with TA = 4 (7x7) or 2 (3x3): image row
//     2 y - P + ky = 2 (y - TA/2 + a) + py  <=>  ky = 2 a + py - 1 (P = R / 2), the same for columns; taps with ky or kx outside [0, R) carry
//     zero weights.  K = TA^2 x 16 = 256 (64): the 7x7 stem is EXACTLY conv_ws_kernel's K = 256 class - weight-stationary, 128 VGPRs of
//     weights per wave as MFMA operands, v_mfma_f32_16x16x32_bf16, the same register epilogue and statistics slabs;
//   * a workgroup (four waves, 64 pixels x 64 channels each) walks tiles of 256 consecutive output pixels (global row-major order: a tile may
//     cross image rows and images).  For tap row `a` the pixels a tile needs are ONE contiguous range of 256 + TA - 1 s2d pixels of the input
//     (start = tile start + (a - TA/2) Ws - TA/2): 2 TA plane-rows of 256 pixels (the two 16-byte halves of a pixel in planes of their own:
//     a fragment read is 16 consecutive 16-byte chunks - no bank conflict) + one shared 1-KiB tail piece = 33 (9) LDS-DMA instructions per
//     tile, double-buffered (2 x 33 KiB: two workgroups per CU); what the contiguous range gets wrong - the left / right zero padding of an
//     image row and the rows above / below an image, which linear addressing fills with the neighbouring row / image - is zeroed in the
//     fragment registers by a per-lane 16-bit validity mask (skipped by waves whose 64 pixels are all interior);
//   * a K block of 32 = two horizontally adjacent s2d pixels; 8 (2) K blocks per tile, 128 (32) MFMAs per wave, nothing re-staged.
// Numerics: bf16 products, fp32 accumulation in the order (a, b) ascending - not the (ky, kx) order of the gather kernels (results agree to fp32
// rounding of a 147-term sum; tests/test_conv_stem_gpu.py compares with torch-CPU).  A non-finite input value next to a zero-weight tap (kx = 7)
// gives NaN where the gather kernels would not read it; a finite image has none.
#include "common.h"
#include "launchers.h"
#include <cstdlib>

#define ST_GRID_STRIDE(idx, total) \
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)
static inline int st_ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

namespace {
typedef __attribute__((address_space(3))) void st_lds_void;
constexpr unsigned ST_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int st_u32x4;
typedef __attribute__((ext_vector_type(4))) float st_f32x4;
typedef __attribute__((ext_vector_type(2))) float st_f32x2;

__device__ __forceinline__ void st_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (st_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ float st_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ void st_swap16(unsigned& a, unsigned& b) {      // (the builtin returns one register for both results: DESIGN.md 3)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float st_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float st_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <int N> __device__ __forceinline__ void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct StemArgs {
  const void* src;       // [N][Hs][Ws][16] bf16 (space-to-depth image)
  const void* wt;        // [64][TA][TA][16] bf16
  void* dst;             // [M][64] bf16, M = N Hs Ws
  float* stats;          // [2 ceil(M / 256)][2][64] fp32 slabs or null
  int Hs, Ws, M, stat_Mg;
  unsigned src_bytes, dst_bytes, stat_bytes;
  FastDiv fd_hw, fd_w;
};

constexpr int STEM_TILE_ROWS = 256;   // pixels per tile = rows per pair of statistics slabs (css_conv2d_stem_s2d_tile_rows reports it to the host)

// grid = 2 n_cu workgroups of 256 threads; tile t = blockIdx.x, blockIdx.x + gridDim.x, ...
template <int TA, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_stem_s2d_kernel(const StemArgs a) {
  constexpr int OFF = TA / 2, KB = TA * TA / 2, BT = STEM_TILE_ROWS, NPR = 2 * TA;      // K blocks of 32; plane-rows (tap row, 16-byte half)
  constexpr int BUF = NPR * 4096 + 1024;                                   // plane-rows of 256 pixels x 16 bytes + the shared tail piece
  constexpr int NPW = NPR;                                                 // full pieces per wave and tile (4 NPR pieces / 4 waves)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF + (STATS ? 2048 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int ntiles = (a.M + BT - 1) / BT;
  if ((int)blockIdx.x >= ntiles) return;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, 64 * KB * 32 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  // ---- weights: MFMA operand fragments for channels 16 j + (lane & 15), k = 32 q + 8 (lane >> 4) .. + 7 (every wave holds all 64 channels) ----
  bf16x8 fw[KB][4];
#pragma unroll
  for (int q = 0; q < KB; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = (unsigned)(16 * j + l15) * (unsigned)(KB * 32 * 2) + (unsigned)(32 * q + 8 * lg) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }

  // ---- issue side: the patch of tile t into buffer `buf`.  Piece c of plane-row pr = (tap row ar, half h): lanes = pixels 64 c .. 64 c + 63 of the
  // tap row's range; wave w issues the pieces 4 NPW-chunk..: piece index p = wave NPW + i -> (pr = p >> 2, c = p & 3).  The tail piece (wave 0):
  // lane l -> plane-row l >> 3, pixel 256 + (l & 7). ----
  auto issue_patch = [&](int t, int buf) {
    unsigned char* const base = smem + buf * BUF;
    const int m0 = t * BT;             // (pixel indices fit an int: 32 M < 2^31 is checked by the launcher; negative = above the first image)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = wave * NPW + i, pr = p >> 2, c = p & 3;
      const int ar = pr >> 1, h = pr & 1;
      const int P = m0 + (ar - OFF) * a.Ws - OFF + 64 * c + lane;
      const bool ok = t < ntiles && P >= 0 && P < a.M;
      st_dma16(rs_a, base + pr * 4096 + c * 1024, ok ? (unsigned)P * 32u + (unsigned)h * 16u : ST_OOB);
    }
    if (wave == 0) {
      const int pr = lane >> 3, ar = pr >> 1, h = pr & 1;
      const int P = m0 + (ar - OFF) * a.Ws - OFF + 256 + (lane & 7);
      const bool ok = t < ntiles && pr < NPR && P >= 0 && P < a.M;
      st_dma16(rs_a, base + NPR * 4096, ok ? (unsigned)P * 32u + (unsigned)h * 16u : ST_OOB);
    }
  };
  int t = blockIdx.x;
  issue_patch(t, 0);
#pragma unroll
  for (int q = 0; q < KB; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fw[q][j]));      // the weights are needed from here on (one counted wait, see conv_ws.hip)

  f32x4 acc[4][4];        // [pixel tile i: pixels 16 i + (lane & 15) of my 64][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int nl = 16 * (lg & 1) + 8 * (lg >> 1);            // first of my 8 channels in a store of channel-tile pair 0 (after the lane swap)
  const int lbase = (l15 + (lg >> 1)) * 16 + (lg & 1) * 4096;      // lane part of a fragment's LDS address: pixel l15 + (lg >> 1), half plane lg & 1
  float* const carry = reinterpret_cast<float*>(smem + 2 * BUF) + (wave * 4 + lg) * 32;             // [j][os 4 | oq 4]: this wave's 64-row half of its slab
  int cur = 0;
  bool first = true;

  for (; t < ntiles; t += gridDim.x) {
    const int m0 = t * BT, mw = m0 + 64 * wave;
    // ---- validity of my four pixels: bit 4 a + b set <=> tap (a, b) reads inside the image ----
    unsigned vm[2] = {0u, 0u};                   // pixel tiles 0 | 1 and 2 | 3, sixteen bits each
    bool interior = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mw + 16 * i + l15;
      const int n = (int)fdiv((uint32_t)m, a.fd_hw), r = m - n * (a.Hs * a.Ws);
      const int y = (int)fdiv((uint32_t)r, a.fd_w), x = r - y * a.Ws;
      unsigned yb = 0, xb = 0;
#pragma unroll
      for (int e = 0; e < TA; ++e) {
        yb |= (unsigned)(y - OFF + e >= 0 && y - OFF + e < a.Hs) << e;
        xb |= (unsigned)(x - OFF + e >= 0 && x - OFF + e < a.Ws) << e;
      }
      unsigned v = 0;
#pragma unroll
      for (int e = 0; e < TA; ++e) v |= ((yb >> e) & 1u) ? (xb << (4 * e)) : 0u;
      if (m >= a.M) v = 0;
      interior = interior && v == (TA == 4 ? 0xFFFFu : 0x33u);
      vm[i >> 1] |= ((v >> (lg >> 1)) & 0xFFFFu) << (16 * (i & 1));        // (my column tap inside a K block is b = 2 bb + (lg >> 1))
    }
    const bool all_in = __builtin_amdgcn_ballot_w64(!interior) == 0;
    __builtin_amdgcn_sched_barrier(0);
    if (first) st_wait_vm<0>();                 // my pieces of this tile have landed (issued a tile ago: younger are that tile's 8 stores)
    else st_wait_vm<8>();
    first = false;
    __builtin_amdgcn_s_barrier();               // everybody's pieces have landed; everybody is done with the other buffer
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_patch(t + (int)gridDim.x, cur ^ 1);   // the next tile's patch (past my last tile: out of range = zeros, no traffic)
    __builtin_amdgcn_sched_barrier(0);

    const unsigned char* const pb = smem + cur * BUF + lbase;
    // byte offset of pixel position po (0 .. 255 + TA - 1) inside a plane-row: lane part (lbase) + wave-uniform part; only the LAST pixel tile of
    // the LAST wave can reach the shared tail piece (po >= 256)
#pragma unroll
    for (int q = 0; q < KB; ++q) {
      const int ar = TA == 4 ? q >> 1 : q, bb = TA == 4 ? q & 1 : 0;
      bf16x8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int off = ar * 8192 + (64 * wave + 16 * i + 2 * bb) * 16;
        if (i == 3 && wave == 3) {
          const int po = 240 + l15 + 2 * bb + (lg >> 1);
          if (po >= 256) off = NPR * 4096 + ((2 * ar + (lg & 1)) * 8 + (po - 256)) * 16 - lbase;
        }
        fa[i] = *reinterpret_cast<const bf16x8*>(pb + off);
      }
      if (!all_in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = (vm[i >> 1] >> (16 * (i & 1) + 4 * ar + 2 * bb)) & 1u;
          st_u32x4 u = __builtin_bit_cast(st_u32x4, fa[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = ok ? u[e] : 0u;
          fa[i] = __builtin_bit_cast(bf16x8, u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], z, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur ^= 1;

    // ---- epilogue (conv_ws_kernel's): registers -> bf16 -> lane swap -> 16-byte stores, 8 per wave ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mw + 16 * i + l15;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        unsigned lo0 = pack2_bf16(acc[i][2 * pr][0], acc[i][2 * pr][1]), hi0 = pack2_bf16(acc[i][2 * pr][2], acc[i][2 * pr][3]);
        unsigned lo1 = pack2_bf16(acc[i][2 * pr + 1][0], acc[i][2 * pr + 1][1]), hi1 = pack2_bf16(acc[i][2 * pr + 1][2], acc[i][2 * pr + 1][3]);
        st_swap16(lo0, lo1);
        st_swap16(hi0, hi1);
        st_u32x4 vv = {lo0, hi0, lo1, hi1};
        __builtin_amdgcn_raw_buffer_store_b128(vv, rs_d, (int)(m < a.M ? ((unsigned)m * 64u + (unsigned)(nl + 32 * pr)) * 2u : ST_OOB), 0, 0);
      }
    }
    if (STATS) {
      // sums of the bf16-ROUNDED outputs per 128-row slab (two waves: the odd one parks its 16-lane sums in LDS, the even one adds and stores);
      // a slab holds the rows of the statistics group of ITS first row (stage 2 sums the rows past a group boundary from the tensor)
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const int slab0 = m0 + 128 * (wave >> 1);
      const int bnd = (slab0 / a.stat_Mg + 1) * a.stat_Mg;
      const bool whole = mw + 64 <= bnd;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st_f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        auto accum = [&](bool test) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 tv = acc[i][j];
            asm volatile("" : "+v"(tv));       // opaque: otherwise the packed values of the store loop stay alive (CSE) across the epilogue
            const unsigned lo = pack2_bf16(tv[0], tv[1]), hi = pack2_bf16(tv[2], tv[3]);
            st_f32x2 v01 = {st_lo(lo), st_hi(lo)}, v23 = {st_lo(hi), st_hi(hi)};
            if (test && !(mw + 16 * i + l15 < bnd)) { v01 = st_f32x2{0.f, 0.f}; v23 = st_f32x2{0.f, 0.f}; }   // (rows >= M hold zeros already)
            s01 += v01; s23 += v23;
            q01 += v01 * v01; q23 += v23 * v23;
          }
        };
        if (whole) accum(false);
        else accum(true);
        const st_f32x4 os = {st_row16_sum(s01[0]), st_row16_sum(s01[1]), st_row16_sum(s23[0]), st_row16_sum(s23[1])};
        const st_f32x4 oq = {st_row16_sum(q01[0]), st_row16_sum(q01[1]), st_row16_sum(q23[0]), st_row16_sum(q23[1])};
        if (l15 == 0) {                        // both halves of a slab park their 16-lane sums; the even wave adds them in (first, second) order
          *reinterpret_cast<st_f32x4*>(carry + j * 8) = os;
          *reinterpret_cast<st_f32x4*>(carry + j * 8 + 4) = oq;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (not __syncthreads(): its fence also waits for the next tile's LDS-DMA pieces)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(wave & 1)) {
        const unsigned base = (unsigned)(slab0 >> 7) * 2u * 64u * 4u;
        const bool lane_ok = l15 == 0 && slab0 < a.M;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const st_f32x4 ts = *reinterpret_cast<const st_f32x4*>(carry + j * 8) + *reinterpret_cast<const st_f32x4*>(carry + 4 * 32 + j * 8);
          const st_f32x4 tq = *reinterpret_cast<const st_f32x4*>(carry + j * 8 + 4) + *reinterpret_cast<const st_f32x4*>(carry + 4 * 32 + j * 8 + 4);
          const int n = 16 * j + 4 * lg;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st_u32x4, ts), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : ST_OOB), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st_u32x4, tq), rs_s, (int)(lane_ok ? base + (unsigned)(64 + n) * 4u : ST_OOB), 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the ghost patch past my last tile must have landed before the LDS is released
}

// ---- fp32 NCHW image [N][3][H][W] -> bf16 space-to-depth [N][Hs][Ws][16]: thread = one s2d pixel, two 16-byte stores ----
__global__ __launch_bounds__(256) void nchw_to_s2d_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int N, int C, int H, int W, int Hs,
                                                          int Ws) {
  const size_t total = (size_t)N * Hs * Ws;
  ST_GRID_STRIDE(p, total) {
    const int n = (int)(p / ((size_t)Hs * Ws));
    const int r = (int)(p - (size_t)n * Hs * Ws);
    const int ys = r / Ws, xs = r - ys * Ws;
    Vec16<bf16_t> o0, o1;
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      const int pos = ch / 3, c = ch - 3 * pos;
      const int iy = 2 * ys + (pos >> 1), ix = 2 * xs + (pos & 1);
      const float v = (ch < 12 && c < C && iy < H && ix < W) ? x[(((size_t)n * C + c) * H + iy) * W + ix] : 0.f;
      if (ch < 8) o0.set(ch, v);
      else o1.set(ch - 8, v);
    }
    o0.store(out + p * 16);
    o1.store(out + p * 16 + 8);
  }
}
// ---- fp32 master weights [64][R][R][3] -> bf16 [64][TA][TA][16]: w2[a][b][(2 py + px) 3 + c] = w[2 a + py - 1][2 b + px - 1][c] (0 outside) ----
__global__ __launch_bounds__(256) void stem_s2d_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int R, int TA) {
  const int total = Cout * TA * TA * 16;
  ST_GRID_STRIDE(idx, (size_t)total) {
    const int ch = (int)(idx & 15), tb = (int)(idx >> 4);
    const int b = tb % TA, ta = (tb / TA) % TA, co = tb / (TA * TA);
    const int pos = ch / 3, c = ch - 3 * pos;
    const int ky = 2 * ta + (pos >> 1) - 1, kx = 2 * b + (pos & 1) - 1;
    const bool ok = ch < 12 && ky >= 0 && ky < R && kx >= 0 && kx < R;
    out[idx] = (bf16_t)(ok ? w[((size_t)(co * R + ky) * R + kx) * 3 + c] : 0.f);
  }
}
// ---- the weight gradient computed in s2d space, dw2 fp32 [64][TA][TA][16], folded back: dw[co][ky][kx][c] += dw2[co][a][b][(2 py + px) 3 + c] ----
__global__ __launch_bounds__(256) void stem_s2d_fold_wgrad_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int Cout, int R, int TA) {
  const int total = Cout * R * R * 3;
  ST_GRID_STRIDE(idx, (size_t)total) {
    const int c = (int)(idx % 3), t = (int)(idx / 3);
    const int kx = t % R, ky = (t / R) % R, co = t / (R * R);
    const int ta = (ky + 1) >> 1, py = (ky + 1) & 1, b = (kx + 1) >> 1, px = (kx + 1) & 1;
    dw[idx] += dw2[((size_t)(co * TA + ta) * TA + b) * 16 + (2 * py + px) * 3 + c];
  }
}
}  // namespace

static int g_stem_off = -1;      // -1: read CSS_NO_STEM_S2D on first use
int css_stem_s2d_tile_rows_() { return STEM_TILE_ROWS; }
int css_stem_s2d_enabled_() {
  if (g_stem_off < 0) {
    const char* e = getenv("CSS_NO_STEM_S2D");
    g_stem_off = (e && e[0] && e[0] != '0') ? 1 : 0;        // (CSS_NO_STEM_S2D=0 means "not switched off": scripts/ab_env.sh passes 0 / 1)
  }
  return g_stem_off ? 0 : 1;
}
int css_launch_nchw_to_s2d(const float* x, void* out, int N, int C, int H, int W, hipStream_t st) {
  if (N <= 0 || C < 1 || C > 3 || H < 1 || W < 1 || (reinterpret_cast<uintptr_t>(out) & 15)) return CSS_ERR_ARG;
  const int Hs = (H + 1) / 2, Ws = (W + 1) / 2;
  hipLaunchKernelGGL(nchw_to_s2d_kernel, dim3(st_ew_grid((size_t)N * Hs * Ws)), dim3(256), 0, st, x, (bf16_t*)out, N, C, H, W, Hs, Ws);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_stem_s2d_weights(const float* w, void* out, int Cout, int R, hipStream_t st) {
  if ((R != 7 && R != 3) || Cout != 64) return CSS_ERR_ARG;
  const int TA = R == 7 ? 4 : 2;
  hipLaunchKernelGGL(stem_s2d_weight_kernel, dim3(cdiv(Cout * TA * TA * 16, 256)), dim3(256), 0, st, w, (bf16_t*)out, Cout, R, TA);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_stem_s2d_fold_wgrad(const float* dw2, float* dw, int Cout, int R, hipStream_t st) {
  if ((R != 7 && R != 3) || Cout != 64) return CSS_ERR_ARG;
  hipLaunchKernelGGL(stem_s2d_fold_wgrad_kernel, dim3(cdiv(Cout * R * R * 3, 256)), dim3(256), 0, st, dw2, dw, Cout, R, R == 7 ? 4 : 2);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_conv_stem_s2d(const void* x, const void* w2, void* y, float* stats, int Mg, int N, int Hs, int Ws, int Cout, int R, int n_cu,
                             hipStream_t st) {
  if ((R != 7 && R != 3) || Cout != 64 || N <= 0 || Hs <= 0 || Ws <= 0) return CSS_ERR_ARG;
  const size_t M = (size_t)N * Hs * Ws;
  if (M * 128 >= 0x7FFFFFF0ull) return CSS_ERR_ARG;                       // 32-bit buffer offsets (output rows of 128 bytes)
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w2) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return CSS_ERR_ARG;
  if (stats && (Mg < 128 || M % (size_t)Mg)) return CSS_ERR_ARG;
  StemArgs a{};
  a.src = x; a.wt = w2; a.dst = y; a.stats = stats;
  a.Hs = Hs; a.Ws = Ws; a.M = (int)M; a.stat_Mg = stats ? Mg : (int)M;
  a.src_bytes = (unsigned)(M * 32); a.dst_bytes = (unsigned)(M * 128);
  a.stat_bytes = (unsigned)((size_t)2 * cdiv((int)M, 256) * 2 * 64 * 4);
  a.fd_hw = make_fastdiv((uint32_t)(Hs * Ws));
  a.fd_w = make_fastdiv((uint32_t)Ws);
  const int ntiles = cdiv((int)M, 256);
  const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
  if (R == 7) {
    if (stats) hipLaunchKernelGGL((conv_stem_s2d_kernel<4, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_stem_s2d_kernel<4, false>), dim3(grid), dim3(256), 0, st, a);
  } else {
    if (stats) hipLaunchKernelGGL((conv_stem_s2d_kernel<2, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_stem_s2d_kernel<2, false>), dim3(grid), dim3(256), 0, st, a);
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
