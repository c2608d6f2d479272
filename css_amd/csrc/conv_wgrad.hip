// Weight gradient of the NHWC convolutions on gfx950 (CDNA4): dW[n][k] += sum_m dY[m][n] * X(m, k), k = (r, s, c).
//   conv_wgrad_kernel            : 128x128 (bf16) / 64x64 (fp32) tiles, register-staged (narrow layers, the fp32 parity path)
//   conv_wgrad_p8_kernel<MF16>   : 256x256 tiles, LDS-DMA, two-phase steps, live-row compaction for dilated 3x3 (Cout >= 256: the bulk of the network)
//   wgrad_slab_reduce*_kernel    : every pixel slice stores its partial tile as a slab; these add the slabs in slice order (bit-reproducible)
// Replaces the weight-gradient half of the cuDNN backward the reference reaches through nn.Conv2d in
//   generalframeworks/networks/resnet.py:119-139, deeplabv3/aspp.py:17-72, deeplabv3/deeplabv3.py:115-133.
#include "common.h"
#include "launchers.h"
#include <cstdlib>


// --------------------------------------------------------------------------
// Weight gradient: dW[n][k] += sum_m dY[m][n] * X(m, k)     (k = (r, s, c))
// Both operands are contiguous along the NON-reduced index in memory, so the bf16 path
// keeps the tiles as loaded ([pixel][channel]) and reads MFMA fragments with the gfx950
// transposing LDS read (ds_read_b64_tr_b16); the fp32 path uses ds_read_b32.
// --------------------------------------------------------------------------
template <typename T> struct WgFrag;
template <> struct WgFrag<bf16_t> {
  static constexpr int KS = 16;
  typedef bf16x8 frag;
  // tile[kk][col] with row stride RS (elements). Operand element (idx = lane&31, kk = 8*(lane>>5)+j).
  static __device__ __forceinline__ frag load(const bf16_t* tile, int RS, int kk0, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const bf16_t* ad = tile + (kk0 + 8 * (g >> 1) + q) * RS + col0 + 16 * (g & 1) + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ad));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ad + 4 * RS));
    union { struct { s16x4 a, b; } s; frag f; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.f;
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct WgFrag<float> {
  static constexpr int KS = 2;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* tile, int RS, int kk0, int col0, int lane) {
    return tile[(kk0 + (lane >> 5)) * RS + col0 + (lane & 31)];
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// Workgroup -> (pixel slice zz, weight tile t).  Workgroups are dealt round-robin over the 8 XCDs (blockIdx & 7); XCD x takes the
// CONTIGUOUS range [x W8, (x + 1) W8) of the work items in slice-major order (W8 = ceil(slices x tiles / 8)), so the tiles of one pixel slice
// run at the same time on ONE XCD and its slice of dY / X is fetched into a single L2 (measured before: 3.7x over-fetch) - a slice that
// straddles two ranges is fetched by two.  Round 4: the number of slices is no longer a multiple of 8 (css_wgrad_plan_), so a launch fills
// whole rounds of the CHIP, not of every XCD: 9 tiles x 28 slices = 252 workgroups in one round instead of 9 x 56 in two.
// Live-row compaction with two classes of kernel rows (WgradArgs::compact == 2): class c holds nc = tiles_n x cls_nrows[c] x (tiles_k / R) tiles per
// slice; XCD x takes [x W8c, (x + 1) W8c) of each class's slice-major list, the long class first.
__device__ __forceinline__ bool wgrad_work_item(const WgradArgs& a, int per_z, int& zz, int& t) {
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
  if (a.compact == 2) {
    const int tpr = a.tiles_k / a.R;
    const int n0c = a.tiles_n * a.cls_nrows[0] * tpr, n1c = a.tiles_n * a.cls_nrows[1] * tpr;
    const int w80 = (n0c * a.splits + 7) >> 3, w81 = (n1c * a.splits + 7) >> 3;
    const int c = j8 < w80 ? 0 : 1;
    const int jj = c ? j8 - w80 : j8, w8c = c ? w81 : w80, nc = c ? n1c : n0c;
    const int w = xcd * w8c + jj;
    if (jj >= w8c || w >= nc * a.splits) return false;
    zz = w / nc;
    const int rem = w - zz * nc, per_nt = a.cls_nrows[c] * tpr;
    const int nt = rem / per_nt, rem2 = rem - nt * per_nt, ri = rem2 / tpr;
    t = nt * a.tiles_k + a.cls_rows[c][ri] * tpr + (rem2 - ri * tpr);
    return true;
  }
  const int total = per_z * a.splits, w8 = (total + 7) >> 3;
  const int w = xcd * w8 + j8;
  if (j8 >= w8 || w >= total) return false;
  zz = w / per_z;
  t = w - zz * per_z;
  return true;
}

// BN_: output-channel tile, BKC: k-column tile, BP: pixels per iteration
template <typename T, int BN_, int BKC, int BP>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  using WF = WgFrag<T>;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int YV = BN_ / VEC, XV = BKC / VEC;        // vectors per tile row
  constexpr int YS = BN_ + 64 / (int)sizeof(T);        // row stride: +64 B keeps the 4 rows of a tr-read on disjoint banks
  constexpr int XS = BKC + 64 / (int)sizeof(T);
  constexpr int Y_RPP = 256 / YV, X_RPP = 256 / XV;
  constexpr int Y_IT = BP / Y_RPP, X_IT = BP / X_RPP;
  constexpr int WTN = BN_ / 2, WTK = BKC / 2;          // 2x2 waves
  constexpr int TN = WTN / 32, TK = WTK / 32;
  static_assert(BP % Y_RPP == 0 && BP % X_RPP == 0, "mapping");
  __shared__ __attribute__((aligned(16))) T smem[2 * BP * (YS + XS)];
  T* Ysm = smem;                 // [2][BP][YS]
  T* Xsm = smem + 2 * BP * YS;   // [2][BP][XS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int per_z = a.tiles_k * a.tiles_n;
  int zz, t;
  if (!wgrad_work_item(a, per_z, zz, t)) return;
  const int k0 = (t % a.tiles_k) * BKC, n0 = (t / a.tiles_k) * BN_;
  const int m_begin = zz * a.m_per_split;
  const int m_end = min(a.M, m_begin + a.m_per_split);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);

  // X-tile column owned by this thread: fixed (tap, channel) for the whole reduction
  const int xv = tid % XV, xrow = tid / XV;
  const int kcol = k0 + xv * VEC;
  const bool k_ok = kcol < a.Ktot;
  int tap = k_ok ? kcol / a.Cs : 0;
  const int xc = k_ok ? kcol - tap * a.Cs : 0;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const int dh = tr * a.dil - a.pad, dw_ = ts * a.dil - a.pad;
  const int yv = tid % YV, yrow = tid / YV;
  const int ncol = n0 + yv * VEC;
  const bool n_ok = ncol < a.Cd;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
  // branch-free loads of the pixel block starting at mb (rows >= m_end, padding taps and tail columns read as zeros)
  auto issue = [&](uint4 (&rx)[X_IT], uint4 (&ry)[Y_IT], int mb) {
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int m = mb + xrow + i * X_RPP;
      const uint32_t mm = (uint32_t)min(m, a.M - 1);
      const uint32_t n_img = fdiv(mm, a.fd_hw);
      const uint32_t rem = mm - n_img * a.fd_hw.d;
      const uint32_t hd = fdiv(rem, a.fd_w);
      const uint32_t wd = rem - hd * a.fd_w.d;
      const int hs = (int)hd * a.stride + dh, ws = (int)wd * a.stride + dw_;
      const bool ok = k_ok && m < m_end && (unsigned)hs < (unsigned)a.Hs && (unsigned)ws < (unsigned)a.Ws;
      const unsigned off = (unsigned)(((int)n_img * a.Hs * a.Ws + hs * a.Ws + ws) * a.ldx + xc) * (unsigned)sizeof(T);
      rx[i] = bload16(rs_x, ok ? off : OOB);
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int m = mb + yrow + i * Y_RPP;
      const unsigned off = (unsigned)(m * a.ldy + ncol) * (unsigned)sizeof(T);
      ry[i] = bload16(rs_y, (n_ok && m < m_end) ? off : OOB);
    }
  };
  auto store_tiles = [&](const uint4 (&rx)[X_IT], const uint4 (&ry)[Y_IT], int buf) {
    T* Yb = Ysm + buf * BP * YS;
    T* Xb = Xsm + buf * BP * XS;
#pragma unroll
    for (int i = 0; i < X_IT; ++i)
      *reinterpret_cast<uint4*>(Xb + (xrow + i * X_RPP) * XS + xv * VEC) = rx[i];
#pragma unroll
    for (int i = 0; i < Y_IT; ++i)
      *reinterpret_cast<uint4*>(Yb + (yrow + i * Y_RPP) * YS + yv * VEC) = ry[i];
  };

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int cur) {
    const T* Yb = Ysm + cur * BP * YS;
    const T* Xb = Xsm + cur * BP * XS;
#pragma unroll
    for (int ks = 0; ks < BP / WF::KS; ++ks) {
      typename WF::frag fy[TN], fx[TK];
#pragma unroll
      for (int i = 0; i < TN; ++i) fy[i] = WF::load(Yb, YS, ks * WF::KS, wn * WTN + i * 32, lane);
#pragma unroll
      for (int j = 0; j < TK; ++j) fx[j] = WF::load(Xb, XS, ks * WF::KS, wk * WTK + j * 32, lane);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = WF::mma(fy[i], fx[j], acc[i][j]);
    }
  };

  const int nit = (m_end - m_begin + BP - 1) / BP;
  if (nit <= 0) return;
  // two pixel blocks in flight (blocks past m_end are all-OOB loads = zeros)
  uint4 rx0[X_IT], ry0[Y_IT], rx1[X_IT], ry1[Y_IT];
  int mb = m_begin;
  issue(rx0, ry0, mb); mb += BP;
  issue(rx1, ry1, mb); mb += BP;
  store_tiles(rx0, ry0, 0);
  __syncthreads();
  for (int it = 0;;) {
    issue(rx0, ry0, mb); mb += BP;
    compute(0);
    store_tiles(rx1, ry1, 1);
    __syncthreads();
    if (++it >= nit) break;
    issue(rx1, ry1, mb); mb += BP;
    compute(1);
    store_tiles(rx0, ry0, 0);
    __syncthreads();
    if (++it >= nit) break;
  }
  // D[row -> n][col -> k]: one 128-byte fp32 segment per half-wave per accumulator register
  const int l31 = lane & 31, lh = lane >> 5;
  if (a.ws) {
    // partial tile of this pixel slice -> its own BN_ x BKC fp32 slab of the workspace with plain stores; wgrad_slab_reduce_gen_kernel adds
    // the slices in slice order (the weight gradient is bit-reproducible; the atomics below add in arrival order)
    float* slab = a.ws + ((size_t)t * a.splits + zz) * (BN_ * BKC);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int kl = wk * WTK + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nl = wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (a.nt & 1) __builtin_nontemporal_store(acc[i][j][r], slab + nl * BKC + kl);
          else slab[nl * BKC + kl] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int k = k0 + wk * WTK + j * 32 + l31;
      if (k >= a.Ktot) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < a.Cd) atomicAdd(a.dw + (size_t)n * a.Ktot + k, acc[i][j][r]);
      }
    }
}

// --------------------------------------------------------------------------
// bf16 weight gradient for Cout >= 256: 256 (cout) x 256 (k columns) tile, 32 pixels per step, 8 waves (2 x 4, 128 x 64
// each), FOUR 32 KiB LDS stages filled by LDS-DMA (three steps in flight) - half the global->LDS bytes per FLOP of the
// 128x128 kernel above.  Tiles stay [pixel][channel] as loaded; MFMA fragments come from ds_read_b64_tr_b16.
//  * LDS rows are 256 channels = 512 B, unpadded (a DMA wave-instruction writes 2 rows).  A transposing read touches 4
//    consecutive rows x 64 B per half-wave, so the 64-byte block b of row r is stored at block b ^ (r & 3): the 4 rows
//    then sit on 4 disjoint bank groups.  The swizzle is applied to the per-lane SOURCE column and on the fragment reads.
//  * thread t owns 16-byte position t & 31 of rows (t >> 5) + 16 i: (r & 3) is the same for all of them, so its source
//    column - and for the X tile its (tap, channel) - is fixed for the whole reduction.
// --------------------------------------------------------------------------
// LDS-DMA as inline asm (M0 = wave-uniform LDS base).  Used by the weight-gradient kernel: with the builtin the waitcnt
// pass knows LDS is being written and puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 (it cannot tell that
// the stage being read is not the one in flight), which would serialise the three-steps-in-flight pipeline.  Ordering is
// this kernel's own job: counted s_waitcnt vmcnt + s_barrier before a stage is read, as in the forward kernels.
__device__ __forceinline__ void dma16_asm(u32x4 rsrc, void* lds_wave_base, unsigned off) {
  const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_off), "v"(off), "s"(rsrc) : "memory");
}
// (the LDS byte address as a wave-uniform integer: no generic -> LDS pointer conversion, with its null check, per instruction)
__device__ __forceinline__ void dma16_lds(u32x4 rsrc, unsigned lds_off, unsigned off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_off), "v"(off), "s"(rsrc) : "memory");
}
__device__ __forceinline__ u32x4 raw_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  u32x4 r = {(unsigned)b, (unsigned)(b >> 32), bytes, 0x00020000u};
  return r;
}

__device__ __forceinline__ bf16x8 wg_frag_sw(const unsigned char* tile, int kk0, int col0, int lane) {
  const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
  const int R = kk0 + 8 * (g >> 1) + q, C = col0 + 16 * (g & 1) + 4 * p;
  const unsigned char* ad = tile + R * 512 + ((((C >> 5) ^ q) << 6) | ((2 * C) & 63));
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  union { struct { s16x4 a, b; } s; bf16x8 f; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ad));
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(ad + 4 * 512));
  return u.f;
}

// --------------------------------------------------------------------------
// conv_wgrad_p8_kernel<false> (r03; the default form): the 256x256 tile, fragments, walk and epilogue described above on the phase structure of conv_p8.hip
// (its one-barrier-per-step predecessors conv_wgrad_dma256_kernel<STAG> live in scripts/proto/conv_wgrad_dma256.hip as the A/B reference).
// What the yardstick GEMM and conv_igemm_p8_kernel taught (profiles/r03_p8_phase_stamps.txt): a load segment must hold nothing but
// the fragment reads and the LDS-DMA issue - every VALU / SALU instruction and branch in it delays the barrier its SIMD partner's MFMA
// segment ends on - and the two waves of a SIMD alternate cleanly only when both segments are short.  So a 32-pixel step becomes two
// phases of one 16-pixel half each:
//     [12 ds_read_b64_tr_b16 of (stage, half h) | Y piece h + X piece h of step t+3 | s_waitcnt vmcnt(10)]  s_barrier
//     [lgkmcnt(0) | 8 MFMAs 32x32x16, the walk of pixel row h and the next phase's offsets spread between them]   s_barrier
// with the second cout half (= the other wave of every SIMD) one barrier behind.  A thread's two LDS-DMA rows are pixel rows
// prow and prow + 16 of a stage, i.e. piece i IS half i, so a half is restaged two phases after its last read (the template's rule),
// and `vmcnt(10)` (five phases' pieces stay in flight) retires the half that the NEXT phase reads.  Same MFMA order per accumulator
// as conv_wgrad_dma256_kernel: bit-identical results.
// --------------------------------------------------------------------------
// conv_wgrad_p8_kernel<true> (r06; CSS_WGRAD_MFMA=16): the same 256 x 256 tile, four-stage LDS-DMA ring, row walk and two-phase step as the 32x32x16 form, on
// v_mfma_f32_16x16x32_bf16 - the shape the chip holds a higher clock on under load (MI355X_MICROARCH.md DVFS give-back item 7; the forward
// kernel made this move in round 3).  An MFMA now reduces over ALL 32 pixels of a step, so the two phases of a step split the wave's
// 128 (cout) x 64 (k column) tile by cout instead of by pixel:
//     phase A [12 transposed reads: X fragments 2, 3 + dY fragments 0-3 | pieces of half 0 of step t+4 | vmcnt(10): ALL of stage t+1 landed]        s_barrier
//             [lgkmcnt(0) | 16 MFMAs 16x16x32, the walk of pixel row 1 between them]                                                              s_barrier
//     phase B [12 transposed reads: dY fragments 4-7 + X fragments 0, 1 of step t+1 | pieces of half 1 of step t+4 | lgkmcnt(0)]                  s_barrier
//             [16 MFMAs, the walk of pixel row 0 and the next stage's read addresses between them]                                                s_barrier
// (five stages: the one read ahead by phase B must have landed one phase earlier than in the 32x32x16 form)
// Phase B reads all 32 pixel rows of the stage, and the very next phase (phase A of the next step; for the other cout half of the workgroup,
// which runs one barrier behind, the same barrier slot) restages its half 0: phase B therefore completes its reads BEFORE its barrier.
// Operand roles: A = X^T (rows = k columns), B = dY (columns = cout), so D has cout on the lane and 4 consecutive k columns in the 4
// registers of an accumulator - the slab rows are written with 16-byte stores.  MFMA lane group g (lane >> 4) takes pixel rows 8 g .. 8 g + 7
// of the step: a 32-lane half reads rows 8 n + q and 8 n + 8 + q of the SAME 16 channels, so on top of the 64-byte block swizzle by (row & 3)
// the two 16-byte chunks pairs of a block swap with bit 3 of the row (chunk ^= ((row >> 3) & 1) << 1): 4 rows x 2 half-blocks = 64 banks once.
// Pixel rows reach an accumulator in the same order as in the 32x32x16 form (rows 0-7, 8-15, 16-23, 24-31 of a step): on the harness data the two
// forms agree BIT FOR BIT on every launch shape of the step (profiles/r06_wgrad_mfma_shape_check.txt).  Measured (profiles/r06_wgrad_mfma_shape_ab.txt):
// the in-kernel clock rises by 30 % (1.43 -> 1.87 GHz on the layer-4 3x3), the cycles of the K loop by 46 %, wall time +7 %: the 32x32x16 form stays the default.
#ifdef WG_STAMP
// Diagnostic build only (scripts/wgrad_bench.hip -DWG_STAMP; in the library no stamp executes): s_memtime / s_memrealtime around the K loop of
// every workgroup -> [workgroup][4] in a buffer of their own; the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md,
// DVFS give-back item 6).
__device__ unsigned long long* wg_stamp_buf = nullptr;
#define WG_STAMP_BEGIN()                                                                                                        \
  unsigned long long st_c0, st_r0;                                                                                              \
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0)::"memory")
#define WG_STAMP_END()                                                                                                          \
  do {                                                                                                                          \
    unsigned long long st_c1, st_r1;                                                                                            \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c1), "=s"(st_r1)::"memory");              \
    if (threadIdx.x == 0 && wg_stamp_buf) {                                                                                     \
      unsigned long long* o_ = wg_stamp_buf + (size_t)blockIdx.x * 4;                                                           \
      o_[0] = st_c0; o_[1] = st_r0; o_[2] = st_c1; o_[3] = st_r1;                                                               \
    }                                                                                                                           \
  } while (0)
#else
#define WG_STAMP_BEGIN()
#define WG_STAMP_END()
#endif
template <bool MF16>
__global__ __launch_bounds__(512) void conv_wgrad_p8_kernel(const WgradArgs a) {
  constexpr int BKC = 256, BP = 32, NST = MF16 ? 5 : 4;         // MF16: five 32 KiB stages (160 KiB: all of the CU's LDS), four steps in flight
  constexpr int T_BYTES = BP * 512, ST_BYTES = 2 * T_BYTES;     // Y tile then X tile
  constexpr int WTN = 128, WTK = 64;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int per_z = a.tiles_k * a.tiles_n;
  int zz, t;
  if (!wgrad_work_item(a, per_z, zz, t)) return;
  const int k0 = (t % a.tiles_k) * BKC, n0 = (t / a.tiles_k) * 256;
  // live-row compaction (WgradArgs): the tile lies inside ONE tap (Cs % 256 == 0), so its kernel row is workgroup-uniform; the pixel index of
  // this kernel runs over the live output rows of every image: j -> image j / L, row row_lo + (j % L) / Wd, L = live rows x Wd
  const int trow = a.compact ? (k0 / a.Cs) / a.S : 0;
  const int hlo = a.compact ? a.row_lo[trow] : 0, nrows = a.compact ? a.row_n[trow] : a.Hd;
  const FastDiv fdL = a.compact ? a.fd_L[trow] : a.fd_hw;
  const int L = nrows * a.Wd, mps = a.compact ? a.row_mps[trow] : a.m_per_split;
  const int m_begin = zz * mps;
  const int m_end = min(a.N * L, m_begin + mps);
  const int nit = (m_end - m_begin + BP - 1) / BP;
  if (nit <= 0) {                                              // (an empty slice of a compacted tile: its slab is still summed by the reduction)
    if (a.ws) {
      float* slab = a.ws + ((size_t)t * a.splits + zz) * (256 * 256);
      for (int e = tid; e < 256 * 256 / 4; e += 512) reinterpret_cast<f32x4*>(slab)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }

  const int prow = tid >> 5;                                    // my two LDS-DMA rows: pixel rows prow and prow + 16 of a stage
  // the source chunk that lands in LDS position tid & 31 of my rows: 64-byte blocks swizzled by (row & 3); MF16: + the half-block swap by bit 3 of the row
  const int schunk = (tid & 31) ^ ((prow & 3) << 2) ^ (MF16 ? ((prow >> 3) & 1) << 1 : 0);
  const int kcol = k0 + schunk * 8;
  const bool k_ok = kcol < a.Ktot;
  const int tap = k_ok ? kcol / a.Cs : 0;
  const int xc = k_ok ? kcol - tap * a.Cs : 0;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const int dh = tr * a.dil - a.pad, dw_ = ts * a.dil - a.pad;
  const int ncol = n0 + schunk * 8;
  const bool n_ok = ncol < a.Cd;

  const u32x4 rs_x = raw_rsrc(a.x, a.x_bytes), rs_y = raw_rsrc(a.dy, a.dy_bytes);
  const int q_w = (int)fdiv((uint32_t)BP, a.fd_w), d_w = BP - q_w * a.Wd;
  const int d_n = (int)fdiv((uint32_t)BP, fdL), d_h = q_w - d_n * nrows;
  const int xrow = a.ldx * 2;
  const int sx_w = a.stride * xrow, sx_h = a.stride * a.Ws * xrow, sx_n = a.Hs * a.Ws * xrow;
  const int D0 = d_n * sx_n + d_h * sx_h + d_w * sx_w;
  const int Dw = sx_h - a.Wd * sx_w, Dh = sx_n - nrows * sx_h;
  const int Yh = (a.Hd * a.Wd - L) * a.ldy * 2;                  // dY: the dead rows between two images' live rows
  const int ystep = BP * a.ldy * 2 + d_n * Yh;
  int r_m[2], r_hs[2], r_ws[2];
  unsigned r_xo[2], r_yo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_begin + prow + i * 16;
    const uint32_t n_img = fdiv((uint32_t)m, fdL);
    const uint32_t rem = (uint32_t)m - n_img * (uint32_t)L;
    const uint32_t hq = fdiv(rem, a.fd_w);
    const uint32_t wd = rem - hq * a.fd_w.d;
    const int hd = hlo + (int)hq;
    r_m[i] = m;
    r_hs[i] = hd * a.stride + dh;
    r_ws[i] = (int)wd * a.stride + dw_;
    r_xo[i] = (unsigned)(((int)n_img * a.Hs * a.Ws + r_hs[i] * a.Ws + r_ws[i]) * a.ldx + xc) * 2u;
    r_yo[i] = (unsigned)((((int)n_img * a.Hd + hd) * a.Wd + (int)wd) * a.ldy + ncol) * 2u;
  }
  const int hs_hi = (hlo + nrows - 1) * a.stride + dh, ws_hi = (a.Wd - 1) * a.stride + dw_;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void*)smem) + (unsigned)wave * 1024u;
  auto row_offsets = [&](int i, unsigned& vy, unsigned& vx) {
    vy = (n_ok && r_m[i] < m_end) ? r_yo[i] : OOB;
    const bool ok = k_ok && r_m[i] < m_end && (unsigned)r_hs[i] < (unsigned)a.Hs && (unsigned)r_ws[i] < (unsigned)a.Ws;
    vx = ok ? r_xo[i] : OOB;
    asm volatile("" : "+v"(vy), "+v"(vx));     // (pinned: the optimizer must not sink this into the load segment that uses it)
    r_m[i] += BP;
    int ws = r_ws[i] + d_w * a.stride, hs = r_hs[i] + d_h * a.stride;
    int dx = D0;
    const bool cw = ws > ws_hi;
    ws -= cw ? a.Wd * a.stride : 0;
    hs += cw ? a.stride : 0;
    dx += cw ? Dw : 0;
    const bool ch = hs > hs_hi;
    hs -= ch ? nrows * a.stride : 0;
    dx += ch ? Dh : 0;
    r_ws[i] = ws;
    r_hs[i] = hs;
    r_xo[i] += (unsigned)dx;
    r_yo[i] += (unsigned)(ystep + (ch ? Yh : 0));
  };
  auto stage_half = [&](int stage, int i, unsigned vy, unsigned vx) {
    const unsigned sy = lds0 + (unsigned)stage * ST_BYTES + (unsigned)i * 8192u;
    dma16_lds(rs_y, sy, vy);
    dma16_lds(rs_x, sy + T_BYTES, vx);
  };

  unsigned vy, vx;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      row_offsets(i, vy, vx);
      stage_half(st, i, vy, vx);
    }
  row_offsets(0, vy, vx);
  if constexpr (MF16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // stage 0 (both halves) has landed
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                    // data phase 0 (half 0 of stage 0) has landed
  __builtin_amdgcn_s_barrier();
  if (wn == 1) __builtin_amdgcn_s_barrier();            // the second cout half runs one barrier behind
  asm volatile("" ::: "memory");

  typedef __attribute__((address_space(3))) s16x4 wgp_lds_v4;
  auto frag = [&](unsigned ad) {
    union { struct { s16x4 a, b; } s; bf16x8 f; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgp_lds_v4*)(uintptr_t)ad);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgp_lds_v4*)(uintptr_t)(ad + 4 * 512));
    return u.f;
  };
  if constexpr (!MF16) {
    constexpr int TN = 4, TK = 2;                                 // 32-wide fragments
    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // LDS byte addresses of my fragment reads in the stage being multiplied (half 0; half 1 = 16 rows = + 8192), see wg_frag_sw; they move
    // on to the next stage inside phase 1's MFMA segment
    unsigned fad[TN + TK];
    {
      const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
      const unsigned base = (unsigned)(uintptr_t)(lds_void*)smem + (unsigned)(8 * (g >> 1) + q) * 512u;
#pragma unroll
      for (int k = 0; k < TN + TK; ++k) {
        const int C = (k < TN ? wn * WTN + k * 32 : wk * WTK + (k - TN) * 32) + 16 * (g & 1) + 4 * pp;
        fad[k] = base + (k < TN ? 0u : (unsigned)T_BYTES) + (unsigned)((((C >> 5) ^ q) << 6) | ((2 * C) & 63));
      }
    }
    WG_STAMP_BEGIN();
    int st_c = 0, st_i = 3;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 fy[TN], fx[TK];
        // ---- load segment: fragments of half h, pieces of half h of step it+3, one counted wait ----
#pragma unroll
        for (int i = 0; i < TN; ++i) fy[i] = frag(fad[i] + h * 8192);
#pragma unroll
        for (int j = 0; j < TK; ++j) fx[j] = frag(fad[TN + j] + h * 8192);
        __builtin_amdgcn_sched_barrier(0);
        stage_half(st_i, h, vy, vx);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // the half the NEXT phase reads has landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment: 8 MFMAs + the walk of the row the next phase issues (+ the next stage's read addresses) ----
        row_offsets(h ^ 1, vy, vx);
        if (h == 1) {
          const unsigned d = st_c == 3 ? (unsigned)(-3 * ST_BYTES) : (unsigned)ST_BYTES;
#pragma unroll
          for (int k = 0; k < TN + TK; ++k) {
            fad[k] += d;
            asm volatile("" : "+v"(fad[k]));
          }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int g_ = 0; g_ < 8; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      st_c = (st_c + 1) & 3;
      st_i = (st_i + 1) & 3;
    }
    WG_STAMP_END();
    if (wn == 0) __builtin_amdgcn_s_barrier();            // the barrier the other half ran at the start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ghost DMAs must have landed before the workgroup's LDS is released
    const int l31 = lane & 31, lh = lane >> 5;
    if (a.ws) {
      float* slab = a.ws + ((size_t)t * a.splits + zz) * (256 * 256);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const int kl = wk * WTK + j * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nl = wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (a.nt & 1) __builtin_nontemporal_store(acc[i][j][r], slab + nl * 256 + kl);
            else slab[nl * 256 + kl] = acc[i][j][r];
          }
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int k = k0 + wk * WTK + j * 32 + l31;
        if (k >= a.Ktot) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < a.Cd) atomicAdd(a.dw + (size_t)n * a.Ktot + k, acc[i][j][r]);
        }
      }
  } else {
    constexpr int TN = 8, TK = 4;                                 // 16-wide fragments
    f32x4 acc[TK][TN];
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS byte addresses of my fragment reads in the stage being multiplied: lane (i = lane & 15, g = lane >> 4; q = i >> 2, p = i & 3) reads
    // 8 bytes of row 8 g + q (second read: + 4 rows), channels c0 + 4 p .. + 3 of a 16-channel block
    unsigned fad[TN + TK];
    {
      const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
      const unsigned base = (unsigned)(uintptr_t)(lds_void*)smem + (unsigned)(8 * g + q) * 512u + 8u * (pp & 1);
#pragma unroll
      for (int k = 0; k < TN + TK; ++k) {
        const int c0 = k < TN ? wn * WTN + k * 16 : wk * WTK + (k - TN) * 16;
        const int ch = ((c0 >> 3) + (pp >> 1)) ^ (q << 2) ^ ((g & 1) << 1);
        fad[k] = base + (k < TN ? 0u : (unsigned)T_BYTES) + (unsigned)ch * 16u;
      }
    }
    // The k-column fragments 0 and 1 of a step are read one phase EARLY (phase B of the step before, from the next stage), so that both load
    // segments issue 12 transposed reads (the first version read 16 + 8: the 16-read segment alone takes the LDS array ~256 cycles for the
    // four waves of a CU that load at the same time - as long as the partner's whole MFMA segment; profiles/r06_wgrad_mfma_shape_ab.txt).
    bf16x8 fxa[2], fxb[2];
    fxa[0] = frag(fad[TN + 0]);
    fxa[1] = frag(fad[TN + 1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WG_STAMP_BEGIN();
    int st_c = 0, st_i = NST - 1;
    // one step: `cur` = this step's k-column fragments 0, 1 (already in registers), `nxt` = where the next step's go
    auto step = [&](bf16x8 (&cur)[2], bf16x8 (&nxt)[2]) {
      bf16x8 fx23[2], fy[4];
      // ================= phase A: k-column fragments 2, 3 + cout fragments 0-3 (12 reads) =================
      fx23[0] = frag(fad[TN + 2]);
      fx23[1] = frag(fad[TN + 3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fy[i] = frag(fad[i]);
      __builtin_amdgcn_sched_barrier(0);
      stage_half(st_i, 0, vy, vx);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // every piece of the NEXT step's stage has landed (phase B reads its k-column tile): 2.5 steps stay in flight
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      row_offsets(1, vy, vx);
      // (the read addresses this phase is done with move on to the next stage here, the other half in phase B: the vector issue port has room for
      // two VALU instructions per 16x16x32 MFMA - 32 per segment - and the walk of a pixel row takes ~22 of them)
      const unsigned dn = st_c == NST - 1 ? (unsigned)(-(NST - 1) * ST_BYTES) : (unsigned)ST_BYTES;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fad[k] += dn;
        asm volatile("" : "+v"(fad[k]));
      }
#pragma unroll
      for (int k = TN + 2; k < TN + 4; ++k) {
        fad[k] += dn;
        asm volatile("" : "+v"(fad[k]));
      }
#pragma unroll
      for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? cur[j] : fx23[j - 2], fy[i], acc[j][i], 0, 0, 0);
#pragma unroll
      for (int g_ = 0; g_ < 16; ++g_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ================= phase B: cout fragments 4-7 + the NEXT step's k-column fragments 0, 1 (12 reads; this stage's reads complete before
      // the barrier: the next phase restages its half 0) =================
#pragma unroll
      for (int i = 0; i < 4; ++i) fy[i] = frag(fad[4 + i]);
      nxt[0] = frag(fad[TN + 0] + dn);
      nxt[1] = frag(fad[TN + 1] + dn);
      __builtin_amdgcn_sched_barrier(0);
      stage_half(st_i, 1, vy, vx);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      row_offsets(0, vy, vx);
#pragma unroll
      for (int k = 4; k < TN + 2; ++k) {
        fad[k] += dn;
        asm volatile("" : "+v"(fad[k]));
      }
#pragma unroll
      for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? cur[j] : fx23[j - 2], fy[i], acc[j][4 + i], 0, 0, 0);
#pragma unroll
      for (int g_ = 0; g_ < 16; ++g_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      st_c = st_c == NST - 1 ? 0 : st_c + 1;
      st_i = st_i == NST - 1 ? 0 : st_i + 1;
    };
    int it = 0;
    for (; it + 1 < nit; it += 2) {          // (pairs: the fragments carried from step to step alternate between two register sets - no moves)
      step(fxa, fxb);
      step(fxb, fxa);
    }
    if (it < nit) step(fxa, fxb);
    WG_STAMP_END();
    if (wn == 0) __builtin_amdgcn_s_barrier();            // the barrier the other half ran at the start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ghost DMAs must have landed before the workgroup's LDS is released
    // D: lane & 15 = cout within the fragment, 4 (lane >> 4) + r = k column within the fragment
    const int l15 = lane & 15, lg = lane >> 4;
    if (a.ws) {
      float* slab = a.ws + ((size_t)t * a.splits + zz) * (256 * 256);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int nl = wn * WTN + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const int kl = wk * WTK + j * 16 + 4 * lg;
          if (a.nt & 1) __builtin_nontemporal_store(acc[j][i], reinterpret_cast<f32x4*>(slab + nl * 256 + kl));
          else *reinterpret_cast<f32x4*>(slab + nl * 256 + kl) = acc[j][i];
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + l15;
      if (n >= a.Cd) continue;
#pragma unroll
      for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = k0 + wk * WTK + j * 16 + 4 * lg + r;
          if (k < a.Ktot) atomicAdd(a.dw + (size_t)n * a.Ktot + k, acc[j][i][r]);
        }
    }
  }
}

// dw[n][k] += sum over the pixel slices of ws[tile][slice][n - n0][k - k0] (fixed order: the weight gradient is bit-reproducible).
// grid = (64, tiles): block (bx, t) owns rows 4 bx .. 4 bx + 3 of tile t; thread = 4 consecutive k columns.
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;
__device__ __forceinline__ float4 wg_ld4(const float4* p, bool nt) {      // (CSS_WGRAD_NT bit 1: the slabs are read exactly once)
  if (!nt) return *p;
  const wg_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const wg_f32x4*>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
}
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int tiles_k,
                                                                int Cd, int Ktot, int nt) {
  const int t = blockIdx.y;
  const int k0 = (t % tiles_k) * 256, n0 = (t / tiles_k) * 256;
  const int nl = blockIdx.x * 4 + (threadIdx.x >> 6), kl = (threadIdx.x & 63) * 4;
  const int n = n0 + nl, k = k0 + kl;
  if (n >= Cd || k >= Ktot) return;
  const float4* p = reinterpret_cast<const float4*>(ws + (size_t)t * splits * (256 * 256) + nl * 256 + kl);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  int z = 0;
  for (; z + 3 < splits; z += 4) {        // four slabs in flight
    const float4 v0 = wg_ld4(p + (size_t)(z + 0) * (256 * 256 / 4), nt), v1 = wg_ld4(p + (size_t)(z + 1) * (256 * 256 / 4), nt);
    const float4 v2 = wg_ld4(p + (size_t)(z + 2) * (256 * 256 / 4), nt), v3 = wg_ld4(p + (size_t)(z + 3) * (256 * 256 / 4), nt);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
    s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
  }
  for (; z < splits; ++z) {
    const float4 v0 = wg_ld4(p + (size_t)z * (256 * 256 / 4), nt);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
  }
  const float r[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  float* o = dw + (size_t)n * Ktot + k;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k + e < Ktot) o[e] += r[e];
}

// The same for the BN x BKC tiles of conv_wgrad_kernel (128 x 128 bf16, 64 x 64 fp32): grid = (BN / 4, tiles), thread = 4 consecutive
// k columns of one row (BKC / 4 threads per row, 1024 / BKC rows per block), slices in order.
template <int BN_, int BKC>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_gen_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int tiles_k,
                                                                    int Cd, int Ktot, int nt) {
  constexpr int TPR = BKC / 4, RPB = 256 / TPR;            // threads per tile row, rows per block
  const int t = blockIdx.y;
  const int k0 = (t % tiles_k) * BKC, n0 = (t / tiles_k) * BN_;
  const int nl = blockIdx.x * RPB + threadIdx.x / TPR, kl = (threadIdx.x % TPR) * 4;
  const int n = n0 + nl, k = k0 + kl;
  if (nl >= BN_ || n >= Cd || k >= Ktot) return;
  const float4* p = reinterpret_cast<const float4*>(ws + (size_t)t * splits * (BN_ * BKC) + nl * BKC + kl);
  constexpr size_t SL = (size_t)BN_ * BKC / 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  int z = 0;
  for (; z + 3 < splits; z += 4) {
    const float4 v0 = wg_ld4(p + (size_t)(z + 0) * SL, nt), v1 = wg_ld4(p + (size_t)(z + 1) * SL, nt), v2 = wg_ld4(p + (size_t)(z + 2) * SL, nt),
                 v3 = wg_ld4(p + (size_t)(z + 3) * SL, nt);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
    s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
  }
  for (; z < splits; ++z) {
    const float4 v0 = wg_ld4(p + (size_t)z * SL, nt);
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
  }
  const float r[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  float* o = dw + (size_t)n * Ktot + k;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k + e < Ktot) o[e] += r[e];
}


// Pixel slices of the weight gradient.  Every slice stores one partial tile (a slab) per weight tile and a second kernel adds the slabs in
// slice order, so the number of slices s trades three things: whole rounds of the chip (tiles x s workgroups on n_cu x slots places), the
// fixed cost per workgroup (pipeline fill, 256 KiB slab store) against the length of its pixel loop, and the slab traffic (tiles x s slabs
// written and read back - at the round-3 choice "whole rounds of every XCD" 377 MB per layer-4 3x3 launch, a fifth of its time).  The
// estimate below (microseconds; the constants are the measured orders of magnitude, only their ratios matter) is minimised over s:
//     rounds(s) x (steps(s) x t_step + t_fix)  +  tiles x s x t_slab,      steps = ceil(M / s / bp)
// with at least four steps per slice.  (Any s is exact: the slabs of a tile are added in slice order whatever their number.)
// Round 5: the Cout <= 64 layers (layer 1, the stems, `project`) on 64 x 64 weight tiles instead of 128 x 128 (of which they fill a half or a quarter):
// twice / four times the useful share of every fragment read and MFMA, three workgroups per CU instead of two.  OPT-IN (CSS_WGRAD_N64=1): measured
// -0.06 ms (c2) / -0.3 ms (c4) of weight-gradient time per step (profiles/r05_wgrad_n64_ab.txt) - inside the noise at the headline workload: these
// layers are bound by streaming their two operands, not by the tile - and another summation order of those gradients (same bits run to run).
static bool wgrad_n64(int dtype, int Cd, bool big) {
  static const bool on = getenv("CSS_WGRAD_N64") && atoi(getenv("CSS_WGRAD_N64")) != 0;
  return on && dtype == CSS_BF16 && !big && Cd <= 64;
}
void css_wgrad_plan_(int M, int Ktot, int Cd, int dtype, int n_cu, int* splits_out, int* mps_out) {
  static const bool no_256 = getenv("CSS_NO_DMA256_WGRAD") != nullptr;
  const bool big = dtype == CSS_BF16 && Cd >= 256 && Ktot >= 256 && !no_256;
  const bool n64 = wgrad_n64(dtype, Cd, big);            // bf16, Cout <= 64: 64 x 64 tiles (the 128-wide tile is half empty there)
  const int bn = dtype == CSS_BF16 ? (big ? 256 : (n64 ? 64 : 128)) : 64, bkc = bn, bp = dtype == CSS_BF16 ? (big ? 32 : 64) : 16;
  const int tiles = cdiv(Ktot, bkc) * cdiv(Cd, bn);
  const int places = n_cu * (big ? 1 : (n64 ? 3 : 2));
  const double t_step = big ? 0.86 : (dtype == CSS_BF16 ? 1.1 : 1.0), t_fix = big ? 5.0 : 3.0;
  const double t_slab = (double)bn * bkc * 4 * 2 / 4.0e6;          // a slab written and read once at ~4 TB/s
  int best_s = 1;
  double best = 1e300;
  for (int s = 1; s <= 1024; ++s) {
    const int mps = cdiv(cdiv(M, s), bp) * bp;
    if (s > 1 && mps < 4 * bp) break;
    if (cdiv(M, mps) != s) continue;                               // (rounding the slice length up made a slice empty: same as a smaller s)
    const double rounds = (double)cdiv((long)tiles * s, places);
    const double cost = rounds * ((double)(mps / bp) * t_step + t_fix) + (double)tiles * s * t_slab;
    if (cost < best - 1e-9) { best = cost; best_s = s; }
  }
  const int mps = cdiv(cdiv(M, best_s), bp) * bp;
  *splits_out = cdiv(M, mps);
  *mps_out = mps;
}

// bytes of workspace that let css_launch_wgrad replace its fp32 atomics by plain partial-tile stores + an ordered reduction (0: the
// shape takes a kernel that has no such path)
size_t css_wgrad_ws_bytes_(int M, int Ktot, int Cd, int dtype, int n_cu) {
  static const bool no_256 = getenv("CSS_NO_DMA256_WGRAD") != nullptr, no_ws = getenv("CSS_WGRAD_ATOMICS") != nullptr;
  if (no_ws || M <= 0 || (dtype != CSS_BF16 && dtype != CSS_F32)) return 0;
  int splits, mps;
  css_wgrad_plan_(M, Ktot, Cd, dtype, n_cu, &splits, &mps);
  const bool big = dtype == CSS_BF16 && Cd >= 256 && Ktot >= 256 && !no_256;
  const int bn = big ? 256 : (dtype == CSS_BF16 && !wgrad_n64(dtype, Cd, big) ? 128 : 64);        // (square tiles: css_launch_wgrad)
  return (size_t)cdiv(Ktot, bn) * cdiv(Cd, bn) * splits * ((size_t)bn * bn * sizeof(float));
}

int css_wgrad_mfma_override_ = 0;     // harnesses: 16 / 32 forces the MFMA shape of the 256 x 256 kernel (0: CSS_WGRAD_MFMA, default 32)
int css_wgrad_no_compact_override_ = 0;   // harnesses: 1 switches the live-row compaction off (as CSS_WGRAD_NO_COMPACT=1)
int css_launch_wgrad(WgradArgs a, int dtype, int n_cu, hipStream_t st, LaunchProf* prof) {
  if (a.M <= 0) return CSS_OK;
  a.fd_hw = make_fastdiv((uint32_t)(a.Hd * a.Wd));
  a.fd_w = make_fastdiv((uint32_t)a.Wd);
  int bn, bkc, bp;
  static const bool no_256 = getenv("CSS_NO_DMA256_WGRAD") != nullptr;
  const bool big = dtype == CSS_BF16 && a.Cd >= 256 && a.Ktot >= 256 && !no_256;   // 256x256 LDS-DMA kernel, one workgroup per CU
  const bool n64 = wgrad_n64(dtype, a.Cd, big);
  if (dtype == CSS_BF16) {
    bn = big ? 256 : (n64 ? 64 : 128); bkc = bn; bp = big ? 32 : 64;
    if (a.Cs % 8 || a.ldx % 8 || a.ldy % 8 || a.Cd % 8) return CSS_ERR_ARG;
  } else if (dtype == CSS_F32) {
    bn = 64; bkc = 64; bp = 16;
    if (a.Cs % 4 || a.ldx % 4 || a.ldy % 4 || a.Cd % 4) return CSS_ERR_ARG;
  } else {
    return CSS_ERR_DTYPE;
  }
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.dy) & 15)) return CSS_ERR_ARG;
  {
    const size_t esz = dtype == CSS_BF16 ? 2 : 4;
    const size_t xb = (size_t)a.N * a.Hs * a.Ws * a.ldx * esz, yb = (size_t)a.M * a.ldy * esz;
    if (xb >= 0x7FFFFFF0ull || yb >= 0x7FFFFFF0ull) return CSS_ERR_ARG;   // 32-bit buffer offsets
    a.x_bytes = (unsigned)xb;
    a.dy_bytes = (unsigned)yb;
  }
  int splits, mps;
  css_wgrad_plan_(a.M, a.Ktot, a.Cd, dtype, n_cu, &splits, &mps);
  a.m_per_split = mps;
  a.splits = splits;
  a.tiles_k = cdiv(a.Ktot, bkc);
  a.tiles_n = cdiv(a.Cd, bn);
  // CSS_WGRAD_NT: bit 0 = non-temporal slab stores (no gain), bit 1 = non-temporal slab loads in the reduction (read exactly once: -0.45 ms per step,
  // most of it in the batch-norm backward reduction that follows - profiles/r06_nontemporal_ab.txt)
  static const int wg_nt = getenv("CSS_WGRAD_NT") ? atoi(getenv("CSS_WGRAD_NT")) & 3 : 2;
  a.nt = wg_nt & 1;
  const int rnt = (wg_nt >> 1) & 1;
  a.compact = 0;
  {
    static const bool no_compact = getenv("CSS_WGRAD_NO_COMPACT") && atoi(getenv("CSS_WGRAD_NO_COMPACT")) != 0;
    if (big && !no_compact && !css_wgrad_no_compact_override_ && a.R > 1 && a.R <= 3 && a.stride == 1 && a.Cs % 256 == 0) {
      bool any = false;
      for (int r = 0; r < a.R; ++r) {
        const int dh = r * a.dil - a.pad;
        int lo = dh < 0 ? -dh : 0, hi = a.Hd - 1 < a.Hs - 1 - dh ? a.Hd - 1 : a.Hs - 1 - dh;
        if (hi < lo) { lo = 0; hi = a.Hd - 1; }                    // (kernel row entirely in the padding: computed as zeros over all rows)
        a.row_lo[r] = lo;
        a.row_n[r] = hi - lo + 1;
        a.row_mps[r] = cdiv(cdiv(a.N * a.row_n[r] * a.Wd, splits), bp) * bp;
        a.fd_L[r] = make_fastdiv((uint32_t)(a.row_n[r] * a.Wd));
        any = any || a.row_n[r] < a.Hd;
      }
      a.compact = any ? 1 : 0;
      static const bool no_lpt = getenv("CSS_WGRAD_NO_LONGEST_FIRST") && atoi(getenv("CSS_WGRAD_NO_LONGEST_FIRST")) != 0;
      if (any && !no_lpt && a.tiles_k % a.R == 0) {              // two classes: rows with every output row live / the others
        a.cls_nrows[0] = a.cls_nrows[1] = 0;
        for (int r = 0; r < a.R; ++r) {
          const int c = a.row_n[r] == a.Hd ? 0 : 1;
          a.cls_rows[c][a.cls_nrows[c]++] = r;
        }
        if (a.cls_nrows[0] > 0 && a.cls_nrows[1] > 0) a.compact = 2;
      }
    }
  }
  dim3 g(cdiv((long)a.tiles_k * a.tiles_n * splits, 8) * 8);      // (wgrad_work_item: XCD x takes work items [x W8, (x + 1) W8))
  if (a.compact == 2) {
    const int tpr = a.tiles_k / a.R;
    g = dim3(8 * (cdiv(a.tiles_n * a.cls_nrows[0] * tpr * splits, 8) + cdiv(a.tiles_n * a.cls_nrows[1] * tpr * splits, 8)));
  }
  if (prof) prof->begin(big, 1.0, false);
  if ((size_t)a.tiles_k * a.tiles_n * a.splits * ((size_t)bn * bkc * sizeof(float)) > a.ws_bytes) a.ws = nullptr;   // atomics path (not reproducible)
  if (big) {
    // the MFMA shape of the 256 x 256 kernel: 32x32x16 (default: faster by wall, profiles/r06_wgrad_mfma_shape_ab.txt) or 16x16x32 (CSS_WGRAD_MFMA=16)
    static const bool env16 = getenv("CSS_WGRAD_MFMA") && atoi(getenv("CSS_WGRAD_MFMA")) == 16;
    const bool mf16 = css_wgrad_mfma_override_ ? css_wgrad_mfma_override_ == 16 : env16;
    if (mf16) hipLaunchKernelGGL(conv_wgrad_p8_kernel<true>, g, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(conv_wgrad_p8_kernel<false>, g, dim3(512), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(64, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits, a.tiles_k, a.Cd, a.Ktot, rnt);
  } else if (n64) {
    hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 64, 64, 64>), g, dim3(256), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL((wgrad_slab_reduce_gen_kernel<64, 64>), dim3(64 / 16, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits,
                         a.tiles_k, a.Cd, a.Ktot, rnt);
  } else if (dtype == CSS_BF16) {
    // (32 pixels per step - 40 KiB of LDS, four workgroups per CU instead of two - was measured and is slower: profiles/r04_small64_nst2_ab.txt)
    hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 128, 128, 64>), g, dim3(256), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL((wgrad_slab_reduce_gen_kernel<128, 128>), dim3(128 / 8, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits,
                         a.tiles_k, a.Cd, a.Ktot, rnt);
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<float, 64, 64, 16>), g, dim3(256), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL((wgrad_slab_reduce_gen_kernel<64, 64>), dim3(64 / 16, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits,
                         a.tiles_k, a.Cd, a.Ktot, rnt);
  }
  if (prof) prof->end();
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
