// conv_igemm_pp_kernel: the bf16 implicit-GEMM convolution (forward and data gradient) for Cout >= 256 on gfx950, second generation.
//
// Same arithmetic and data layout as conv_igemm_dma256_kernel (conv.hip: NHWC activations, [Cout][R][S][Cin] weights, 256 x 256
// output tile per workgroup, 8 waves as 2 (pixel halves) x 4 (channel quarters), 128 x 64 outputs per wave, four 32 KiB LDS stages
// of K = 32 filled by LDS-DMA, three stages in flight).  What changed, each from a counter of profiles/r01_*:
//  * PING-PONG: the two waves of a SIMD (wave w and w + 4 = the two pixel halves) run half a K step apart - one is in its LOAD
//    segment (12 ds_read_b128 fragment reads, 4 LDS-DMA issues, address update, counted waits) while the other is in its MFMA
//    segment (32 x v_mfma_f32_16x16x32_bf16 = 512 cycles, nothing else), two s_barrier per K step.  r01: 33 % of the wave time
//    parked on s_waitcnt / barriers because both waves of a SIMD read, waited and multiplied at the same time.
//  * K ORDER channel-slice outer, tap inner: the 9 taps of a dilated 3x3 re-read ONE 32-channel slice of the tile's neighbourhood
//    (<= 1 MB per XCD: L2) instead of sweeping all channels per tap (r01: FETCH_SIZE 3.3x the unique input, served by the Infinity
//    Cache at half the L2 rate).  Needs no new weight layout: the weight row of output channel n at (slice, tap) is 64 contiguous bytes.
//  * PERSISTENT: one workgroup per CU walks its tiles (XCD-contiguous, n fastest); the LDS-DMA stream runs on across tile
//    boundaries, so the next tile's first three K steps are in flight during the epilogue.
//  * EPILOGUE from registers: accumulator (16x16 tiles: a lane holds 4 consecutive channels of one pixel) -> bf16 ->
//    v_permlane16_swap pairs -> one 16-byte store per lane (64 contiguous bytes per pixel per instruction); no LDS staging,
//    no barrier, so it overlaps the other half's MFMA segment.  Batch-norm statistics: a wave owns a whole 128-row slab of its
//    64 channels, so the slab sums need no hand-over between waves (DPP row reduction, plain 16-byte stores).
//
// Reference shapes: generalframeworks/networks/resnet.py:119-139 (Bottleneck 1x1 / dilated 3x3), deeplabv3/aspp.py:17-72,
// deeplabv3/deeplabv3.py:115-133,151-169.
#include "common.h"
#include "launchers.h"
#include <cstdlib>
#include <type_traits>

namespace {
typedef __attribute__((address_space(3))) void pp_lds_void;
constexpr unsigned PP_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int pp_u32x4;
typedef __attribute__((ext_vector_type(4))) float pp_f32x4;

__device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
#ifdef PP_ABL_NOLOAD          // ablation (scripts/conv_bench.hip): no LDS-DMA at all, the MFMAs run on whatever the LDS holds
  asm volatile("" ::"v"(off));
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
#endif
}

// v += (same register rotated inside its 16-lane row): every lane of a row ends with the row's sum
__device__ __forceinline__ float pp_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// lanes 16-31 of a <-> lanes 0-15 of b, lanes 48-63 of a <-> lanes 32-47 of b (inline asm: the builtin of this toolchain returns
// the same register for both results, see conv.hip lanes_sum)
__device__ __forceinline__ void pp_swap16(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned pp_pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
__device__ __forceinline__ float pp_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pp_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

// Tile = 256 rows x 256 channels: TI0 = 8 sixteen-pixel tiles per pixel half.  NST: LDS stages (5 x 32 KiB = all 160 KiB of the CU; NST-1 K
// steps in flight).  STATS: batch-norm statistics of the output (forward).  ADD: + a.addend before the store (data gradient with a residual
// branch).  (Rows that do not fill a round of the chip go to the 128-row kernels of conv.hip; a taller tile that absorbed them was measured
// slower per row in round 2 and left the library in round 5.)
template <bool STATS, bool ADD>
__global__ __launch_bounds__(512) void conv_igemm_pp_kernel(const ConvArgs a) {
  constexpr int TI0 = 8;
  constexpr int S0 = 16 * TI0, BM = S0 + 128, BN = 256, BK = 32, NST = 5;
  constexpr int A_BYTES = BM * 64, ST_BYTES = A_BYTES + BN * 64;   // 64-byte LDS rows (32 bf16)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                       // pixel half (= ping-pong group), channel quarter
  const int l15 = lane & 15, lg = lane >> 4;
  // LDS-DMA instructions of this wave per K step: 2 + 2 (16 rows of each operand tile per instruction); the counted waits below assume 4 per step.

  // ---- tile schedule: this workgroup's position inside a round of gridDim.x tiles; XCD x owns a contiguous run of logical tiles
  const int G = gridDim.x, q8 = G >> 3, r8 = G & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int nt_n = (a.Cd + BN - 1) / BN;
  const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * nt_n;
  const int nmy = pos < tiles ? (tiles - pos + G - 1) / G : 0;
  const int ncs = (a.Cs + BK - 1) / BK;                         // channel slices (the last one may be ragged: Cs = 304)
  const int hw = a.Hd * a.Wd;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  // uniform description of my ti-th tile: first pixel / channel, valid kernel rows (rows that read only zero padding for EVERY
  // pixel of the tile are skipped: ASPP dilations 12/24/36 on a 65x65 map), number of K steps
  struct Tile { int m0, n0; unsigned trm; int nk; };
  auto tile_info = [&](int ti) {
    Tile t;
    const int lt = ti * G + pos;
    const int mt = nt_n == 1 ? lt : lt / nt_n;
    t.m0 = a.m_begin + mt * BM;
    t.n0 = (lt - mt * nt_n) * BN;
    t.trm = (1u << a.R) - 1;
    int nvr = a.R;
    if (a.R > 1) {
      const int mlast = min(t.m0 + BM, a.M) - 1;
      const int i0 = (int)fdiv((uint32_t)t.m0, a.fd_hw), i1 = (int)fdiv((uint32_t)mlast, a.fd_hw);
      const int h0 = (int)fdiv((uint32_t)(t.m0 - i0 * hw), a.fd_w), h1 = (int)fdiv((uint32_t)(mlast - i1 * hw), a.fd_w);
      if (i1 - i0 <= 1) {
        const int alo = h0, ahi = i1 == i0 ? h1 : a.Hd - 1, blo = i1 == i0 ? h0 : 0, bhi = h1;
        unsigned msk = 0;
        int cnt = 0;
#pragma unroll 1
        for (int r = 0; r < a.R; ++r) {
          const int o = a.mode == 0 ? r * a.dil - a.pad : a.pad - r * a.dil;     // source row = output row + o (stride 1 whenever R > 1)
          const bool v = (alo + o <= a.Hs - 1 && ahi + o >= 0) || (blo + o <= a.Hs - 1 && bhi + o >= 0);
          if (v) { msk |= 1u << r; ++cnt; }
        }
        if (cnt > 0) { t.trm = msk; nvr = cnt; }
      }
    }
    t.nk = ncs * nvr * a.S;
    return t;
  };

  // ---- issue side (LDS-DMA producer state, NST-1 K steps ahead of the MFMAs) --------------------------------------------------
  // thread -> rows wave*32 + 16 i + (lane >> 2) of BOTH operand tiles (i = 0, 1), 16-byte position lane & 3 of the row;
  // the chunk stored at position p of row r is source chunk p ^ f((r >> 2) & 3), f = {2,0,1,3}: with 64-byte rows the 16 lanes of
  // every ds_read_b128 group of the 16x16x32 operand reads (rows r..r+15 at chunks c, c, c+1, c+1 per quad) then hit 16 distinct
  // 16-byte slots of the 256-byte bank row
  const int prow = wave * 32 + (lane >> 2);       // (which wave loads a row is independent of which half consumes it)
  const int cch = (lane & 3) ^ ((0xD2 >> (2 * ((lane >> 4) & 3))) & 3);
  int it_ti = 0, it_cs = 0, it_tr = 0, it_ts = 0;
  Tile itile = {0, 0, 0, 0};
  // per-lane source description of a tile, [0..1] = my two rows:
  //   rowoff: byte offset of the tap-(0,0) source pixel (+ my chunk), may be out of range (see rmask)
  //   rmask : bit (tr*S + ts) = that tap of the row lies inside the source image
  //   boff  : byte offset of my weight rows (+ my chunk), or OOB
  // "cur" feeds the DMAs; "nxt" is the tile after it, prepared by the consumer side at the end of an epilogue (when the 128
  // accumulator registers are free) so that the LOAD segments never pay for a tile change
  constexpr int AR = 2;                   // A rows per thread
  int rowoff[AR], nrowoff[AR];
  unsigned rmask[AR], nrmask[AR];
  unsigned boff[2], nboff[2];
  bool it_live = false, it_need = true;
  int it_step = 0;        // K steps issued so far (all tiles)
  auto lane_setup = [&](int ti, int (&ro)[AR], unsigned (&rm)[AR], unsigned (&bo)[2]) {
    if (ti >= nmy) {
#pragma unroll
      for (int i = 0; i < AR; ++i) { rm[i] = 0; ro[i] = 0; }
      bo[0] = bo[1] = PP_OOB;
      return;
    }
    const int lt = ti * G + pos;
    const int mt = nt_n == 1 ? lt : lt / nt_n;
    const int m0 = a.m_begin + mt * BM, n0 = (lt - mt * nt_n) * BN;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = m0 + prow + 16 * i;
      unsigned msk = 0;
      int off = 0;
      if (m < a.M) {
        const uint32_t n_img = fdiv((uint32_t)m, a.fd_hw);
        const uint32_t rem = (uint32_t)m - n_img * (uint32_t)hw;
        const int hd = (int)fdiv(rem, a.fd_w);
        const int wd = (int)rem - hd * a.Wd;
        int h0, w0;      // source coordinate of tap (0,0)
        bool ok0 = true;
        if (a.mode == 0) {
          h0 = hd * a.stride - a.pad;
          w0 = wd * a.stride - a.pad;
        } else {
          h0 = hd + a.pad;
          w0 = wd + a.pad;
          if (a.stride == 2) {          // (1x1 only, checked by the launcher): the pixel has a source only at even coordinates
            ok0 = !((h0 | w0) & 1);
            h0 >>= 1;
            w0 >>= 1;
          }
        }
        off = (((int)n_img * a.Hs + h0) * a.Ws + w0) * a.lds * 2 + cch * 16;
        const int sgn = a.mode == 0 ? a.dil : -a.dil;
        unsigned bit = 1;
#pragma unroll 1
        for (int r = 0; r < a.R; ++r)
#pragma unroll 1
          for (int s = 0; s < a.S; ++s, bit <<= 1) {
            const int hs = h0 + sgn * r, ws = w0 + sgn * s;
            if (ok0 && (unsigned)hs < (unsigned)a.Hs && (unsigned)ws < (unsigned)a.Ws) msk |= bit;
          }
      }
      ro[i] = off;
      rm[i] = msk;
      if (i < 2) {
        const int n = n0 + prow + 16 * i;
        bo[i] = n < a.Cd ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)cch * 16u : PP_OOB;
      }
    }
  };
  // the issue side moves on to tile it_ti: take over the prepared lane state, reset the K position
  auto issue_tile_switch = [&]() {
    it_live = it_ti < nmy;
#pragma unroll
    for (int i = 0; i < AR; ++i) { rowoff[i] = nrowoff[i]; rmask[i] = nrmask[i]; }
    boff[0] = nboff[0];
    boff[1] = nboff[1];
    if (!it_live) return;
    itile = tile_info(it_ti);
    it_cs = 0;
    it_ts = 0;
    it_tr = __builtin_ctz(itile.trm);
  };
  const int tapstep = (a.mode == 0 ? a.dil : -a.dil) * a.lds * 2;     // bytes per kernel column; a kernel row is tapstep * Ws
  // next tap of the tile (valid kernel rows only); true when the taps wrapped around to the first one
  auto next_tap = [&]() -> bool {
    if (++it_ts < a.S) return false;
    it_ts = 0;
    const unsigned rest = itile.trm >> (it_tr + 1);
    if (rest) {
      it_tr += 1 + __builtin_ctz(rest);
      return false;
    }
    it_tr = __builtin_ctz(itile.trm);
    return true;
  };
  auto issue = [&](int stage) {
    if (it_need) {
      it_need = false;
      issue_tile_switch();
    }
    unsigned char* sa = smem + stage * ST_BYTES + wave * (32 * 64);
    const int tap = it_tr * a.S + it_ts;
    const bool cok = it_cs * BK + cch * 8 < a.Cs;                                  // ragged last channel slice
    const int da = (it_tr * a.Ws + it_ts) * tapstep + it_cs * (BK * 2);
    const unsigned kb = (unsigned)(tap * a.Cs + it_cs * BK) * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = cok && ((rmask[i] >> tap) & 1);
      pp_dma16(rs_a, sa + i * 1024, ok ? (unsigned)(rowoff[i] + da) : PP_OOB);
    }
    ++it_step;
#pragma unroll
    for (int i = 0; i < 2; ++i) pp_dma16(rs_b, sa + A_BYTES + i * 1024, (cok && boff[i] != PP_OOB) ? boff[i] + kb : PP_OOB);
    if (!it_live) return;
    // next K step of this tile.  korder 0: tap outer, channel slice inner (the order of the weight rows); 1: channel slice outer,
    // tap inner; 2: PAIRS of channel slices (= one 128-byte line per pixel) outer, tap, then the two halves of the line
    bool done = false;
    if (a.korder == 0) {
      if (++it_cs == ncs) {
        it_cs = 0;
        done = next_tap();
      }
    } else if (a.korder == 1) {
      if (next_tap()) done = ++it_cs == ncs;
    } else {
      if (!(it_cs & 1) && it_cs + 1 < ncs) {
        ++it_cs;
      } else {
        it_cs &= ~1;
        if (next_tap()) {
          it_cs += 2;
          done = it_cs >= ncs;
        }
      }
    }
    if (done) {
      ++it_ti;
      it_need = true;
    }
  };

  // ---- consumer state ---------------------------------------------------------------------------------------------------
  f32x4 acc[TI0][4];      // [pixel tile i: pixels 16 i + (lane & 15) of my half][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int ti_n = TI0;                                // pixel tiles of my half
  const int koff = ((lg ^ ((0xD2 >> (2 * ((lane >> 2) & 3))) & 3)) << 4);
  const int a_addr = (wm * S0 + l15) * 64 + koff, b_addr = A_BYTES + (wn * 64 + l15) * 64 + koff;

  lane_setup(0, nrowoff, nrmask, nboff);
  issue_tile_switch();                                 // tile 0 becomes current
  it_need = false;
  lane_setup(1, nrowoff, nrmask, nboff);
#pragma unroll 1
  for (int s0 = 0; s0 < NST - 1; ++s0) issue(s0);
  // all but the youngest NST-2 K steps of my LDS-DMAs (4 or 5 instructions each) have landed
  auto wait_dma = [&](bool stores_behind) {
    if (!stores_behind) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2)) : "memory");
    } else {
      // + the previous tile's stores, issued behind the DMAs of the first NST-2 K steps of this tile: at least 16 per wave, + 8
      // statistics stores
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2) + (STATS ? 24 : 16)) : "memory");
    }
  };
  wait_dma(false);
  __builtin_amdgcn_s_barrier();                        // every wave's share of K step 0 has landed
#ifndef PP_ABL_NOSTAGGER      // ablation: both pixel halves in phase (the schedule of round 1's kernel)
  if (wm == 1) __builtin_amdgcn_s_barrier();           // half a K step behind: LOAD of one half runs beside MFMA of the other
#endif
  asm volatile("" ::: "memory");
  int st_c = 0, st_i = NST - 1;                        // stage read by the next LOAD segment / filled by its issue
  int st_pending = 0;                                  // K steps during which the epilogue's stores may still be in flight

  for (int ti = 0; ti < nmy; ++ti) {
    const Tile ct = tile_info(ti);
    for (int kt = 0; kt < ct.nk; ++kt) {
      // ---------------- LOAD segment ----------------
      const unsigned char* sb = smem + st_c * ST_BYTES;
      bf16x8 fa[TI0], fw[4];
#ifdef PP_ABL_NOCOMPUTE       // ablation: LDS-DMA stream only (no fragment reads, one MFMA per step keeps the accumulators alive)
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(smem + b_addr + j * 1024);
#pragma unroll
      for (int i = 0; i < TI0; ++i) fa[i] = fw[i & 3];
      if (kt > 1000000) {
#pragma unroll
        for (int i = 0; i < TI0; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sb + a_addr + i * 1024);
      }
#else
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(sb + b_addr + j * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sb + a_addr + i * 1024);
#endif
      issue(st_i);                                     // K step + NST-1 (past the last tile: all-OOB = zeros into a free stage)
      st_c = st_c == NST - 1 ? 0 : st_c + 1;
      st_i = st_i == NST - 1 ? 0 : st_i + 1;
      // my share of the NEXT K step has landed.  The previous tile's stores (16 per wave, + 8 statistics stores) were issued
      // behind the DMAs of the first NST-2 K steps of this tile: while one of those is the step waited for they may stay in flight
      wait_dma(st_pending > 0);
      if (st_pending > 0) --st_pending;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my fragment reads are done: the stage may be refilled after the barrier
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA segment ----------------
      __builtin_amdgcn_s_setprio(1);
#ifdef PP_ABL_NOCOMPUTE
      if (kt == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TI0; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = z;
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fa[0], acc[0][0], 0, 0, 0);
#else
      if (kt == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], z, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
      }
#endif
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");                   // (the next LOAD segment's fragment reads must stay behind this barrier)
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- epilogue of tile ti (no LDS, no barrier, no load but the optional addend) ----------------
    const int mrow0 = ct.m0 + wm * S0, n0w = ct.n0 + wn * 64;
    // a wave's 128 rows are one "slab" of a.stats: its row holds the sums of the rows that belong to the
    // statistics group of the slab's FIRST row; rows of a slab past a group boundary (< 144 per boundary) are summed from the
    // stored tensor by stage 2 (bn_reduce_slabs_kernel)
    const int bnd = STATS ? (mrow0 / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;
    // lane -> 8 consecutive channels of one pixel after the swaps below:
    //   lane group g = 0: tile j ch 0-7 | g = 1: tile j+1 ch 0-7 | g = 2: tile j ch 8-15 | g = 3: tile j+1 ch 8-15   (j = 0, 2)
    const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);
    pp_u32x4 radd[ADD ? TI0 : 1][2];
    const bool has_mask = ADD && a.add_mask != nullptr;
    if (ADD) {
      // all 16 addend vectors of the wave tile are requested before the first one is used: ONE drain of the vector-memory queue per
      // tile (the compiler waits for an ordinary load with everything older, i.e. with the LDS-DMA of the next tile's first K steps)
      const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.addend), 0, (int)a.add_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < TI0; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mrow0 + 16 * i + l15, n = nl + 32 * h;
          radd[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((i < ti_n && m < a.M && n < a.Cd) ? ((unsigned)m * (unsigned)a.ld_add + (unsigned)n) * 2u : PP_OOB), 0, 0);
        }
    }
    auto store_pixel_tile = [&](auto ic) {
      constexpr int i = decltype(ic)::value;           // (static accumulator indices: a run-time index sends acc to scratch memory)
      const int m = mrow0 + 16 * i + l15;
      unsigned lo[4], hi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = pp_pack2(acc[i][j][0], acc[i][j][1]);
        hi[j] = pp_pack2(acc[i][j][2], acc[i][j][3]);
      }
      const unsigned rowb = (unsigned)m * (unsigned)a.ldd * 2u;
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        pp_swap16(lo[jp], lo[jp + 1]);
        pp_swap16(hi[jp], hi[jp + 1]);
        pp_u32x4 v = {lo[jp], hi[jp], lo[jp + 1], hi[jp + 1]};
        const int n = nl + 16 * jp;
        const bool ok = m < a.M && n < a.Cd;
        if (ADD) {
          // dgrad: + the residual branch's gradient (bf16 + bf16 in fp32, rounded once: what autograd's add would give)
          pp_u32x4 r = radd[ADD ? i : 0][jp >> 1];
          // optional ReLU bit mask of the addend (css_conv2d_dgrad_add_masked), one byte per 16-byte vector: fetched where it is used (this
          // kernel is a fallback of conv_igemm_p8_kernel, which requests the bytes together with the addend; 16 more registers spill here)
          unsigned mk = 0xFFu;
          if (has_mask) {
            const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.add_mask), 0, (int)a.mask_bytes, 0x00020000);
            mk = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_k, (int)(ok ? (unsigned)m * ((unsigned)a.Cd >> 3) + ((unsigned)n >> 3) : PP_OOB), 0, 0);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] &= keep_mask_bf16x2(mk, e);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pp_pack2(pp_lo(v[e]) + pp_lo(r[e]), pp_hi(v[e]) + pp_hi(r[e]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(ok ? rowb + (unsigned)n * 2u : PP_OOB), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);              // one pixel tile at a time: short live ranges next to the 128 accumulators
    };
    store_pixel_tile(std::integral_constant<int, 0>());
    store_pixel_tile(std::integral_constant<int, 1>());
    store_pixel_tile(std::integral_constant<int, 2>());
    store_pixel_tile(std::integral_constant<int, 3>());
    store_pixel_tile(std::integral_constant<int, 4>());
    store_pixel_tile(std::integral_constant<int, 5>());
    store_pixel_tile(std::integral_constant<int, 6>());
    store_pixel_tile(std::integral_constant<int, 7>());
    if (STATS) {
      // Batch-norm statistics of exactly the bf16 values stored (what bn_apply reads back; rows >= M are exact zeros), one channel
      // tile at a time (8 live sums next to the 128 accumulators): per lane its 8 pixels, then the 16 pixels of a lane row by DPP;
      // lane 0 of each row stores 4 consecutive channels into the slab's row of a.stats.  Always 8 store instructions
      // (out-of-range offsets are dropped): the vmcnt arithmetic of the LOAD segments counts on them.
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(2 * ((ct.m0 - a.m_begin) / BM + a.m_begin / 256) + wm) * 2u * (unsigned)a.Cd * 4u;   // slab index: 2 per tile
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TI0; ++i) {
          f32x4 t = acc[i][j];
          asm volatile("" : "+v"(t));       // opaque: otherwise the packed values of the store loop stay alive (CSE) across the epilogue
          const unsigned lo = pp_pack2(t[0], t[1]), hi = pp_pack2(t[2], t[3]);
          // rows of the next statistics group are left to stage 2; the ninth tile of the second half does not exist (its registers
          // hold garbage: select, do not multiply)
          const bool keep = mrow0 + 16 * i + l15 < bnd && i < ti_n;
          const float v0 = keep ? pp_lo(lo) : 0.f, v1 = keep ? pp_hi(lo) : 0.f, v2 = keep ? pp_lo(hi) : 0.f, v3 = keep ? pp_hi(hi) : 0.f;
          ss[0] += v0; ss[1] += v1; ss[2] += v2; ss[3] += v3;
          qq[0] += v0 * v0; qq[1] += v1 * v1; qq[2] += v2 * v2; qq[3] += v3 * v3;
        }
        pp_f32x4 os, oq;
#pragma unroll
        for (int r = 0; r < 4; ++r) { os[r] = pp_row16_sum(ss[r]); oq[r] = pp_row16_sum(qq[r]); }
        const int n = n0w + 16 * j + 4 * lg;
        const bool lane_ok = l15 == 0 && n < a.Cd && mrow0 < a.M;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_u32x4, os), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : PP_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_u32x4, oq), rs_s, (int)(lane_ok ? base + (unsigned)(a.Cd + n) * 4u : PP_OOB), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st_pending = NST - 2;
    __builtin_amdgcn_sched_barrier(0);
    // the issue side is inside tile ti+1 by now (every tile has more K steps than stages): prepare tile ti+2 for it
    lane_setup(ti + 2, nrowoff, nrmask, nboff);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifndef PP_ABL_NOSTAGGER
  if (wm == 0) __builtin_amdgcn_s_barrier();           // the barrier the other half ran at the start
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ghost DMAs must have landed before the workgroup's LDS is released
}

// ---- launcher -------------------------------------------------------------------------------------------------------------
// Supported: bf16, Cout >= 256, no bias, stride 1 (any kernel size up to 31 taps) or 1x1 with stride 1 / 2, at least 4 K steps per tile.
bool css_conv_pp_supported(const ConvArgs& a) {
  static const bool off = getenv("CSS_NO_PP_CONV") != nullptr;
  if (off || a.bias || a.Cd < 256 || a.Cs % 8 || a.R * a.S > 31 || a.R >= 31) return false;
  if (a.stride != 1 && !(a.R == 1 && a.S == 1 && a.stride == 2)) return false;
  if ((a.Cd % 8) || (a.ldd % 8) || (reinterpret_cast<uintptr_t>(a.dst) & 15)) return false;
  if (a.addend && ((a.ld_add % 8) || (reinterpret_cast<uintptr_t>(a.addend) & 15) || (size_t)a.M * a.ld_add * 2 >= 0x7FFFFFF0ull)) return false;
  if (a.stats && ((a.Cd % 4) || a.addend || (reinterpret_cast<uintptr_t>(a.stats) & 15))) return false;
  const int ncs = (a.Cs + 31) / 32;
  if (ncs * a.S < 5) return false;                       // (a tile with one valid kernel row still has more K steps than stages)
  return true;
}

// 256 when the persistent 256x256-tile kernels take the launch (whole rounds of the chip there, leftover rows on the 128-row kernels of
// conv.hip); 0: not a shape for them.
int css_conv_pp_plan(const ConvArgs& a, int n_cu) {
  (void)n_cu;
  if (!css_conv_pp_supported(a) || (size_t)a.M * a.ldd * 2 >= 0x7FFFFFF0ull) return 0;
  return 256;
}

void css_launch_conv_pp(ConvArgs a, int grid, hipStream_t st) {
  a.fd_hw = make_fastdiv((uint32_t)(a.Hd * a.Wd));
  a.fd_w = make_fastdiv((uint32_t)a.Wd);
  // K order (see the kernel's issue()): channel slice outer / tap inner for kernels with more than one tap
  static const int korder_env = getenv("CSS_PP_KORDER") ? atoi(getenv("CSS_PP_KORDER")) : -1;
  a.korder = korder_env >= 0 ? korder_env : (a.R * a.S > 1 ? 1 : 0);   // (measured on the harness: 1 beats 0 by 3-5 % on the 3x3 shapes, 2 loses)
  if (a.stats) a.stat_bytes = (unsigned)((size_t)2 * cdiv(a.M, 256) * 2 * a.Cd * 4);
  if (a.addend) a.add_bytes = (unsigned)((size_t)a.M * a.ld_add * 2);
  if (a.add_mask) a.mask_bytes = (unsigned)((size_t)a.M * (a.Cd / 8));
  const dim3 g(grid), b(512);
  if (a.stats) hipLaunchKernelGGL((conv_igemm_pp_kernel<true, false>), g, b, 0, st, a);
  else if (a.addend) hipLaunchKernelGGL((conv_igemm_pp_kernel<false, true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_pp_kernel<false, false>), g, b, 0, st, a);
}
