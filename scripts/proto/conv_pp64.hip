// conv_igemm_pp64_kernel: conv_igemm_pp_kernel (conv_pp.hip: persistent, ping-pong pixel halves, register epilogue) with K steps of
// 64 channels, i.e. 128-BYTE rows in global memory and LDS.
//
// Why: the ablations of conv_pp.hip (PP_ABL_NOLOAD / PP_ABL_NOCOMPUTE, profiles/r02_conv_ablation.txt) show the 256x256 kernels
// bound by the per-CU vector-memory path: with 64-byte rows (K = 32) an LDS-DMA instruction touches 16 half cache lines and the
// global -> LDS stream of a CU runs at ~13-17 B/clk, whatever serves it (HBM, Infinity Cache or L2) and however many stages are in
// flight (4 vs 5 stages, tap-inner vs tap-outer K order: +-3 %); 32 KiB per 1024 MFMA cycles of work is then 40 % of the MFMA rate.
// With 128-byte rows an instruction touches 8 whole lines: ~27 B/clk measured on the loads-only ablation of round 1's 256x128x64
// kernel.  LDS: three pixel-tile buffers (A: 256 rows x 128 B = 32 KiB each, streamed from HBM / Infinity Cache: two K steps in
// flight) and two weight-tile buffers (B: L2-resident, one K step in flight) = 160 KiB.  The in-order vmcnt allows the asymmetry
// because within a LOAD segment the sooner-needed tile is issued first: B(s+1) in the first half of K step s, A(s+2) in the second,
// one `s_waitcnt vmcnt(4)` per K step (A(s+2) keeps flying).
//
// Everything else as in conv_pp.hip: 16x16x32 MFMA, two s_barrier per K = 32 sub-step, pixel halves half a sub-step apart,
// channel-slice-outer / tap-inner K order, the epilogue from registers with the statistics slabs.
// ARCHIVED in round 4: no default shape runs on this kernel since conv_igemm_p8_kernel (css_amd/csrc/conv_p8.hip, bit-identical
// outputs) replaced it; kept as the A/B reference the harnesses under scripts/ time p8 against.  Not part of libcss_hip.so.
#include "../../css_amd/csrc/common.h"
#include "../../css_amd/csrc/launchers.h"
bool css_conv_pp64_supported(const ConvArgs& a);
void css_launch_conv_pp64(ConvArgs a, int grid, hipStream_t st);
#include <cstdlib>
#include <type_traits>

namespace {
typedef __attribute__((address_space(3))) void p6_lds_void;
constexpr unsigned P6_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int p6_u32x4;
typedef __attribute__((ext_vector_type(4))) float p6_f32x4;

__device__ __forceinline__ void p6_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (p6_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ float p6_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ void p6_swap16(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned p6_pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
__device__ __forceinline__ float p6_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float p6_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

template <bool STATS, bool ADD>
__global__ __launch_bounds__(512) void conv_igemm_pp64_kernel(const ConvArgs a) {
  constexpr int BM = 256, BN = 256, BK = 64, NA = 3, NB = 2;
  constexpr int BUF = 256 * 128;                                 // one operand tile: 256 rows of 128 bytes
  __shared__ __attribute__((aligned(1024))) unsigned char smem[(NA + NB) * BUF];
  unsigned char* const smem_b = smem + NA * BUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                       // pixel half (= ping-pong group), channel quarter
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- tile schedule (as conv_pp.hip) ----
  const int G = gridDim.x, q8 = G >> 3, r8 = G & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int nt_n = (a.Cd + BN - 1) / BN;
  const int tiles = ((a.M - a.m_begin + BM - 1) / BM) * nt_n;
  const int nmy = pos < tiles ? (tiles - pos + G - 1) / G : 0;
  const int ncs = (a.Cs + BK - 1) / BK;                         // 64-channel slices (the last one may be ragged: Cs = 304)
  const int hw = a.Hd * a.Wd;

  // (opaque copies: under SGPR pressure the compiler otherwise re-loads these kernel arguments from memory inside every LOAD
  // segment - an s_load and an lgkmcnt(0) wait ahead of the LDS-DMA issue; a spilled SGPR costs one v_readlane instead)
  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

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

  // ---- issue side: two streams over the same (tile, K step) sequence: A two K steps ahead of the MFMAs, B one ------------------
  // thread -> rows wave*32 + 8 i + (lane >> 3) of an operand tile (i = 0..3), 16-byte position lane & 7 of the 128-byte row; the
  // chunk stored at position p of row r is source chunk p ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of the 16x16x32 operands:
  // rows r..r+15 at chunks c, c, c+1, c+1 per quad hit 16 distinct 16-byte slots of the 256-byte bank row = 2 LDS rows).
  // (r >> 1) & 7 = (4 i + (lane >> 4)) & 7: rows i = 1, 3 take the chunk of rows i = 0, 2 with bit 2 flipped.
  const int prow = wave * 32 + (lane >> 3);
  const int cch0 = (lane & 7) ^ ((lane >> 4) & 3);
  // Position of a stream inside its tile: channel slice cs and index it into the tile's list of taps (a.tab_*: one list per set of
  // valid kernel rows, built by the launcher).  The bookkeeping of a K step is a handful of scalar instructions: in the first
  // version the (kernel row, column, slice) iteration with its valid-row bit tricks cost ~280 SALU instructions per LOAD segment
  // - more than the 32 MFMAs of the other pixel half take.
  // The launcher's tap tables (63 entries each) live in three VGPRs, entry e in lane e: a lookup is one v_readlane, where indexing
  // the kernel-argument copy costs an s_load and an lgkmcnt(0) wait (which also waits for every fragment read in flight).
  const int tabv_da = a.tab_da[lane < 63 ? lane : 62], tabv_kb = a.tab_kb[lane < 63 ? lane : 62], tabv_tap = a.tab_tap[lane < 63 ? lane : 62];
  struct KPos { int ti, cs, it, nt, vb; bool live, need; };
  KPos pa = {0, 0, 0, 1, 0, false, true}, pb = {0, 0, 0, 1, 0, false, true};
  int rowoff[4], nrowoff[4];      // byte offset of the tap-(0,0) source pixel of my rows (+ my chunk), may be out of range: see rmask
  unsigned rmask[4], nrmask[4];   // bit (tr*S + ts): that tap of the row lies inside the source image
  unsigned boff[4], nboff[4];     // byte offset of my weight rows (+ my chunk), or OOB
  auto lane_setup = [&](int ti, int (&ro)[4], unsigned (&rm)[4], unsigned (&bo)[4]) {
    if (ti >= nmy) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { rm[i] = 0; ro[i] = 0; bo[i] = P6_OOB; }
      return;
    }
    const int lt = ti * G + pos;
    const int mt = nt_n == 1 ? lt : lt / nt_n;
    const int m0 = a.m_begin + mt * BM, n0 = (lt - mt * nt_n) * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cch = cch0 ^ ((i & 1) << 2);
      const int m = m0 + prow + 8 * i;
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
      const int n = n0 + prow + 8 * i;
      bo[i] = n < a.Cd ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)cch * 16u : P6_OOB;
    }
  };
  // a stream moves on to tile p.ti
  auto kpos_switch = [&](KPos& p) {
    p.need = false;
    p.live = p.ti < nmy;
    if (!p.live) return;
    const Tile t = tile_info(p.ti);
    p.cs = 0;
    p.it = 0;
    p.nt = __builtin_popcount(t.trm) * a.S;
    p.vb = ((int)t.trm - 1) * 9;
  };
  // next K step of a stream: channel slice outer, tap inner (korder 1) or the reverse (korder 0)
  auto kpos_next = [&](KPos& p) {
    if (!p.live) return;
    bool done = false;
    if (a.korder == 0) {
      if (++p.cs == ncs) {
        p.cs = 0;
        done = ++p.it == p.nt;
      }
    } else {
      if (++p.it == p.nt) {
        p.it = 0;
        done = ++p.cs == ncs;
      }
    }
    if (done) {
      ++p.ti;
      p.need = true;
    }
  };
  // An operand tile is issued as four 1-KiB pieces per wave: *_begin = the K step's scalars (and the tile switch), *_piece(i) = one
  // LDS-DMA instruction.  In the K loop they are issued INSIDE the MFMA segments (see there); the prologue issues whole tiles.
  unsigned char* sa_cur = smem;
  int tap_cur = 0, da_cur = 0, a_lim = 0;
  auto a_begin = [&](int buf) {
    if (pa.need) {
      kpos_switch(pa);
#pragma unroll
      for (int i = 0; i < 4; ++i) { rowoff[i] = nrowoff[i]; rmask[i] = nrmask[i]; }
    }
    sa_cur = smem + buf * BUF + wave * (32 * 128);
    const int ix = pa.vb + pa.it;
    tap_cur = __builtin_amdgcn_readlane(tabv_tap, ix);
    da_cur = __builtin_amdgcn_readlane(tabv_da, ix) + pa.cs * (BK * 2);
    a_lim = a.Cs - pa.cs * BK;                               // channels left in this slice (ragged last slice)
    kpos_next(pa);
  };
  auto a_piece = [&](int i) {
    const bool ok = (cch0 ^ ((i & 1) << 2)) * 8 < a_lim && ((rmask[i] >> tap_cur) & 1);
    p6_dma16(rs_a, sa_cur + i * 1024, ok ? (unsigned)(rowoff[i] + da_cur) : P6_OOB);
  };
  unsigned char* sb_cur = smem;
  unsigned kb_cur = 0;
  int b_lim = 0;
  auto b_begin = [&](int buf) {
    if (pb.need) {
      kpos_switch(pb);
#pragma unroll
      for (int i = 0; i < 4; ++i) boff[i] = nboff[i];
    }
    sb_cur = smem_b + buf * BUF + wave * (32 * 128);
    kb_cur = (unsigned)(__builtin_amdgcn_readlane(tabv_kb, pb.vb + pb.it) + pb.cs * (BK * 2));
    b_lim = pb.live ? a.Cs - pb.cs * BK : 0;
    kpos_next(pb);
  };
  auto b_piece = [&](int i) {
    const bool ok = (cch0 ^ ((i & 1) << 2)) * 8 < b_lim && boff[i] != P6_OOB;
    p6_dma16(rs_b, sb_cur + i * 1024, ok ? boff[i] + kb_cur : P6_OOB);
  };
  auto issue_a = [&](int buf) {
    a_begin(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_piece(i);
  };
  auto issue_b = [&](int buf) {
    b_begin(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) b_piece(i);
  };

  // ---- consumer state ----
  f32x4 acc[8][4];        // [pixel tile i: pixels 16 i + (lane & 15) of my half][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int sw = (l15 >> 1) & 7;
  const int koff0 = ((lg ^ sw) << 4), koff1 = (((4 + lg) ^ sw) << 4);       // k halves 0 / 1 of a 128-byte row
  const int a_row = (wm * 128 + l15) * 128, b_row = (wn * 64 + l15) * 128;

  // tile 0 -> both streams' current state (via the "next" slots), tile 1 -> next
  lane_setup(0, nrowoff, nrmask, nboff);
  issue_a(0);                        // A(0)
  issue_b(0);                        // B(0)
  lane_setup(1, nrowoff, nrmask, nboff);
  issue_a(1);                        // A(1)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // A(0), B(0) landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();           // half a sub-step behind: LOAD of one half runs beside MFMA of the other
  asm volatile("" ::: "memory");
  int ca = 0, cb = 0;                // buffers read by the current K step
  int ia = 2, ib = 1;                // buffers filled next (A two steps ahead, B one)

  for (int ti = 0; ti < nmy; ++ti) {
    const Tile ct = tile_info(ti);
    for (int kt = 0; kt < ct.nk; ++kt) {
      const unsigned char* ab = smem + ca * BUF + a_row;
      const unsigned char* bb = smem_b + cb * BUF + b_row;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // ---------------- LOAD segment of K half h ----------------
        const int ko = h ? koff1 : koff0;
        bf16x8 fa[8], fw[4];
        // Both segments of the ping-pong are critical (a phase lasts max(LOAD of one half, MFMA of the other), and they are about
        // equal), so the LOAD segment holds: the 12 fragment reads, the step's scalars (~35 SALU, no memory access: the tap tables
        // sit in VGPR lanes - they run while the reads are in flight), the four LDS-DMA pieces of this K half - B(s+1) in half 0 (its
        // buffer was last read one K step ago), A(s+2) in half 1 - and the waits.  Measured alternatives (profiles/
        // r02_conv_ablation.txt section 6): pieces or scalars inside the MFMA segment +10..15 % time (they stall MFMA issue),
        // pieces ahead of the reads or directly behind them +3..5 % (an LDS-DMA issued into a queue of ds_reads is slow).
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(bb + j * 2048 + ko);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + ko);
        if (h == 0) { b_begin(ib); ib ^= 1; }
        else { a_begin(ia); ia = ia == NA - 1 ? 0 : ia + 1; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) { if (h == 0) b_piece(p); else a_piece(p); }
        // my share of the next K step (A and B) has landed; A(+2) keeps flying.  (vmcnt is in order and counts stores: in the first K
        // step of a tile this also waits for the previous epilogue's stores; issuing B(1), A(2) ahead of those stores was tried - the
        // K loop with the extra first-step case ran 12-15 % slower on every shape, profiles/r02_conv_ablation.txt section 6)
        if (h == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my fragment reads are done (buffers may be refilled after the barrier)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA segment ----------------
#ifndef P6_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        if (kt == 0 && h == 0) {
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
#ifndef P6_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                   // (the next LOAD segment's fragment reads must stay behind this barrier)
        __builtin_amdgcn_sched_barrier(0);
      }
      ca = ca == NA - 1 ? 0 : ca + 1;
      cb ^= 1;
    }

    // ---------------- epilogue of tile ti (as conv_pp.hip: no LDS, no barrier) ----------------
    const int mrow0 = ct.m0 + wm * 128, n0w = ct.n0 + wn * 64;
    const int bnd = STATS ? (mrow0 / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;     // rows >= bnd: next statistics group (stage 2 sums them)
    const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);
    p6_u32x4 radd[ADD ? 8 : 1][2];
    const bool has_mask = ADD && a.add_mask != nullptr;
    if (ADD) {
      const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.addend), 0, (int)a.add_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mrow0 + 16 * i + l15, n = nl + 32 * h;
          radd[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((m < a.M && n < a.Cd) ? ((unsigned)m * (unsigned)a.ld_add + (unsigned)n) * 2u : P6_OOB), 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mrow0 + 16 * i + l15;
      unsigned lo[4], hi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = p6_pack2(acc[i][j][0], acc[i][j][1]);
        hi[j] = p6_pack2(acc[i][j][2], acc[i][j][3]);
      }
      const unsigned rowb = (unsigned)m * (unsigned)a.ldd * 2u;
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        p6_swap16(lo[jp], lo[jp + 1]);
        p6_swap16(hi[jp], hi[jp + 1]);
        p6_u32x4 v = {lo[jp], hi[jp], lo[jp + 1], hi[jp + 1]};
        const int n = nl + 16 * jp;
        const bool ok = m < a.M && n < a.Cd;
        if (ADD) {
          p6_u32x4 r = radd[ADD ? i : 0][jp >> 1];
          // optional ReLU bit mask of the addend (css_conv2d_dgrad_add_masked), one byte per 16-byte vector: fetched where it is used (this
          // kernel is a fallback of conv_igemm_p8_kernel, which requests the bytes together with the addend; 16 more registers spill here)
          unsigned mk = 0xFFu;
          if (has_mask) {
            const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.add_mask), 0, (int)a.mask_bytes, 0x00020000);
            mk = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_k, (int)(ok ? (unsigned)m * ((unsigned)a.Cd >> 3) + ((unsigned)n >> 3) : P6_OOB), 0, 0);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] &= keep_mask_bf16x2(mk, e);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p6_pack2(p6_lo(v[e]) + p6_lo(r[e]), p6_hi(v[e]) + p6_hi(r[e]));
        }
#if defined(P6_ABL_NOSTORE)       // ablation: keep the values alive, drop the stores
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#elif defined(P6_ABL_NTSTORE)     // ablation: non-temporal stores
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(ok ? rowb + (unsigned)n * 2u : P6_OOB), 0, 2);
#else
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(ok ? rowb + (unsigned)n * 2u : P6_OOB), 0, 0);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (STATS) {
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(mrow0 >> 7) * 2u * (unsigned)a.Cd * 4u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x4 t = acc[i][j];
          asm volatile("" : "+v"(t));       // opaque: otherwise the packed values of the store loop stay alive (CSE) across the epilogue
          const unsigned lo = p6_pack2(t[0], t[1]), hi = p6_pack2(t[2], t[3]);
          const bool keep = mrow0 + 16 * i + l15 < bnd;
          const float v0 = keep ? p6_lo(lo) : 0.f, v1 = keep ? p6_hi(lo) : 0.f, v2 = keep ? p6_lo(hi) : 0.f, v3 = keep ? p6_hi(hi) : 0.f;
          ss[0] += v0; ss[1] += v1; ss[2] += v2; ss[3] += v3;
          qq[0] += v0 * v0; qq[1] += v1 * v1; qq[2] += v2 * v2; qq[3] += v3 * v3;
        }
        p6_f32x4 os, oq;
#pragma unroll
        for (int r = 0; r < 4; ++r) { os[r] = p6_row16_sum(ss[r]); oq[r] = p6_row16_sum(qq[r]); }
        const int n = n0w + 16 * j + 4 * lg;
        const bool lane_ok = l15 == 0 && n < a.Cd && mrow0 < a.M;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p6_u32x4, os), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : P6_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p6_u32x4, oq), rs_s, (int)(lane_ok ? base + (unsigned)(a.Cd + n) * 4u : P6_OOB), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // both streams are inside tile ti+1 by now (every tile has at least three K steps): prepare tile ti+2 for them
    lane_setup(ti + 2, nrowoff, nrmask, nboff);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();           // the barrier the other half ran at the start
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ghost DMAs must have landed before the workgroup's LDS is released
}

// Supported: what conv_pp.hip supports, with at least three 64-channel K steps per tile and a whole number of 16-byte chunks.
bool css_conv_pp64_supported(const ConvArgs& a) {
  static const bool off = getenv("CSS_NO_PP64_CONV") != nullptr;
  if (off || !css_conv_pp_supported(a) || a.R > 3 || a.S > 3) return false;     // (tap tables: 7 sets of valid kernel rows x 9 taps)
  const int ncs = (a.Cs + 63) / 64;
  return ncs * a.S >= 3;
}

void css_launch_conv_pp64(ConvArgs a, int grid, hipStream_t st) {
  a.fd_hw = make_fastdiv((uint32_t)(a.Hd * a.Wd));
  a.fd_w = make_fastdiv((uint32_t)a.Wd);
  static const int korder_env = getenv("CSS_PP_KORDER") ? atoi(getenv("CSS_PP_KORDER")) : -1;
  a.korder = korder_env >= 0 ? (korder_env != 0) : (a.R * a.S > 1 ? 1 : 0);
  if (a.stats) a.stat_bytes = (unsigned)((size_t)2 * cdiv(a.M, 256) * 2 * a.Cd * 4);
  if (a.addend) a.add_bytes = (unsigned)((size_t)a.M * a.ld_add * 2);
  if (a.add_mask) a.mask_bytes = (unsigned)((size_t)a.M * (a.Cd / 8));
  // tap lists: for every non-empty set v of valid kernel rows (bit r of v: row r reads something but padding for the tile), the taps
  // (r, s) in order with the byte offset of the source pixel relative to tap (0,0), the byte offset inside a weight row, and the bit
  // index of the tap in the per-pixel validity masks
  const int tapstep = (a.mode == 0 ? a.dil : -a.dil) * a.lds * 2;
  for (int v = 1; v < (1 << a.R); ++v) {
    int n = 0;
    for (int r = 0; r < a.R; ++r) {
      if (!((v >> r) & 1)) continue;
      for (int sx = 0; sx < a.S; ++sx, ++n) {
        const int ix = (v - 1) * 9 + n;
        a.tab_da[ix] = (r * a.Ws + sx) * tapstep;
        a.tab_kb[ix] = (r * a.S + sx) * a.Cs * 2;
        a.tab_tap[ix] = r * a.S + sx;
      }
    }
  }
  const dim3 g(grid), b(512);
  if (a.stats) hipLaunchKernelGGL((conv_igemm_pp64_kernel<true, false>), g, b, 0, st, a);
  else if (a.addend) hipLaunchKernelGGL((conv_igemm_pp64_kernel<false, true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_pp64_kernel<false, false>), g, b, 0, st, a);
}
