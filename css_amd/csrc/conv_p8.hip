// conv_igemm_p8_kernel: the persistent 256x256 implicit-GEMM convolution of scripts/proto/conv_pp64.hip with its K loop rebuilt in the 8-phase
// structure of the guide's 256^2 GEMM template (cdna_hip_programming.md, "The 256^2 8-phase template").
//
// Why (profiles/r03_yardstick_gemm8p_vs_pp64.txt, same box, uniform random operands): scripts/gemm8p.hip - that template written from
// its specification, one workgroup per tile, no persistence - runs the GEMMs of the conv shapes 5-29 % faster than
// conv_igemm_pp64_kernel (M = 131072: N = 256, K = 4608: 1056 vs 886 TFLOP/s; N = 512, K = 4608: 1290 vs 999; 8192^3: 1357 vs 1249 on
// 65536 x 4096 x 4096), so pp64's coarse LOAD / MFMA segments (12 ds_read_b128 + 4 LDS-DMA | 32 MFMAs, two barriers per K = 32) leave
// that much on the table - it was not the vector-memory path of the CU (scripts/proto/ta_path_bench.hip: 68-103 GB/s of LDS-DMA per
// CU from L2, against the ~25 GB/s this kernel needs).
//
// Structure.  8 waves as 2 pixel halves (wm) x 4 channel quarters (wn), 128 x 64 outputs per wave (32 accumulator tiles of
// v_mfma_f32_16x16x32_bf16) as in pp64 - same epilogue from registers, same statistics slabs, same persistent tile schedule, same
// tap tables; the outputs are bit-identical to pp64's (same instruction on the same K blocks in the same order).  What changed:
//   * LDS = 2 buffers x 4 HALF-tiles of 16 KiB (128 rows x 128 B): A0 / A1 = pixel rows [0, 64) / [64, 128) of BOTH pixel halves,
//     B0 / B1 = channels [0, 32) / [32, 64) of all four channel quarters, so that every wave needs (A0, B0) for the first quadrant of
//     its block, then B1, then A1: a K step (64 channels of one tap) is four phases of ONE 64 x 32 quadrant = 16 MFMAs each:
//       phase 1: read B0 (4 ds_read_b128), A0 (8);  stage A1 of step t+1;  (A0 x B0)
//       phase 2: read B1 (4);                       stage B0 of step t+2;  (A0 x B1)
//       phase 3: read A1 (8);                       stage A0 of step t+2;  (A1 x B1)
//       phase 4: -                                  stage B1 of step t+2; scalars of step t+3; s_waitcnt vmcnt(6);  (A1 x B0)
//     (the per-lane offsets of a phase's two LDS-DMA pieces are computed inside the MFMA segment of the phase before)
//     each phase = reads + 2 LDS-DMA pieces per thread | s_barrier | lgkmcnt(0) | 16 MFMAs under s_setprio 1 | s_barrier; the two
//     pixel halves (= the two waves of a SIMD) run one barrier apart, so the reads / DMA issue of one overlap the MFMAs of the other.
//   * ONE counted wait per K step (phase 4: three half-tiles stay in flight), never vmcnt(0) in the loop.
//   * Hazards (the template's rules): a half-tile is read one phase AFTER the wait that retires it (phase 4's wait -> phases 1-3 of
//     the next step); a buffer is restaged two phases after its last read, or one phase after when an lgkmcnt BEFORE the reading
//     phase's first barrier retired those reads (phase 1's `lgkmcnt(8)`: the four B0 reads, issued first -> B0 restaged in phase 2).
// Persistent: the staging stream runs two K steps ahead of the MFMAs across tile boundaries; the epilogue (no LDS, no barrier) of one
// pixel half overlaps the other half's last MFMAs.
#include "common.h"
#include "launchers.h"
#include <cstdlib>
#include <type_traits>

namespace {
typedef __attribute__((address_space(3))) void p8_lds_void;
constexpr unsigned P8_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int p8_u32x4;
typedef __attribute__((ext_vector_type(4))) float p8_f32x4;
typedef __attribute__((ext_vector_type(2))) float p8_f32x2;

// (-DP8_A_NT=1 / -DP8_B_NT=1, A/B builds only - CSS_HIP_LIB: the pixel / weight operand fetched with the non-temporal policy, aux = 2)
#ifndef P8_A_NT
#define P8_A_NT 0
#endif
#ifndef P8_B_NT
#define P8_B_NT 0
#endif
#ifndef P8_STORE_AUX
#define P8_STORE_AUX 0    // cache policy of the output stores (A/B builds: 2 = nt)
#endif
#ifndef P8_ADD_AUX
#define P8_ADD_AUX 0      // cache policy of the residual-gradient addend loads (2 = nt: read exactly once)
#endif
__device__ __forceinline__ void p8_dma16_nt(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (p8_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 2);
}
__device__ __forceinline__ void p8_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (p8_lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
}
__device__ __forceinline__ float p8_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ void p8_swap16(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned p8_pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
__device__ __forceinline__ float p8_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float p8_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

// Diagnostic build only (-DP8_STAMP, scripts/p8_bench.hip): s_memtime stamps at the segment boundaries of the first two tiles, written at
// the very end through a.bias (unused by this kernel) - [workgroup][wave group][8] 64-bit ticks.
#ifdef P8_STAMP
#define P8_STAMP_AT(i)                                                                                  \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    unsigned long long t_;                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if ((i) < 8) stamps[(i)] = t_;                                                                      \
  } while (0)
#else
#define P8_STAMP_AT(i)
#endif

template <bool STATS, bool ADD>
__global__ __launch_bounds__(512) void conv_igemm_p8_kernel(const ConvArgs a) {
#ifdef P8_STAMP
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  P8_STAMP_AT(0);
#endif
  constexpr int BM = 256, BN = 256, BK = 64;
  // Kernel arguments used more than once: ONE batch of scalar loads into locals the compiler cannot re-load (opaque "+s" copies).  With
  // ~100 live SGPRs the tile / row set-up code otherwise fetches each ConvArgs field again at every use - 157 s_load + s_waitcnt pairs in
  // the first build, ~10,000 cycles of prologue per launch (profiles/r03_p8_phase_stamps.txt) - where a spilled SGPR costs one v_readlane.
  int a_R = a.R, a_M = a.M, a_Cd = a.Cd, a_S = a.S, a_pad = a.pad, a_dil = a.dil, a_korder = a.korder, a_mode = a.mode, a_Wd = a.Wd, a_Hs = a.Hs, a_stride = a.stride, a_m_begin = a.m_begin, a_Ws = a.Ws, a_Hd = a.Hd, a_Cs = a.Cs, a_lds = a.lds, a_ldd = a.ldd, a_Ktot = a.Ktot, a_stat_Mg = a.stat_Mg, a_ld_add = a.ld_add;
  FastDiv a_fd_hw = a.fd_hw, a_fd_w = a.fd_w;
  asm volatile("" : "+s"(a_R), "+s"(a_M), "+s"(a_Cd), "+s"(a_S), "+s"(a_pad), "+s"(a_dil), "+s"(a_korder), "+s"(a_mode), "+s"(a_Wd), "+s"(a_Hs));
  asm volatile("" : "+s"(a_stride), "+s"(a_m_begin), "+s"(a_Ws), "+s"(a_Hd), "+s"(a_Cs), "+s"(a_lds), "+s"(a_ldd), "+s"(a_Ktot), "+s"(a_stat_Mg), "+s"(a_ld_add));
  asm volatile("" : "+s"(a_fd_hw.mul), "+s"(a_fd_hw.shr), "+s"(a_fd_hw.d), "+s"(a_fd_w.mul), "+s"(a_fd_w.shr), "+s"(a_fd_w.d));
  constexpr int HALF = 128 * 128;                                // one half-tile: 128 rows of 128 bytes
  constexpr int NEPI = 16 + (STATS ? 8 : 0);                     // store instructions of an epilogue (the addend loads of ADD are consumed inside it)
#ifdef P8_PSTAMPS
  constexpr bool MSTAT = false;                                  // (the diagnostic build keeps its stamp words where the statistics slices would sit)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * HALF + 8 * 256];     // + 64 stamp words per wave (diagnostic build)
#else
  // Round 5: the statistics of a whole slab on the MATRIX pipe (conv_ws.hip's form; profiles/r05_p8_mstat.txt): the 32 KiB of LDS beside the ring hold one
  // 32-pixel x 64-channel slice of the packed tile per wave, written as the store loop produces it and read back transposed (ds_read_b64_tr_b16: the A - and
  // B - operand of v_mfma_f32_16x16x32_bf16 with K = pixels); per 16 channels F x F (Gram matrix: the diagonal is the sum of squares) and F x ones (the
  // sums), four slices per tile: 16 ds_write_b128 + 32 transposed reads + 32 MFMAs instead of ~450 VALU instructions per wave.  Same quantity in another
  // fixed summation order (slabs equal to 3e-7 of their magnitude, outputs untouched); a slab that straddles a statistics-group boundary keeps the
  // VALU path.  P8_NO_MFMA_STATS (scripts/p8_bench.hip): every slab on the VALU path.
#ifdef P8_NO_MFMA_STATS
  constexpr bool MSTAT = false;
#else
  constexpr bool MSTAT = STATS;
#endif
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * HALF + (MSTAT ? 8 * 4096 : 0)];     // [buffer 0 / 1][A0, A1, B0, B1]
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                       // pixel half (= phase group), channel quarter
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- tile schedule (as conv_pp.hip / scripts/proto/conv_pp64.hip) ----
  const int G = gridDim.x, q8 = G >> 3, r8 = G & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int nt_n = (a_Cd + BN - 1) / BN;
  const int tiles = ((a_M - a_m_begin + BM - 1) / BM) * nt_n;
  const int nmy = pos < tiles ? (tiles - pos + G - 1) / G : 0;
  const int ncs = (a_Cs + BK - 1) / BK;                         // 64-channel slices (the last one may be ragged: Cs = 304)
  const int hw = a_Hd * a_Wd;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));       // (opaque copies: see scripts/proto/conv_pp64.hip)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  struct Tile { int m0, n0; unsigned trm; int nk; };
  auto tile_info = [&](int ti) {
    Tile t;
    const int lt = ti * G + pos;
    const int mt = nt_n == 1 ? lt : lt / nt_n;
    t.m0 = a_m_begin + mt * BM;
    t.n0 = (lt - mt * nt_n) * BN;
    t.trm = (1u << a_R) - 1;
    int nvr = a_R;
    if (a_R > 1) {
      const int mlast = min(t.m0 + BM, a_M) - 1;
      const int i0 = (int)fdiv((uint32_t)t.m0, a_fd_hw), i1 = (int)fdiv((uint32_t)mlast, a_fd_hw);
      const int h0 = (int)fdiv((uint32_t)(t.m0 - i0 * hw), a_fd_w), h1 = (int)fdiv((uint32_t)(mlast - i1 * hw), a_fd_w);
      if (i1 - i0 <= 1) {
        const int alo = h0, ahi = i1 == i0 ? h1 : a_Hd - 1, blo = i1 == i0 ? h0 : 0, bhi = h1;
        unsigned msk = 0;
        int cnt = 0;
#pragma unroll 1
        for (int r = 0; r < a_R; ++r) {
          const int o = a_mode == 0 ? r * a_dil - a_pad : a_pad - r * a_dil;     // source row = output row + o (stride 1 whenever R > 1)
          const bool v = (alo + o <= a_Hs - 1 && ahi + o >= 0) || (blo + o <= a_Hs - 1 && bhi + o >= 0);
          if (v) { msk |= 1u << r; ++cnt; }
        }
        if (cnt > 0) { t.trm = msk; nvr = cnt; }
      }
    }
    t.nk = ncs * nvr * a_S;
    return t;
  };

  // ---- staging side -----------------------------------------------------------------------------------------------------------
  // A half-tile is 16 pieces of 1 KiB (8 rows x 128 B); thread -> piece i * 8 + wave (i = 0, 1) of every half-tile, i.e. local rows
  // 64 i + 8 wave + (lane >> 3), 16-byte position lane & 7.  The chunk stored at position p of local row r is source chunk
  // p ^ ((r >> 1) & 7) (conflict-free ds_read_b128 of 16 consecutive rows, as scripts/proto/conv_pp64.hip); (r >> 1) & 7 = 4 (wave & 1) + (lane >> 4)
  // for all four of my rows, so ONE source chunk per thread.
  //   A half h, local row r -> pixel row (r >> 6) * 128 + h * 64 + (r & 63) of the tile (piece i serves pixel half i)
  //   B half h, local row r -> channel (r >> 5) * 64 + h * 32 + (r & 31) of the tile
  const int cch = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const int tabv_da = a.tab_da[lane < 63 ? lane : 62], tabv_kb = a.tab_kb[lane < 63 ? lane : 62], tabv_tap = a.tab_tap[lane < 63 ? lane : 62];
  // Position of the staging stream: tile ti, inner / outer counter of the K order (korder 1: tap inner, channel slice outer; 0: the
  // reverse), with their limits; live = the tile exists (else ghost steps: zero fills), need = the next step is the first of tile ti.
  struct KPos { int ti, pin, pout, lin, lout, vb; int live, need; };
  KPos ps = {0, 0, 0, 1, 1, 0, 0, 1};
  // row q = 2 h + i (h: half-tile, i: piece)
  int rowoff[4], nrowoff[4];      // byte offset of the tap-(0,0) source pixel of my rows (+ my chunk), may be out of range: see rmask
  unsigned rmask[4], nrmask[4];   // bit (tr*S + ts): that tap of the row lies inside the source image
  unsigned boff[4], nboff[4];     // byte offset of my weight rows (+ my chunk), or OOB
  auto lane_setup = [&](int ti, int (&ro)[4], unsigned (&rm)[4], unsigned (&bo)[4]) {
    if (ti >= nmy) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { rm[q] = 0; ro[q] = 0; bo[q] = P8_OOB; }
      return;
    }
    const int lt = ti * G + pos;
    const int mt = nt_n == 1 ? lt : lt / nt_n;
    const int m0 = a_m_begin + mt * BM, n0 = (lt - mt * nt_n) * BN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = q >> 1, i = q & 1;
      const int m = m0 + i * 128 + h * 64 + wave * 8 + (lane >> 3);
      unsigned msk = 0;
      int off = 0;
      if (m < a_M) {
        const uint32_t n_img = fdiv((uint32_t)m, a_fd_hw);
        const uint32_t rem = (uint32_t)m - n_img * (uint32_t)hw;
        const int hd = (int)fdiv(rem, a_fd_w);
        const int wd = (int)rem - hd * a_Wd;
        int h0, w0;      // source coordinate of tap (0,0)
        bool ok0 = true;
        if (a_mode == 0) {
          h0 = hd * a_stride - a_pad;
          w0 = wd * a_stride - a_pad;
        } else {
          h0 = hd + a_pad;
          w0 = wd + a_pad;
          if (a_stride == 2) {          // (1x1 only, checked by the launcher): the pixel has a source only at even coordinates
            ok0 = !((h0 | w0) & 1);
            h0 >>= 1;
            w0 >>= 1;
          }
        }
        off = (((int)n_img * a_Hs + h0) * a_Ws + w0) * a_lds * 2 + cch * 16;
        const int sgn = a_mode == 0 ? a_dil : -a_dil;
        unsigned bit = 1;
#pragma unroll 1
        for (int r = 0; r < a_R; ++r)
#pragma unroll 1
          for (int s = 0; s < a_S; ++s, bit <<= 1) {
            const int hs = h0 + sgn * r, ws = w0 + sgn * s;
            if (ok0 && (unsigned)hs < (unsigned)a_Hs && (unsigned)ws < (unsigned)a_Ws) msk |= bit;
          }
      }
      ro[q] = off;
      rm[q] = msk;
      const int n = n0 + (i * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + (lane >> 3);
      bo[q] = n < a_Cd ? (unsigned)n * (unsigned)a_Ktot * 2u + (unsigned)cch * 16u : P8_OOB;
    }
  };
  // scalars of the K step whose offsets are being computed (set in phase 4; the offsets of that step's four half-tiles are computed from
  // them in the MFMA segments of the next K step's phases 1-3)
  int tap_cur = 0, da_cur = 0, lim_cur = 0;
  unsigned kb_cur = 0;
  // The stream enters tile ps.ti (rare: once per tile, so it may branch; it sits in phase 4's load segment).
  auto step_switch = [&]() {
    if (ps.need) {
      ps.need = 0;
      ps.live = ps.ti < nmy;
      if (ps.live) {
        const Tile t = tile_info(ps.ti);
        const int nt = __builtin_popcount(t.trm) * a_S;
        ps.pin = 0;
        ps.pout = 0;
        ps.lin = a_korder ? nt : ncs;
        ps.lout = a_korder ? ncs : nt;
        ps.vb = ((int)t.trm - 1) * 9;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { rowoff[q] = nrowoff[q]; rmask[q] = nrmask[q]; boff[q] = nboff[q]; }
    }
  };
  // Scalars of the next K step + advance: BRANCH-FREE (selects only), so that it can sit inside phase 4's MFMA segment, spread between
  // the MFMAs by the scheduler.  (The first version - if / else chains, ~60 SALU with five taken branches in phase 4's load segment -
  // made phase 4 the longest phase of the K step: profiles/r03_p8_phase_stamps.txt.)
  auto step_scalars = [&]() {
    const int it = a_korder ? ps.pin : ps.pout, cs = a_korder ? ps.pout : ps.pin;
    const int ix = ps.vb + it;
    tap_cur = __builtin_amdgcn_readlane(tabv_tap, ix);
    da_cur = __builtin_amdgcn_readlane(tabv_da, ix) + cs * (BK * 2);
    kb_cur = (unsigned)(__builtin_amdgcn_readlane(tabv_kb, ix) + cs * (BK * 2));
    lim_cur = ps.live ? a_Cs - cs * BK : 0;                    // channels left in this slice (ragged last slice); 0: ghost step
    const int pin1 = ps.pin + 1;
    const int wrap = pin1 == ps.lin;
    const int pout1 = ps.pout + wrap;
    const int done = wrap & (pout1 == ps.lout) & ps.live;
    ps.pin = wrap ? 0 : pin1;
    ps.pout = pout1;
    ps.ti += done;
    ps.need |= done;
  };
  auto step_begin = [&]() {
    step_switch();
    step_scalars();
  };
  unsigned char* const st_base = smem + wave * 1024;
  // The per-piece offsets are computed one phase AHEAD, inside the MFMA segment of the wave that will issue them (voff_a / voff_b
  // below), so that a load segment holds no vector ALU work at all: while one wave of a SIMD multiplies under s_setprio 1 its partner's
  // VALU instructions wait for issue slots (MI355X_MICROARCH.md, Two waves per SIMD, item 2: ~20 cycles each), and a load segment
  // with ten of them outlasts the 256 cycles of the partner's 16 MFMAs.  Beside a wave's OWN MFMAs a VALU instruction per MFMA is
  // nearly free (an MFMA holds the vector issue port for 8 of its 16 cycles).
  auto voff_a = [&](int h, unsigned (&v)[2]) {
    const unsigned tb = cch * 8 < lim_cur ? 1u << tap_cur : 0u;            // my lane's tap bit, or none past the ragged end of the slice
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = h * 2 + i;
      v[i] = (rmask[q] & tb) ? (unsigned)(rowoff[q] + da_cur) : P8_OOB;
      asm volatile("" : "+v"(v[i]));      // (pinned here: otherwise the optimizer sinks the arithmetic into the load segment of the phase that uses it)
    }
  };
  auto voff_b = [&](int h, unsigned (&v)[2]) {
    const bool okc = cch * 8 < lim_cur;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v[i] = okc ? boff[h * 2 + i] + kb_cur : P8_OOB;          // (an OOB row stays out of range with kb added)
      asm volatile("" : "+v"(v[i]));
    }
  };
  auto stage = [&](__amdgpu_buffer_rsrc_t rs, int par, int slot, const unsigned (&v)[2]) {      // slot: 0 = A0, 1 = A1, 2 = B0, 3 = B1
    unsigned char* const d = st_base + (par * 4 + slot) * HALF;
    if ((P8_A_NT && slot < 2) || (P8_B_NT && slot >= 2)) {
      p8_dma16_nt(rs, d, v[0]);
      p8_dma16_nt(rs, d + 8192, v[1]);
    } else {
      p8_dma16(rs, d, v[0]);
      p8_dma16(rs, d + 8192, v[1]);
    }
  };

  // ---- consumer state ----
  f32x4 acc[8][4];        // [pixel tile i: pixels 16 i + (lane & 15) of my half][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int sw = (l15 >> 1) & 7;
  const int ko0 = ((lg ^ sw) << 4), ko1 = (((4 + lg) ^ sw) << 4);           // k halves 0 / 1 of a 128-byte row
  const unsigned char* const a_rd = smem + (wm * 64 + l15) * 128;
  const unsigned char* const b_rd = smem + 2 * HALF + (wn * 32 + l15) * 128;

  // prologue: K step 0 whole -> buffer 0, K step 1 up to B1 -> buffer 1 (its A1 goes out in phase 1 of step 0), scalars of step 2
  unsigned vb0[2], va0[2], vb1[2], va1[2];
  lane_setup(0, nrowoff, nrmask, nboff);
  step_begin();
  voff_b(0, vb0); voff_a(0, va0); voff_b(1, vb1); voff_a(1, va1);
  stage(rs_b, 0, 2, vb0); stage(rs_a, 0, 0, va0); stage(rs_b, 0, 3, vb1); stage(rs_a, 0, 1, va1);
  lane_setup(1, nrowoff, nrmask, nboff);
  step_begin();
  voff_b(0, vb0); voff_a(0, va0); voff_b(1, vb1); voff_a(1, va1);
  stage(rs_b, 1, 2, vb0); stage(rs_a, 1, 0, va0); stage(rs_b, 1, 3, vb1);
  step_begin();
  P8_STAMP_AT(1);                                      // everything of the prologue issued
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // K step 0 has landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();           // the second pixel half runs one barrier behind
  asm volatile("" ::: "memory");
  P8_STAMP_AT(2);                                      // first data landed: the K loop starts

// s_setprio around the MFMA segments: measured 1-2 % SLOWER here (profiles/r03_p8_final_harness.txt; the template needs it to keep
// hipcc from moving MFMAs across its barriers - here sched_barriers pin the segments), so it is off unless -DP8_SETPRIO.
#ifdef P8_SETPRIO
#define P8_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define P8_PRIO(x)
#endif
// interleave hint for the scheduler: after every MFMA of the segment up to two VALU / SALU instructions of the offset arithmetic
#define P8_MIX()                                                                  \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);                            \
  }
// -DP8_PSTAMPS=1: a stamp at every phase boundary of ONE K step (tile 0, K step 8); =2: also before / after each MFMA segment.
// Lane 0 writes the low word into the wave's LDS stamp area (no LDS read is in flight at these points); copied out at the end.
#ifdef P8_PSTAMPS
#define P8_PS(idx, lvl)                                                                                                     \
  if ((lvl) <= P8_PSTAMPS && pstamp_on) {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
    unsigned long long t_;                                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                              \
    if (lane == 0) *(volatile unsigned*)(smem + 8 * HALF + wave * 256 + 4 * (idx)) = (unsigned)t_;                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                      \
  }
#else
#define P8_PS(idx, lvl)
#endif
#define P8_MID()                                        \
  __builtin_amdgcn_sched_barrier(0);                    \
  __builtin_amdgcn_s_barrier();                         \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
  __builtin_amdgcn_sched_barrier(0);                    \
  P8_PRIO(1);
#define P8_END()                         \
  P8_PRIO(0);                            \
  __builtin_amdgcn_sched_barrier(0);     \
  __builtin_amdgcn_s_barrier();          \
  asm volatile("" ::: "memory");         \
  __builtin_amdgcn_sched_barrier(0);
  // one 64 x 32 quadrant x K = 64: rows AH * 64 + 16 i, channels BH * 32 + 16 j.  CODE = the offset arithmetic for a later phase:
  // it sits in the same basic block as the MFMAs (there is no first-K-step branch: the accumulators are zeroed by the epilogue) so
  // that the scheduler spreads it between them, one VALU instruction per MFMA (P8_MIX).
#define P8_QUAD(AH, FB, BH, CODE)                                                                                                      \
  CODE;                                                                                                                                \
  _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[(AH) * 4 + i][(BH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j][kh], fa[i][kh], acc[(AH) * 4 + i][(BH) * 2 + j], 0, 0, 0); \
  P8_MIX();

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int par = 0;                       // buffer of the K step being multiplied
  // LDS addresses of my fragment rows in buffer par, k halves 0 / 1 (recomputed for the next K step inside phase 4's MFMA segment)
  const unsigned ra_base = (unsigned)(uintptr_t)(p8_lds_void*)a_rd, rb_base = (unsigned)(uintptr_t)(p8_lds_void*)b_rd;
  unsigned ra0 = ra_base + ko0, ra1 = ra_base + ko1, rb0 = rb_base + ko0, rb1 = rb_base + ko1;
  typedef __attribute__((address_space(3))) const bf16x8 p8_lds_frag;
  auto frag = [&](unsigned base, int off) { return *(p8_lds_frag*)(uintptr_t)(base + (unsigned)off); };
  for (int ti = 0; ti < nmy; ++ti) {
    const Tile ct = tile_info(ti);
    for (int kt = 0; kt < ct.nk; ++kt) {
      const bool after_epi = kt == 0 && ti > 0;
      bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#ifdef P8_PSTAMPS
      const bool pstamp_on = ti == 0 && (kt == 8 || kt == 9);
      const int pbase = (kt - 8) * 16;
#endif
      P8_PS(pbase + 0, 1);
      // ---- phase 1: A0 x B0 ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb0[j][0] = frag(rb0, j * 2048);
        fb0[j][1] = frag(rb1, j * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i][0] = frag(ra0, i * 2048);
        fa[i][1] = frag(ra1, i * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!after_epi) stage(rs_a, par ^ 1, 1, va1);             // A1 of step t+1 (a tile's first K step: already issued ahead of the epilogue's stores)
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");        // my four B0 reads are done: B0 of this buffer is restaged next phase
      P8_MID();
      P8_PS(pbase + 1, 2);
      P8_QUAD(0, fb0, 0, voff_b(0, vb0));                       // (+ offsets for phase 2; scalars of step t+2: set in phase 4 of the previous K step)
      P8_PS(pbase + 2, 2);
      P8_END();
      P8_PS(pbase + 3, 1);
      // ---- phase 2: A0 x B1 ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb1[j][0] = frag(rb0, HALF + j * 2048);
        fb1[j][1] = frag(rb1, HALF + j * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage(rs_b, par, 2, vb0);                                 // B0 of step t+2
      P8_MID();
      P8_PS(pbase + 4, 2);
      P8_QUAD(0, fb1, 1, voff_a(0, va0));
      P8_PS(pbase + 5, 2);
      P8_END();
      P8_PS(pbase + 6, 1);
      // ---- phase 3: A1 x B1 ----
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i][0] = frag(ra0, HALF + i * 2048);
        fa[i][1] = frag(ra1, HALF + i * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage(rs_a, par, 0, va0);                                 // A0 of step t+2
      P8_MID();
      P8_PS(pbase + 7, 2);
      P8_QUAD(1, fb1, 1, (voff_b(1, vb1), voff_a(1, va1)));
      P8_PS(pbase + 8, 2);
      P8_END();
      P8_PS(pbase + 9, 1);
      // ---- phase 4: A1 x B0 (no fragment reads: the scalar bookkeeping of the stream sits here) ----
      stage(rs_b, par, 3, vb1);                                 // B1 of step t+2
      step_switch();                                            // the stream enters a new tile (once per tile)
      // everything up to A1 of step t+1 has landed: step t+1 is whole and is read from the next phase on.  vmcnt is in order and counts
      // stores: in a tile's first K step the NEPI stores of the previous epilogue are younger than that A1 (issued ahead of them), so the
      // wait can leave them outstanding - they get a second K step to drain before the next wait, which does include them (every CU ends
      // its tile at the same time: 256 x 128 KiB of stores at once, ~7 us: profiles/r03_store_stall.txt)
      if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NEPI) : "memory");      // (leaves the epilogue's stores in flight)
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      P8_MID();
      P8_PS(pbase + 10, 2);
      par ^= 1;
      const unsigned po = (unsigned)par * (4 * HALF);
      P8_QUAD(1, fb0, 0, (ra0 = ra_base + ko0 + po, ra1 = ra_base + ko1 + po, rb0 = rb_base + ko0 + po, rb1 = rb_base + ko1 + po,
                          step_scalars()));                     // + scalars of step t+3
      P8_PS(pbase + 11, 2);
      P8_END();
      P8_PS(pbase + 12, 1);
    }

    if (ti < 2) P8_STAMP_AT(3 + 2 * ti);                 // K loop of tile ti done
    // A1 of the next K step goes out HERE, ahead of the stores (it is phase 1's piece of that step: see there and phase 4's wait)
    stage(rs_a, par ^ 1, 1, va1);
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- epilogue of tile ti (as scripts/proto/conv_pp64.hip: no LDS, no barrier) ----------------
    const int mrow0 = ct.m0 + wm * 128, n0w = ct.n0 + wn * 64;
    const int bnd = STATS ? (mrow0 / a_stat_Mg + 1) * a_stat_Mg : 0x7fffffff;     // rows >= bnd: next statistics group (stage 2 sums them)
    const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);
    // last tile of this workgroup: the ghost LDS-DMAs (issued by the last two K steps) must have landed before the workgroup's LDS is
    // released - wait for them HERE, ahead of the stores, so that the wave can end with its stores in flight.  (A `vmcnt(0)` behind
    // the stores holds the CU until the whole chip's output burst - 256 tiles x 128 KiB at once - has drained: 5-20 % of a launch
    // in profiles/r03_store_stall.txt.)
    if (ti == nmy - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- sums / sums of squares of a WHOLE slab on the matrix pipe (see MSTAT above) ----
    const bool whole_m = MSTAT && mrow0 + 128 <= bnd;
    f32x4 dsum[MSTAT ? 4 : 1], dsq[MSTAT ? 4 : 1];
    unsigned mwb = 0, mtb = 0;
    if constexpr (MSTAT) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) dsum[cb] = dsq[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned sbase = (unsigned)(uintptr_t)(p8_lds_void*)(smem + 8 * HALF) + (unsigned)wave * 4096u;
      // write side: row 16 (i & 1) + l15, chunk 2 (lg & 1) + (lg >> 1) [+ 4 for the second channel pair] at position chunk ^ ((row >> 1) & 7)
      mwb = sbase + (unsigned)(l15 * 128 + (((2 * (lg & 1) + (lg >> 1)) ^ ((l15 >> 1) & 7)) << 4));
      // read side (ds_read_b64_tr_b16): lane 4 q + p of lane group lg -> row 8 lg + q, channels 4 p .. of the 16-channel block
      const int q4 = l15 >> 2, p4 = l15 & 3, R = 8 * lg + q4;
      mtb = sbase + (unsigned)(R * 128 + ((((p4 >> 1) ^ ((R >> 1) & 7))) << 4) + 8 * (p4 & 1));
      asm volatile("" : "+v"(mwb), "+v"(mtb));
    }
    p8_u32x4 radd[ADD ? 8 : 1][2];
    unsigned rmk[ADD ? 8 : 1][2];
    const bool has_mask = ADD && a.add_mask != nullptr;
    if (ADD) {
      const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.addend), 0, (int)a.add_bytes, 0x00020000);
      // optional ReLU bit mask of the addend (css_conv2d_dgrad_add_masked): one byte per 16-byte vector
      const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.add_mask), 0, (int)a.mask_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mrow0 + 16 * i + l15, n = nl + 32 * h;
          radd[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)((m < a_M && n < a_Cd) ? ((unsigned)m * (unsigned)a_ld_add + (unsigned)n) * 2u : P8_OOB), 0, P8_ADD_AUX);
          if (has_mask) rmk[i][h] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_k, (int)((m < a_M && n < a_Cd) ? (unsigned)m * ((unsigned)a_Cd >> 3) + ((unsigned)n >> 3) : P8_OOB), 0, 0);
          else rmk[i][h] = 0xFFu;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mrow0 + 16 * i + l15;
      unsigned lo[4], hi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = p8_pack2(acc[i][j][0], acc[i][j][1]);
        hi[j] = p8_pack2(acc[i][j][2], acc[i][j][3]);
      }
      const unsigned rowb = (unsigned)m * (unsigned)a_ldd * 2u;
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        p8_swap16(lo[jp], lo[jp + 1]);
        p8_swap16(hi[jp], hi[jp + 1]);
        p8_u32x4 v = {lo[jp], hi[jp], lo[jp + 1], hi[jp + 1]};
        const int n = nl + 16 * jp;
        const bool ok = m < a_M && n < a_Cd;
        if (ADD) {
          p8_u32x4 r = radd[ADD ? i : 0][jp >> 1];
          const unsigned mk = rmk[ADD ? i : 0][jp >> 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] &= keep_mask_bf16x2(mk, e);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p8_pack2(p8_lo(v[e]) + p8_lo(r[e]), p8_hi(v[e]) + p8_hi(r[e]));
        }
#if defined(P8_ABL_NOSTORE)       // timing ablation (scripts/p8_bench.hip): keep the values alive, drop the stores
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(ok ? rowb + (unsigned)n * 2u : P8_OOB), 0, P8_STORE_AUX);
#endif
        if constexpr (MSTAT) {
          if (whole_m) {
            // (row 16 (i & 1) + l15: (row >> 1) & 7 = 8 (i & 1) | (l15 >> 1) -> & 7 = (l15 >> 1) & 7: the lane part covers both i; second pair: chunk + 4 = bit 6)
            if (i & 1) asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"((mwb ^ (unsigned)(jp << 5))), "v"(v) : "memory");
            else asm volatile("ds_write_b128 %0, %1" ::"v"((mwb ^ (unsigned)(jp << 5))), "v"(v) : "memory");
          }
        }
      }
      if constexpr (MSTAT) {
        if (whole_m && (i & 1)) {
          // the slice (pixel tiles i - 1, i: 32 pixels x 64 channels) is in LDS: per 16-channel block one transposed fragment, Gram + ones
          const bf16x8 ones = __builtin_bit_cast(bf16x8, p8_u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
          s16x4 fa4[4], fb4[4];
          const unsigned t0 = mtb, t1 = mtb ^ 32u, t2 = mtb ^ 64u, t3 = mtb ^ 96u;
          asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:512\n\t"
                       "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %8 offset:512\n\t"
                       "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %11 offset:512\n\t"
                       "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %10 offset:512\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(fa4[0]), "=&v"(fb4[0]), "=&v"(fa4[1]), "=&v"(fb4[1]), "=&v"(fa4[2]), "=&v"(fb4[2]), "=&v"(fa4[3]), "=&v"(fb4[3])
                       : "v"(t0), "v"(t1), "v"(t2), "v"(t3)
                       : "memory");
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            union { struct { s16x4 a, b; } h; bf16x8 f; } u;
            u.h.a = fa4[cb];
            u.h.b = fb4[cb];
            dsq[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f, u.f, dsq[cb], 0, 0, 0);
            dsum[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f, ones, dsum[cb], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (STATS) {
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(mrow0 >> 7) * 2u * (unsigned)a_Cd * 4u;
      // sum and sum of squares of the bf16-ROUNDED outputs (what the batch norm will read), two values per instruction (v_pk_add_f32 /
      // v_pk_fma_f32); the row test only in the one slab per statistics group that straddles the group boundary (as conv_ws.hip)
      const bool whole = mrow0 + 128 <= bnd;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p8_f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        auto accum = [&](bool test) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            f32x4 t = acc[i][j];
            asm volatile("" : "+v"(t));       // opaque: otherwise the packed values of the store loop stay alive (CSE) across the epilogue
            const unsigned lo = p8_pack2(t[0], t[1]), hi = p8_pack2(t[2], t[3]);
            p8_f32x2 v01 = {p8_lo(lo), p8_hi(lo)}, v23 = {p8_lo(hi), p8_hi(hi)};
            if (test && !(mrow0 + 16 * i + l15 < bnd)) { v01 = p8_f32x2{0.f, 0.f}; v23 = p8_f32x2{0.f, 0.f}; }   // (rows >= M hold zeros already)
            s01 += v01; s23 += v23;
            q01 += v01 * v01; q23 += v23 * v23;
          }
        };
        p8_f32x4 os, oq;
        if (MSTAT && whole_m) {
          const int p4 = l15 & 3;
          const float dg = p4 == 0 ? dsq[MSTAT ? j : 0][0] : p4 == 1 ? dsq[MSTAT ? j : 0][1] : p4 == 2 ? dsq[MSTAT ? j : 0][2] : dsq[MSTAT ? j : 0][3];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            os[r] = dsum[MSTAT ? j : 0][r];
            oq[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((20 * lg + r) * 4, __builtin_bit_cast(int, dg)));
          }
        } else {
          if (whole) accum(false);
          else accum(true);
          const float ss[4] = {s01[0], s01[1], s23[0], s23[1]}, qq[4] = {q01[0], q01[1], q23[0], q23[1]};
#pragma unroll
          for (int r = 0; r < 4; ++r) { os[r] = p8_row16_sum(ss[r]); oq[r] = p8_row16_sum(qq[r]); }
        }
        const int n = n0w + 16 * j + 4 * lg;
        const bool lane_ok = l15 == 0 && n < a_Cd && mrow0 < a_M;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p8_u32x4, os), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : P8_OOB), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p8_u32x4, oq), rs_s, (int)(lane_ok ? base + (unsigned)(a_Cd + n) * 4u : P8_OOB), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile accumulates from zero (no first-K-step form of the MFMA segments: they stay one basic block)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    // the staging stream is inside tile ti+1 by now (every tile has at least three K steps): prepare tile ti+2 for it
    lane_setup(ti + 2, nrowoff, nrmask, nboff);
    __builtin_amdgcn_sched_barrier(0);
    if (ti < 2) P8_STAMP_AT(4 + 2 * ti);                 // epilogue (stores issued, accumulators zeroed, next-next tile set up)
  }
#ifdef P8_PSTAMPS
  if (a.bias && lane < 32 && (wave & 3) == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ((unsigned*)a.bias)[((size_t)blockIdx.x * 2 + wm) * 32 + lane] = *(volatile unsigned*)(smem + 8 * HALF + wave * 256 + 4 * lane);
  }
#endif
#ifdef P8_STAMP
  P8_STAMP_AT(7);
  if (a.bias && lane == 0 && (wave & 3) == 0) {
    unsigned long long* o = (unsigned long long*)a.bias + ((size_t)blockIdx.x * 2 + wm) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = stamps[i];
  }
#endif
  if (wm == 0) __builtin_amdgcn_s_barrier();           // the barrier the other half ran at the start
  if (nmy == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (a workgroup without a tile: only its prologue's ghost DMAs)
}

// Supported: what conv_pp.hip supports, with at most 3 x 3 taps (tap tables: 7 sets of valid kernel rows x 9 taps) and at least three
// 64-channel K steps per valid kernel row.  CSS_NO_P8_CONV=1: every such shape back on conv_igemm_pp_kernel (A/B reference; the
// intermediate generation conv_igemm_pp64_kernel is archived in scripts/proto/scripts/proto/conv_pp64.hip).
bool css_conv_p8_supported(const ConvArgs& a) {
  static const bool off = getenv("CSS_NO_P8_CONV") != nullptr;
  if (off || !css_conv_pp_supported(a) || a.R > 3 || a.S > 3) return false;
  const int ncs = (a.Cs + 63) / 64;
  return ncs * a.S >= 3;
}

void css_launch_conv_p8(ConvArgs a, int grid, hipStream_t st) {
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
  if (a.stats) hipLaunchKernelGGL((conv_igemm_p8_kernel<true, false>), g, b, 0, st, a);
  else if (a.addend) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_p8_kernel<false, false>), g, b, 0, st, a);
}
