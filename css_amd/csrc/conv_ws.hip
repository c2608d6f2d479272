// conv_ws_kernel: weight-STATIONARY 1x1 convolution for the short-K class (K = Cin in {64, 128, 256, 512}, Cout a multiple of 256):
// conv3 of a Bottleneck (generalframeworks/networks/resnet.py:131-133: planes -> 4 x planes) in the forward pass and conv1 of a
// Bottleneck (resnet.py:123-125: 4 x planes -> planes) in its data-gradient form, 23 + 22 of each in layer 3 alone.
//
// Why a kernel of its own (profiles/r02_conv_ablation.txt section 4): on these shapes the 256x256 persistent
// kernels are bound by the global -> LDS fill path, not by MFMA, HBM or the stores: a 256x256 tile with K = 256 re-stages 128 KiB of
// weights AND 128 KiB of pixels for 33.5 MFLOP (10 us per tile against 3.9 us of MFMA).  Here
//   * a workgroup owns ONE 256-channel panel of the output for its whole life and keeps the panel's weights in REGISTERS as MFMA
//     operands: wave w holds channels 32 w .. 32 w + 31 for all of K (K = 256: 16 fragments = 64 VGPRs), loaded once per launch;
//   * the only LDS traffic from global memory is the pixel stream: tiles of 128 pixels, one 16 KiB stage per 64 channels, a ring of
//     LA + 2 stages (K <= 128: LA = two tiles; K = 256: one tile = 4 stages; K = 512: half a tile = 4 stages - 64 KiB in flight per CU either
//     way, which measures the same as 128) - half the bytes per FLOP of the 256x256 tiling, and none of them weights;
//   * every wave reads the whole pixel stage (128 x 64) and multiplies it with its own 32 channels: 16 ds_read_b128 + 32
//     v_mfma_f32_16x16x32_bf16 per stage and wave, ONE barrier per stage;
//   * the 4 (Cout / 256) panels of a pixel tile are worked on at the same time by 4 CUs of one XCD (the tile is fetched from HBM once
//     and hits that XCD's L2 three times); 128-row tiles quantise to the chip well enough (1057 tiles x 4 panels on 256 CUs = 16.5
//     rounds) that there is no leftover launch;
//   * the epilogue (round 5: the ROW form) packs the tile to bf16 in registers (v_permlane16_swap: 8 consecutive channels per lane),
//     passes the tile through a 64 KiB LDS buffer (one barrier) and stores it two whole 512-byte panel rows per instruction; the BN statistics
//     (slabs of 128 rows, which a wave owns whole: ONE 16-lane store per wave and tile) come from the LDS tile on the MATRIX pipe - transposed
//     reads of the wave's own 128 x 32 block, Gram diagonal + product with ones - or, for the one tile per statistics group that straddles its
//     boundary, from the packing loop.  Why: a store instruction that writes 16 HALF cache lines (what a wave's 32 channels give) slows the in-order
//     vector-memory path that the LDS-DMA stream shares - the fill + store skeleton of the kernel cost 90 us where its fill alone takes 24
//     and its stores alone 44; with whole lines it is 73 (profiles/r05_ws_skeleton*.txt, r05_ws_seg.txt: 128-byte segments already give all
//     of it).  The residual-gradient addend of css_conv2d_dgrad_add is requested at the START of the tile, in the row form as well (whole
//     lines), so that its wait does not drain the LDS-DMA queue.
// vmcnt bookkeeping: loads, stores and LDS-DMA retire in order, so the wait for "my two pieces of stage s" is a COUNT of what the
// wave has issued since: 2 (LA - 1) pieces + the epilogue stores (+ addend loads) of the tiles in between - a constant when LA is a
// whole number of tiles; K = 512 (LA = half a tile, 128 VGPRs of weights, no addend form) counts by slice: the previous tile's stores
// lie between issue and wait for the first LA slices of a tile only.
// tests/test_host_cpu.py replays the issue order against these counts and pins them from the ISA.
// Compile-time switches (scripts/ws_bench.hip only; the library is built with none of them): timing ablations whose results are garbage -
// WS_ABL_NOSTORE / _NOMFMA / _NODMA / _NOSTATSTORE (they combine), _PANEL_XCD, _WRAP_SRC / _WRAP_DST (the stream from / into a cache-resident
// window), _STORE_ROWS / _STORE_SEG=128|256 (the old epilogue's bytes in whole rows / segments per store instruction), WS_NT, WS_DESYNC,
// WS_STORE_AUX - and bit-identical variants: WS_NO_ROWS (the round-4 epilogue: 64 bytes per pixel straight from registers, two tiles of
// look-ahead at K = 256), WS_NO_ROWS8 (that, for K = 512 only), WS_PP=1 (the ping-pong form of the K loop, waves 0-3 and 4-7 half a stage
// apart, K <= 256: 10 % faster without stores, not faster with them in either store form - profiles/r02_ws_kernel.txt section 4,
// r05_ws_pp_rows.txt - so not shipped).
#include "common.h"
#include "launchers.h"
#include <cstdlib>

#ifndef WS_STORE_AUX
#define WS_STORE_AUX 0      // cache policy of the output stores (timing ablation: 2 = nt)
#endif
#ifndef WS_PIX_AUX
#define WS_PIX_AUX 0        // ... of the pixel stream (LDS-DMA)
#endif
#ifndef WS_ADD_AUX
#define WS_ADD_AUX 0        // ... of the residual-gradient addend loads
#endif
namespace {
typedef __attribute__((address_space(3))) void ws_lds_void;
constexpr unsigned WS_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int ws_u32x4;
typedef __attribute__((ext_vector_type(4))) float ws_f32x4;
typedef __attribute__((ext_vector_type(2))) float ws_f32x2;

__device__ __forceinline__ void ws_dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (ws_lds_void*)lds_wave_base, 16, (int)off, 0, 0, WS_PIX_AUX);
}
__device__ __forceinline__ float ws_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// (the builtin returns one register for both results in this toolchain: DESIGN.md 3)
__device__ __forceinline__ void ws_swap16(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ unsigned ws_pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
__device__ __forceinline__ float ws_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float ws_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// vmcnt(BASE + c_st * NST + c_ld * NLD) for c_st in 0..2, c_ld in 1..2 (the counts fold to constants once the K loop is unrolled)
template <int BASE, int NST, int NLD> __device__ __forceinline__ void ws_wait_stage(int c_st, int c_ld) {
  static_assert(BASE + 2 * NST + 2 * NLD <= 63, "vmcnt is a 6-bit counter");
  if (c_ld <= 1) {
    if (c_st == 0) ws_wait_vm<BASE + NLD>();
    else if (c_st == 1) ws_wait_vm<BASE + NST + NLD>();
    else ws_wait_vm<BASE + 2 * NST + NLD>();
  } else {
    if (c_st == 0) ws_wait_vm<BASE + 2 * NLD>();
    else if (c_st == 1) ws_wait_vm<BASE + NST + 2 * NLD>();
    else ws_wait_vm<BASE + 2 * NST + 2 * NLD>();
  }
}
}  // namespace
#ifndef WS_PP
#define WS_PP 0
#endif
// Diagnostic build only (-DWS_STAMP, scripts/ws_bench.hip): s_memtime differences summed per wave over the steady-state tiles (ti >= 2) -
// stage wait and barrier of a tile's first stage (slots 0, 1) and of its other stages (5, 6) / LDS-DMA issue (2) / fragment reads + MFMAs
// (3) / epilogue (4) - and written at the end through a.bias as [workgroup][wave][8] 64-bit ticks (slot 7: number of tiles counted).
#ifdef WS_STAMP
#define WS_T(var)                                                                        \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)
#define WS_ACC(i, t1, t0) do { if (ti >= 2) wsum[i] += (t1) - (t0); } while (0)
#else
#define WS_T(var)
#define WS_ACC(i, t1, t0)
#endif

// KS = K / 64 (stages per pixel tile).  grid = n_cu workgroups of 512 threads, (n_cu / 8) % (Cd / 256) == 0.
template <int KS, bool STATS, bool ADD>
__global__ __launch_bounds__(512) void conv_ws_kernel(const ConvArgs a) {
  // Round 5 (profiles/r05_ws_skeleton.txt): the fill + store skeleton of this kernel costs MORE than its fill and its stores alone (24 + 44 us
  // against 90) when every store instruction writes 16 half cache lines (64 bytes of 16 pixel rows - what a wave's 32 channels give), and 73
  // when it writes whole 512-byte rows of the panel.  ROWS: the packed tile goes through a 64 KiB LDS buffer (written as the waves hold it,
  // read back two whole rows per instruction) - the space comes from ONE tile of look-ahead at K = 256, which measures the same as two.
#if WS_PP != 0 || defined(WS_NO_ROWS)
  constexpr bool ROWS = false;
#elif defined(WS_NO_ROWS8)
  constexpr bool ROWS = KS <= 4;
#else
  constexpr bool ROWS = true;
#endif
#ifdef WS_NT      // harness experiment: tiles of look-ahead
  constexpr int NT = WS_NT;
#else
  constexpr int NT = (ROWS ? KS <= 2 : KS <= 4) ? 2 : 1;         // whole tiles of look-ahead
#endif
  // stages of look-ahead: whole tiles, except K = 512 in the row form - HALF a tile (four stages, 64 KiB in flight as at K = 256), which
  // leaves the 64 KiB of the output tile
  constexpr int BM = 128, BN = 256, LA = (ROWS && KS == 8) ? 4 : NT * KS, NS = LA + 2;
  constexpr int STG = BM * 128;                                  // one stage: 128 pixels x 128 bytes (64 channels)
#ifdef WS_ABL_NOSTORE              // timing ablations (scripts/ws_bench.hip; they combine): results are garbage
  constexpr bool ABL_NOSTORE = true;
#else
  constexpr bool ABL_NOSTORE = false;
#endif
#ifdef WS_ABL_NOMFMA
  constexpr bool ABL_NOMFMA = true;
#else
  constexpr bool ABL_NOMFMA = false;
#endif
#ifdef WS_ABL_NODMA
  constexpr bool ABL_NODMA = true;
#else
  constexpr bool ABL_NODMA = false;
#endif
#ifdef WS_ABL_NOSTATSTORE          // (the statistics arithmetic without its four slab stores per tile and wave)
  constexpr bool ABL_NOSTATSTORE = true;
#else
  constexpr bool ABL_NOSTATSTORE = false;
#endif
  constexpr int NLD = ADD ? 10 : 0, NST = (ABL_NOSTORE ? 0 : 8) + (STATS && !ABL_NOSTATSTORE ? 1 : 0);    // vector-memory operations of a tile besides its LDS-DMA pieces
  constexpr int NPC = ABL_NODMA ? 0 : 2;                          // LDS-DMA pieces per stage and wave
  constexpr int W0 = NPC * (LA - 1) + NLD, W1 = W0 + (NT >= 2 ? NLD : 0) + NST, W2 = W1 + (NT >= 2 ? NST : 0);   // vmcnt of the stage wait in tile 0, tile 1, later tiles
  // (LA < KS: the pieces of slice k were issued LA slices earlier - in the previous tile, ahead of its epilogue's stores, for k < LA only)
  constexpr int WH0 = NPC * (LA - 1), WH1 = WH0 + NST;
  static_assert(LA >= KS || !ADD, "the half-tile look-ahead has no addend form");
  static_assert(W2 <= 63, "vmcnt is a 6-bit counter");
  constexpr int OB = ROWS ? BM * 512 : 0;                        // the output tile of the panel: 128 rows x 512 bytes
#ifdef WS_NO_MFMA_STATS      // (harness: the statistics of every tile on the VALU path, as the first row-form build)
  constexpr bool MSTAT = false;
#else
  constexpr bool MSTAT = STATS && ROWS;
#endif
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STG + OB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- schedule: workgroup -> (panel, pixel-tile stream); the panels of a pixel tile sit on CUs of ONE XCD ----
  const int G = gridDim.x, c8 = G >> 3;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int np = a.Cd / BN;
  const int spx = c8 / np;                                       // streams per XCD
#if defined(WS_ABL_PANEL_XCD)      // timing ablation: the panels of a pixel tile on different XCDs
  const int lin = xcd * c8 + idx8, nstreams = G / np;
  const int panel = lin % np, stream = lin / np;
#else
  const int panel = idx8 % np, stream = xcd * spx + idx8 / np, nstreams = 8 * spx;
#endif
  const int mt_total = (a.M + BM - 1) / BM;
  const int nmy = stream < mt_total ? (mt_total - stream + nstreams - 1) / nstreams : 0;
  if (nmy == 0) return;
  const int n0w = panel * BN + wave * 32;
#ifdef WS_DESYNC
  // harness experiment (round 5): do the store bursts of all 256 CUs at once cost the skeleton its bandwidth?  Start the streams of a launch
  // spread over WS_DESYNC x 64 cycles (the four panels of a pixel tile keep their common start)
  for (int i = 0; i < ((stream * 37) % 16) * (WS_DESYNC / 16); ++i) __builtin_amdgcn_s_sleep(1);
#endif

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  // ---- my weights: MFMA operand fragments for channels n0w + 16 j + (lane & 15), k = 32 q + 8 (lane >> 4) .. + 7 ----
  bf16x8 fw[2 * KS][2];
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = (unsigned)(n0w + 16 * j + l15) * (unsigned)a.Ktot * 2u + (unsigned)(32 * q + 8 * lg) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }

  // ---- issue side: stage (tile it_ti, slice it_k) = 16 pieces of 1 KiB, two per wave: rows 16 wave + 8 i + (lane >> 3), the
  // 16-byte chunk stored at position lane & 7 of row r is source chunk (lane & 7) ^ ((r >> 1) & 7) (conflict-free fragment reads) ----
  const int prow = wave * 16 + (lane >> 3);
  const int cch0 = (lane & 7) ^ ((lane >> 4) & 3);
  const unsigned lds2 = (unsigned)a.lds * 2u;
  int it_ti = 0, it_k = 0, islot = 0;
  unsigned rowoff[2];
  auto issue_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = (stream + it_ti * nstreams) * BM + prow + 8 * i;
#ifdef WS_ABL_WRAP_SRC      // timing ablation: the pixel stream from an L2- / MALL-resident window of WS_ABL_WRAP_SRC rows (results are garbage)
      rowoff[i] = (it_ti < nmy && m < a.M) ? (unsigned)(m % WS_ABL_WRAP_SRC) * lds2 + (unsigned)((cch0 ^ ((i & 1) << 2)) * 16) : WS_OOB;
#else
      rowoff[i] = (it_ti < nmy && m < a.M) ? (unsigned)m * lds2 + (unsigned)((cch0 ^ ((i & 1) << 2)) * 16) : WS_OOB;
#endif
    }
  };
  auto issue_stage = [&]() {
    unsigned char* const sa = smem + islot * STG + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) ws_dma16(rs_a, sa + i * 1024, rowoff[i] != WS_OOB ? rowoff[i] + (unsigned)(it_k * 128) : WS_OOB);
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
  // the weights are needed from here on: one counted wait now (the pieces above stay in flight).  Without this the compiler's
  // waitcnt pass carries the pending loads into the loop and waits for them, with ever smaller counts, in front of every MFMA group.
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fw[q][j]));

  // ---- consumer ----
  f32x4 acc[8][2];        // [pixel tile i: pixels 16 i + (lane & 15)][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int sw = (l15 >> 1) & 7;
  int cslot = 0;
  const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);            // first of my 8 channels in a store (after the lane swap)

  ws_u32x4 radd[ADD ? 8 : 1];
  // optional ReLU bit mask of the addend (css_conv2d_dgrad_add_masked): one byte per 16-byte vector.  The wave's 16 x 4 vectors of pixel
  // tile i are 16 rows x one dword of mask: lane (l15, lg) fetches the dwords of pixel tiles lg and lg + 4 (two registers, two requests -
  // one register per vector sends the K = 256 instance to scratch) and the epilogue takes its byte from the owner with ds_bpermute
  unsigned rmk[ADD ? 2 : 1];
  const bool has_mask = ADD && a.add_mask != nullptr;
  // ---------------- epilogue of a tile (rows m0e ..) ----------------
  // ROWS form (the default): the packed tile is written to the 64 KiB LDS output buffer and read back as whole 512-byte rows behind an
  // s_barrier - EVERY wave of the workgroup must call the epilogue, for the same tile, in uniform control flow (the persistent loop
  // guarantees it: the tile index is workgroup-uniform).  Only the WS_NO_ROWS / WS_PP forms store straight from registers without LDS or barrier.
  auto epilogue = [&](int m0e) {
    const int bnd = STATS ? (m0e / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;     // rows >= bnd: next statistics group (stage 2 sums them)
    // sum and sum of squares of the bf16-ROUNDED outputs (what the batch norm will read), two values per instruction (v_pk_add_f32 /
    // v_pk_fma_f32), accumulated in the loop that packs the tile (same order per lane as conv_igemm_p8_kernel: i = 0 .. 7, then the 16 pixels
    // of a lane row); the row test only in a tile that straddles a statistics-group boundary (one tile per group)
    ws_f32x2 s01[2], s23[2], q01[2], q23[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s01[j] = s23[j] = q01[j] = q23[j] = ws_f32x2{0.f, 0.f};
    const bool whole = m0e + BM <= bnd;
    // pixel tile i: pack (the statistics see the packed values), exchange -> lane (l15, lg) holds 8 consecutive channels of pixel 16 i + l15
    auto pack_tile = [&](int i, bool test, bool accumulate = true) -> ws_u32x4 {
      unsigned lo[2], hi[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        lo[j] = ws_pack2(acc[i][j][0], acc[i][j][1]);
        hi[j] = ws_pack2(acc[i][j][2], acc[i][j][3]);
        if (STATS && accumulate) {
          ws_f32x2 v01 = {ws_lo(lo[j]), ws_hi(lo[j])}, v23 = {ws_lo(hi[j]), ws_hi(hi[j])};
          if (test && !(m0e + 16 * i + l15 < bnd)) { v01 = ws_f32x2{0.f, 0.f}; v23 = ws_f32x2{0.f, 0.f}; }   // (rows >= M hold zeros already)
          s01[j] += v01; s23[j] += v23;
          q01[j] += v01 * v01; q23[j] += v23 * v23;
        }
      }
      ws_swap16(lo[0], lo[1]);
      ws_swap16(hi[0], hi[1]);
      return ws_u32x4{lo[0], hi[0], lo[1], hi[1]};
    };
    if constexpr (ROWS) {
      // registers -> LDS as the waves hold the tile (pixel 16 i + l15, 8 channels = one 16-byte chunk; chunk c of row r at position
      // c ^ (r & 31): 16 lanes of a write cover the 64 banks once) -> one barrier -> two whole 512-byte rows per instruction
      // (the LDS addresses do not depend on the tile: made opaque here, or the compiler keeps all 16 of them in registers across the K loop;
      // row 16 i + l15 -> (r & 31) = 16 (i & 1) | l15, row 16 wave + 2 i + half -> 16 (wave & 1) | 2 i | half: the swizzle splits into a lane part
      // and a constant XOR per i)
      const int cc = wave * 4 + 2 * (lg & 1) + (lg >> 1);
      int wb0 = NS * STG + l15 * 512 + ((cc ^ l15) << 4);
      int rb0 = NS * STG + (16 * wave + (lane >> 5)) * 512 + ((((lane & 31) ^ (16 * (wave & 1) + (lane >> 5)))) << 4);
      asm volatile("" : "+v"(wb0), "+v"(rb0));
      auto to_lds = [&](bool test, bool accumulate) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<ws_u32x4*>(smem + (wb0 ^ ((i & 1) << 8)) + i * 8192) = pack_tile(i, test, accumulate);
      };
      if (!STATS || (MSTAT && whole)) to_lds(false, false);
      else if (whole) to_lds(false, true);
      else to_lds(true, true);
      if constexpr (MSTAT) {
        if (whole) {
          // Round 5: the statistics of a whole tile on the MATRIX pipe (idle in the epilogue) instead of ~190 VALU instructions per wave: my 128 pixels x 32
          // channels come back from the LDS tile TRANSPOSED (ds_read_b64_tr_b16: lane (l15, lg) gets channel 16 j + l15 of pixels 8 lg .. 8 lg + 7 of a
          // 32-pixel block - the A operand of v_mfma_f32_16x16x32_bf16 with K = pixels, and equally its B operand); per 16 channels and 32 pixels
          // D_sq += F x F (the Gram matrix: its diagonal is the sum of squares) and D_sum += F x ones (every column the channel sums).  The sums arrive in
          // the lanes the DPP path leaves them in; the diagonal is fetched with four ds_bpermute.  bf16 x bf16 products are exact in fp32, so this is
          // the same quantity summed in another (fixed) order; a tile that straddles a statistics-group boundary keeps the VALU path with its row test.
          // (inline asm on purpose: through the builtin the waitcnt pass sees an LDS read that may alias the LDS-DMA pieces in flight and drains
          // them all - s_waitcnt vmcnt(0) once per tile - as csrc/conv_wgrad.hip found; the tile read here was written by this wave's own ds_writes)
          const int q4 = l15 >> 2, p4 = l15 & 3, R = 8 * lg + q4;
          unsigned tb0 = (unsigned)(uintptr_t)(ws_lds_void*)(smem + NS * STG) + (unsigned)(R * 512 + (((4 * wave + (p4 >> 1)) ^ R) << 4) + 8 * (p4 & 1));
          asm volatile("" : "+v"(tb0));
          const bf16x8 ones = __builtin_bit_cast(bf16x8, ws_u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const unsigned ta = tb0 ^ (unsigned)(32 * j), tb = ta ^ 64u;
            s16x4 fa4[4], fb4[4];
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:2048\n\t"
                         "ds_read_b64_tr_b16 %2, %8 offset:16384\n\tds_read_b64_tr_b16 %3, %9 offset:18432\n\t"
                         "ds_read_b64_tr_b16 %4, %8 offset:32768\n\tds_read_b64_tr_b16 %5, %9 offset:34816\n\t"
                         "ds_read_b64_tr_b16 %6, %8 offset:49152\n\tds_read_b64_tr_b16 %7, %9 offset:51200\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(fa4[0]), "=&v"(fb4[0]), "=&v"(fa4[1]), "=&v"(fb4[1]), "=&v"(fa4[2]), "=&v"(fb4[2]), "=&v"(fa4[3]), "=&v"(fb4[3])
                         : "v"(ta), "v"(tb)
                         : "memory");
            f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
              union { struct { s16x4 a, b; } h; bf16x8 f; } u;
              u.h.a = fa4[kc];
              u.h.b = fb4[kc];
              dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f, u.f, dsq, 0, 0, 0);
              dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f, ones, dsum, 0, 0, 0);
            }
            // D[channel 4 lg + r][column l15]: every column of dsum is the channel sums; the diagonal of dsq sits in lane l15 = 4 lg + r, register r
            const float dg = p4 == 0 ? dsq[0] : p4 == 1 ? dsq[1] : p4 == 2 ? dsq[2] : dsq[3];
            s01[j] = ws_f32x2{dsum[0], dsum[1]};
            s23[j] = ws_f32x2{dsum[2], dsum[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((20 * lg + r) * 4, __builtin_bit_cast(int, dg)));
              if (r == 0) q01[j][0] = v; else if (r == 1) q01[j][1] = v; else if (r == 2) q23[j][0] = v; else q23[j][1] = v;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int ch = lane & 31;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 16 * wave + 2 * i + (lane >> 5);
        const int m = m0e + r;
        ws_u32x4 v = *reinterpret_cast<const ws_u32x4*>(smem + (rb0 ^ (i << 5)) + i * 1024);
        if (ADD) {
          ws_u32x4 q = radd[ADD ? i : 0];
          // mask bytes: dword (row 8 q' + (lane >> 3), chunks 4 (lane & 7) ..) sits in rmk[q'] of lane 8 (row & 7) + (chunk >> 2)
          const unsigned mw = (unsigned)__builtin_amdgcn_ds_bpermute((((2 * i + (lane >> 5)) & 7) * 8 + (ch >> 2)) * 4, (int)rmk[ADD ? i >> 2 : 0]);
          const unsigned mk = has_mask ? (mw >> (8 * (ch & 3))) & 0xFFu : 0xFFu;
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] &= keep_mask_bf16x2(mk, e);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ws_pack2(ws_lo(v[e]) + ws_lo(q[e]), ws_hi(v[e]) + ws_hi(q[e]));
        }
        if (ABL_NOSTORE) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        else if (a.st_nt & 2) __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ldd + (unsigned)(panel * BN + 8 * ch)) * 2u : WS_OOB), 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ldd + (unsigned)(panel * BN + 8 * ch)) * 2u : WS_OOB), 0, WS_STORE_AUX);
      }
    } else {
      auto to_mem = [&](bool test) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int m = m0e + 16 * i + l15;
          ws_u32x4 v = pack_tile(i, test);
          if (ADD) {
            ws_u32x4 r = radd[ADD ? i : 0];
            const unsigned mw = (unsigned)__builtin_amdgcn_ds_bpermute(((i & 3) * 16 + l15) * 4, (int)rmk[ADD ? i >> 2 : 0]);
            const unsigned mk = has_mask ? (mw >> (8 * (2 * (lg & 1) + (lg >> 1)))) & 0xFFu : 0xFFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] &= keep_mask_bf16x2(mk, e);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ws_pack2(ws_lo(v[e]) + ws_lo(r[e]), ws_hi(v[e]) + ws_hi(r[e]));
          }
          if (ABL_NOSTORE) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#ifdef WS_ABL_STORE_ROWS    // timing ablation: the same bytes of the same tile, but every store instruction covers 2 rows x 512 contiguous bytes (results are garbage)
          else __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m0e + 16 * wave + 2 * i + (lane >> 5) < a.M ? ((unsigned)(m0e + 16 * wave + 2 * i + (lane >> 5)) * (unsigned)a.ldd + (unsigned)(panel * BN + 8 * (lane & 31))) * 2u : WS_OOB), 0, WS_STORE_AUX);
#elif defined(WS_ABL_STORE_SEG)      // timing ablation: as _STORE_ROWS with segments of WS_ABL_STORE_SEG bytes (128: 8 rows x one cache line per instruction; 256: 4 rows x two)
          else {
            constexpr int SPR = 512 / WS_ABL_STORE_SEG, LPS = WS_ABL_STORE_SEG / 16, RPI = 64 / LPS;      // segments per panel row, lanes per segment, rows per instruction
            const int mr = m0e + (wave / SPR) * (128 / (8 / SPR)) + RPI * i + lane / LPS;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(mr < a.M ? ((unsigned)mr * (unsigned)a.ldd + (unsigned)(panel * BN + (wave % SPR) * (WS_ABL_STORE_SEG / 2) + 8 * (lane % LPS))) * 2u : WS_OOB), 0, WS_STORE_AUX);
          }
#elif defined(WS_ABL_WRAP_DST)      // timing ablation: the output into a window of WS_ABL_WRAP_DST rows that stays in L2 / MALL (results are garbage)
          else __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m < a.M ? ((unsigned)(m % WS_ABL_WRAP_DST) * (unsigned)a.ldd + (unsigned)nl) * 2u : WS_OOB), 0, WS_STORE_AUX);
#else
          else __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ldd + (unsigned)nl) * 2u : WS_OOB), 0, WS_STORE_AUX);
#endif
        }
      };
      if (!STATS || whole) to_mem(false);
      else to_mem(true);
    }
    if (STATS) {
      // the wave's slab entry (32 channels: sums, then sums of squares) in ONE store: every lane of a row holds the row's sums after the
      // DPP rotations; lane l15 = 0 / 1 / 2 / 3 of each lane row sends (sum, j = 0) / (squares, j = 0) / (sum, j = 1) / (squares, j = 1)
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(m0e >> 7) * 2u * (unsigned)a.Cd * 4u;
      ws_f32x4 o[2][2];
      if (MSTAT && whole) {      // (the matrix-pipe path left the totals in every lane of a lane row already)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          o[j][0] = ws_f32x4{s01[j][0], s01[j][1], s23[j][0], s23[j][1]};
          o[j][1] = ws_f32x4{q01[j][0], q01[j][1], q23[j][0], q23[j][1]};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          o[j][0] = ws_f32x4{ws_row16_sum(s01[j][0]), ws_row16_sum(s01[j][1]), ws_row16_sum(s23[j][0]), ws_row16_sum(s23[j][1])};
          o[j][1] = ws_f32x4{ws_row16_sum(q01[j][0]), ws_row16_sum(q01[j][1]), ws_row16_sum(q23[j][0]), ws_row16_sum(q23[j][1])};
        }
      }
      ws_f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = (l15 & 1) ? o[0][1][e] : o[0][0][e], x1 = (l15 & 1) ? o[1][1][e] : o[1][0][e];
        ov[e] = (l15 & 2) ? x1 : x0;
      }
      const int n = (l15 & 1) * a.Cd + n0w + 16 * ((l15 >> 1) & 1) + 4 * lg;
      if (ABL_NOSTATSTORE) asm volatile("" ::"v"(ov[0]), "v"(ov[1]), "v"(ov[2]), "v"(ov[3]));
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, ov), rs_s, (int)(l15 < 4 ? base + (unsigned)n * 4u : WS_OOB), 0, 0);
    }
  };
  auto load_addend = [&](int m0e) {
    // requested a whole tile ahead of its use: by then everything older (LDS-DMA pieces of earlier stages) has landed anyway, so
    // waiting for it costs nothing - an addend requested inside the epilogue drains the whole in-order queue once per tile
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.addend), 0, (int)a.add_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.add_mask), 0, (int)a.mask_bytes, 0x00020000);
    // (always issued - the vmcnt bookkeeping counts NLD = 10 loads per tile; without a mask the offset is out of range and the dword unused)
    if constexpr (ROWS) {
      // the row form: my vectors are (row 16 wave + 2 i + (lane >> 5), chunk lane & 31) - whole lines of the addend, too
#pragma unroll
      for (int q = 0; q < (ADD ? 2 : 0); ++q) {
        const int m = m0e + 16 * wave + 8 * q + (lane >> 3);
        rmk[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_k, (int)((has_mask && m < a.M) ? (unsigned)m * ((unsigned)a.Cd >> 3) + (unsigned)(panel * 32 + 4 * (lane & 7)) : WS_OOB), 0, 0);
      }
#pragma unroll
      for (int i = 0; i < (ADD ? 8 : 0); ++i) {
        const int m = m0e + 16 * wave + 2 * i + (lane >> 5);
        radd[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ld_add + (unsigned)(panel * BN + 8 * (lane & 31))) * 2u : WS_OOB), 0, WS_ADD_AUX);
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < (ADD ? 2 : 0); ++q) {
      const int m = m0e + 16 * (lg + 4 * q) + l15;
      rmk[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_k, (int)((has_mask && m < a.M) ? (unsigned)m * ((unsigned)a.Cd >> 3) + ((unsigned)n0w >> 3) : WS_OOB), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < (ADD ? 8 : 0); ++i) {
      const int m = m0e + 16 * i + l15;
      radd[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ld_add + (unsigned)nl) * 2u : WS_OOB), 0, WS_ADD_AUX);
    }
  };
  auto mfma_half = [&](const bf16x8 (&fa)[8], int q, bool first) {
    if (first) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], z, 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[i], acc[i][j], 0, 0, 0);
    }
  };

#ifdef WS_STAMP
  unsigned long long wsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt0 = 0, wt1 = 0;
#endif
  if constexpr (WS_PP != 0) {
    static_assert(NT == 2 || KS == 8, "the ping-pong variant counts two tiles of look-ahead (K <= 256 only)");
    // ---- ping-pong: waves 0-3 and waves 4-7 (one of each per SIMD) run half a stage apart - READ segment (last tile's epilogue,
    // fragment reads, next LDS-DMA pieces, waits) of one group beside the MFMA segment of the other, two barriers per stage ----
    const int grp = wave >> 2;
    ws_wait_vm<NPC * (LA - 1)>();           // my pieces of stage 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int ti = 0; ti < nmy; ++ti) {
      const int m0 = (stream + ti * nstreams) * BM;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        if (k == 0) {
          if (ti > 0) epilogue(m0 - nstreams * BM);       // beside the other group's MFMAs
          if (ADD) load_addend(m0);
          __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned char* ab = smem + cslot * STG + l15 * 128;
        bf16x8 fa[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 8; ++i) fa[h][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + (((4 * h + lg) ^ sw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL_NODMA) issue_stage();      // stage + LA into the slot read two stages ago
        __builtin_amdgcn_sched_barrier(0);
        // my pieces of the NEXT stage have landed: younger than them are 2 (LA - 1) pieces and the epilogues / addend requests at the
        // starts of this tile and (for all but the last slice) of the previous one
        ws_wait_stage<NPC * (LA - 1), NST, NLD>((ti >= 1 ? 1 : 0) + ((ti >= 2 && k <= KS - 2) ? 1 : 0), 1 + ((ti >= 1 && k <= KS - 2) ? 1 : 0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!ABL_NOMFMA) {
          mfma_half(fa[0], 2 * k, k == 0);
          mfma_half(fa[1], 2 * k + 1, false);
        } else if (k == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        cslot = cslot == NS - 1 ? 0 : cslot + 1;
      }
    }
    epilogue((stream + (nmy - 1) * nstreams) * BM);
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
  for (int ti = 0; ti < nmy; ++ti) {
    const int m0 = (stream + ti * nstreams) * BM;
    if (ABL_NOMFMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (ADD) load_addend(m0);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      WS_T(wt0);
      // my two pieces of this stage have landed (see the header for the counts)
      if constexpr (LA < KS) {
        if (ti >= 1 && k < LA) ws_wait_vm<WH1>();
        else ws_wait_vm<WH0>();
      } else {
        if (ti >= 2) ws_wait_vm<W2>();
        else if (ti == 1) ws_wait_vm<W1>();
        else ws_wait_vm<W0>();
      }
      WS_T(wt1); WS_ACC(k == 0 ? 0 : 5, wt1, wt0);
      __builtin_amdgcn_s_barrier();       // everybody's pieces have landed; everybody is done reading the previous stages
      asm volatile("" ::: "memory");
      WS_T(wt0); WS_ACC(k == 0 ? 1 : 6, wt0, wt1);
      __builtin_amdgcn_sched_barrier(0);
      if (!ABL_NODMA) issue_stage();      // stage + LA into the slot read two stages ago
      WS_T(wt1); WS_ACC(2, wt1, wt0);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* ab = smem + cslot * STG + l15 * 128;
      // all fragment reads of the stage first (16 x ds_read_b128, back to back), then its 32 MFMAs back to back: left to itself the
      // compiler reads one fragment, waits for it, issues its two MFMAs, and so on - every LDS latency exposed.  (The addend variant
      // holds 32 more registers: it reads and multiplies one 32-channel half of the stage at a time.)
      constexpr int HB = (ADD || KS > 4) ? 1 : 2;        // K halves per batch (K = 512: 128 registers of weights)
#pragma unroll
      for (int hb = 0; hb < (ABL_NOMFMA ? 0 : 2); hb += HB) {
        bf16x8 fa[HB][8];
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
          for (int i = 0; i < 8; ++i) fa[h][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + (((4 * (hb + h) + lg) ^ sw) << 4));
        if (KS <= 4) __builtin_amdgcn_sched_barrier(0);      // (K = 512: 128 registers of weights - the compiler interleaves reads and MFMAs itself)
#pragma unroll
        for (int h = 0; h < HB; ++h) mfma_half(fa[h], 2 * k + hb + h, k == 0 && hb + h == 0);
        if (KS <= 4) __builtin_amdgcn_sched_barrier(0);
      }
      cslot = cslot == NS - 1 ? 0 : cslot + 1;
      WS_T(wt0); WS_ACC(3, wt0, wt1);
    }
    __builtin_amdgcn_sched_barrier(0);
    WS_T(wt0);
    epilogue(m0);
    WS_T(wt1); WS_ACC(4, wt1, wt0);
#ifdef WS_STAMP
    if (ti >= 2) wsum[7] += 1;
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  }
#ifdef WS_STAMP
  if (a.bias && lane == 0) {
    unsigned long long* o = (unsigned long long*)a.bias + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = wsum[i];
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ghost pieces (stages past my last tile) must have landed before the LDS is released
}

// Shapes this kernel takes (everything else stays on the 256x256 persistent kernels)
static int g_ws_off = -1;      // -1: read CSS_NO_WS_CONV on first use
void css_conv_ws_set_enabled(int on) { g_ws_off = on ? 0 : 1; }     // (A/B timing and the parity harness: scripts/conv_bench.hip)
bool css_conv_ws_supported(const ConvArgs& a, int n_cu) {
  if (g_ws_off < 0) g_ws_off = getenv("CSS_NO_WS_CONV") != nullptr;
  const bool off = g_ws_off != 0;
#ifdef WS_STAMP
  if (off || a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0 || a.Hs != a.Hd || a.Ws != a.Wd) return false;      // (a.bias carries the stamp buffer)
#else
  if (off || a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0 || a.Hs != a.Hd || a.Ws != a.Wd || a.bias) return false;
#endif
  if (a.Ktot != a.Cs || (a.Cs != 64 && a.Cs != 128 && a.Cs != 256 && a.Cs != 512) || a.Cd < 256 || a.Cd % 256 || a.lds % 8 || a.ldd % 8) return false;
  if (a.Cs == 512 && a.addend) return false;        // (128 registers of weights + 32 of addend + 64 accumulators + fragments: no room)
  if (a.stats && a.addend) return false;
  if (a.addend && a.ld_add % 8) return false;
  const int np = a.Cd / 256;
  if (n_cu < 8 || n_cu % 8 || (n_cu / 8) % np) return false;
  if ((size_t)a.M * a.ldd * 2 >= 0x7FFFFFF0ull || (a.addend && (size_t)a.M * a.ld_add * 2 >= 0x7FFFFFF0ull)) return false;
  return a.M > 0;
}

template <int KS>
static void launch_ws(const ConvArgs& a, int grid, hipStream_t st) {
  const dim3 g(grid), b(512);
  if (a.stats) hipLaunchKernelGGL((conv_ws_kernel<KS, true, false>), g, b, 0, st, a);
  else if (a.addend) hipLaunchKernelGGL((conv_ws_kernel<KS, false, true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_ws_kernel<KS, false, false>), g, b, 0, st, a);
}
void css_launch_conv_ws(ConvArgs a, int n_cu, hipStream_t st) {
  a.dst_bytes = (unsigned)((size_t)a.M * a.ldd * 2);
  if (a.stats) a.stat_bytes = (unsigned)((size_t)2 * cdiv(a.M, 256) * 2 * a.Cd * 4);
  if (a.addend) a.add_bytes = (unsigned)((size_t)a.M * a.ld_add * 2);
  if (a.add_mask) a.mask_bytes = (unsigned)((size_t)a.M * (a.Cd / 8));
  if (a.Cs == 512) {
    if (a.stats) hipLaunchKernelGGL((conv_ws_kernel<8, true, false>), dim3(n_cu), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_ws_kernel<8, false, false>), dim3(n_cu), dim3(512), 0, st, a);
  } else if (a.Cs == 256) launch_ws<4>(a, n_cu, st);
  else if (a.Cs == 128) launch_ws<2>(a, n_cu, st);
  else launch_ws<1>(a, n_cu, st);
}
