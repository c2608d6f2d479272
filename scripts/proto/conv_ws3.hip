// PROTOTYPE (round 3, not part of the library): conv_ws3_kernel - conv_ws_kernel's short-K 1x1 class with 64-pixel half tiles, two
// accumulator sets and the epilogue of one half interleaved with the MFMAs of the next (template parameter PP: the two wave groups half a
// stage apart).  Bit-identical to conv_ws_kernel on scripts/ws_bench.hip; measured in profiles/r03_conv_ws_where_the_time_goes.txt
// (section 4): 256 -> 512 plain 42-43 us against 50-55, statistics 56-66 against 61-67; 256 -> 1024 plain 106-114 against 101-109,
// statistics 115-125 against 125-128 - the N = 1024 shape sits on a floor of ~105 us that neither schedule moves.
// To build it into the harness: paste this block into css_amd/csrc/conv_ws.hip in front of "Shapes this kernel takes" (it uses that
// file's helpers and WS_T / WS_ACC stamp macros; needs <type_traits>) and route K = 256 without addend to it in css_launch_conv_ws.
// ---------------------------------------------------------------------------------------------------------------------------------
// conv_ws3_kernel (round 3, K = 256 without addend; CSS_WS3=1): the epilogue of a half tile runs INSIDE the next half tile's MFMA stream.
// The stamps of conv_ws_kernel (profiles/r03_conv_ws_where_the_time_goes.txt) show its eight waves converting and storing a tile in
// lockstep while the matrix pipe idles (fragment reads + MFMAs are a third of a tile).  Here a tile is two HALVES of 64 pixels with an
// accumulator set each (2 x 32 registers - what one 128-pixel tile takes): while half h multiplies into its set, the other set - the
// previous half - is converted, summed for the statistics and stored, a few VALU instructions and one store between MFMAs.
//   * stage = 64 pixels x 128 channels (16 KiB, 256-byte rows: 16-byte chunk c of row r at c ^ (r & 15)), 32 MFMAs per wave and stage
//     as before, two stages per half at K = 256; ring of 10 stages, 8 in flight (LA);
//   * per stage two row blocks (16 pixels x 32 channels) of the previous half leave: convert, lane swap, [statistics from the packed
//     values before the swap - no second conversion], one 16-byte store each; the statistics of a 128-row slab (two halves) are written
//     after its second half has been drained;
//   * counted waits: younger than the pieces of stage g are at least 2 (LA - 1) pieces + 2 stores per stage since (the statistics
//     stores come on top: waiting with the smaller count is safe - it can only ask for more of the oldest operations to be complete).
template <bool STATS, bool PP>
__global__ __launch_bounds__(512) void conv_ws3_kernel(const ConvArgs a) {
  constexpr int KH = 2;                                          // stages (128 channels) per half tile: K = 256
  constexpr int BN = 256, LA = 8, NS = LA + 2, STG = 64 * 256, SPS = 4 / KH;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int G = gridDim.x, c8 = G >> 3;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int np = a.Cd / BN, spx = c8 / np;
  const int panel = idx8 % np, stream = xcd * spx + idx8 / np, nstreams = 8 * spx;
  const int mt_total = (a.M + 127) / 128;
  const int nmy = stream < mt_total ? (mt_total - stream + nstreams - 1) / nstreams : 0;
  if (nmy == 0) return;
  const int n0w = panel * BN + wave * 32;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);

  // my weights: channels n0w + 16 j + (lane & 15), k = 32 q + 8 (lane >> 4) .. + 7, q = 0 .. 4 KH - 1
  bf16x8 fw[4 * KH][2];
#pragma unroll
  for (int q = 0; q < 4 * KH; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = (unsigned)(n0w + 16 * j + l15) * (unsigned)a.Ktot * 2u + (unsigned)(32 * q + 8 * lg) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }

  // issue side: a stage is 16 pieces of 1 KiB (4 rows x 256 bytes), two per wave: rows 8 wave + 4 i + (lane >> 4) of the half tile
  const int prow = wave * 8 + (lane >> 4);
  const unsigned lds2 = (unsigned)a.lds * 2u;
  int it_ti = 0, it_h = 0, it_k = 0, islot = 0;
  unsigned rowoff[2];
  auto issue_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = prow + 4 * i;                                 // row inside the half tile
      const int m = (stream + it_ti * nstreams) * 128 + 64 * it_h + r;
      rowoff[i] = (it_ti < nmy && m < a.M) ? (unsigned)m * lds2 + (unsigned)((((lane & 15) ^ (r & 15)) << 4)) : WS_OOB;
    }
  };
  auto issue_stage = [&]() {
    unsigned char* const sa = smem + islot * STG + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) ws_dma16(rs_a, sa + i * 1024, rowoff[i] != WS_OOB ? rowoff[i] + (unsigned)(it_k * 256) : WS_OOB);
    islot = islot == NS - 1 ? 0 : islot + 1;
    if (++it_k == KH) {
      it_k = 0;
      if (++it_h == 2) { it_h = 0; ++it_ti; }
      issue_rows();
    }
  };
  issue_rows();
#pragma unroll
  for (int s0 = 0; s0 < LA; ++s0) issue_stage();
#pragma unroll
  for (int q = 0; q < 4 * KH; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fw[q][j]));

  f32x4 acc[2][4][2];        // [half][pixel block i: pixels 16 i + (lane & 15)][channel block j: channels 16 j + 4 (lane >> 4) + reg]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cslot = 0;
  const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);
  ws_f32x2 s01[2], s23[2], q01[2], q23[2];                        // statistics of the slab being drained, per channel block j
#pragma unroll
  for (int j = 0; j < 2; ++j) { s01[j] = s23[j] = q01[j] = q23[j] = ws_f32x2{0.f, 0.f}; }

  // one row block (16 pixels x my 32 channels) of accumulator set hs, rows mrow + 16 rb .., leaves: convert, statistics, swap, store
  auto drain = [&](auto hs_c, auto rb_c, int mrow, int bnd, bool live) {
    constexpr int hs = decltype(hs_c)::value, rb = decltype(rb_c)::value;
    const int m = mrow + 16 * rb + l15;
    unsigned lo0 = ws_pack2(acc[hs][rb][0][0], acc[hs][rb][0][1]), hi0 = ws_pack2(acc[hs][rb][0][2], acc[hs][rb][0][3]);
    unsigned lo1 = ws_pack2(acc[hs][rb][1][0], acc[hs][rb][1][1]), hi1 = ws_pack2(acc[hs][rb][1][2], acc[hs][rb][1][3]);
    if (STATS) {
      const bool keep = m < bnd;                                  // (rows >= M hold zeros already; rows >= bnd belong to the next statistics group)
      const unsigned w[2][2] = {{lo0, hi0}, {lo1, hi1}};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ws_f32x2 v01 = {ws_lo(w[j][0]), ws_hi(w[j][0])}, v23 = {ws_lo(w[j][1]), ws_hi(w[j][1])};
        if (!keep) { v01 = ws_f32x2{0.f, 0.f}; v23 = ws_f32x2{0.f, 0.f}; }
        s01[j] += v01; s23[j] += v23;
        q01[j] += v01 * v01; q23[j] += v23 * v23;
      }
    }
    ws_swap16(lo0, lo1);
    ws_swap16(hi0, hi1);
    const ws_u32x4 v = {lo0, hi0, lo1, hi1};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs_d, (int)((live && m < a.M) ? ((unsigned)m * (unsigned)a.ldd + (unsigned)nl) * 2u : WS_OOB), 0, 0);
  };
  auto emit_stats = [&](int m0s, bool live) {                     // the slab whose rows start at m0s has been drained whole
    const unsigned base = (unsigned)(m0s >> 7) * 2u * (unsigned)a.Cd * 4u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ws_f32x4 os = {ws_row16_sum(s01[j][0]), ws_row16_sum(s01[j][1]), ws_row16_sum(s23[j][0]), ws_row16_sum(s23[j][1])};
      ws_f32x4 oq = {ws_row16_sum(q01[j][0]), ws_row16_sum(q01[j][1]), ws_row16_sum(q23[j][0]), ws_row16_sum(q23[j][1])};
      const int n = n0w + 16 * j + 4 * lg;
      const bool ok = live && l15 == 0;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, os), rs_s, (int)(ok ? base + (unsigned)n * 4u : WS_OOB), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, oq), rs_s, (int)(ok ? base + (unsigned)(a.Cd + n) * 4u : WS_OOB), 0, 0);
      s01[j] = s23[j] = q01[j] = q23[j] = ws_f32x2{0.f, 0.f};
    }
  };
  auto wait_stage = [&](int g) {      // pieces of global stage g: >= 2 (LA - 1) pieces + SPS stores per loop stage since their issue are younger
    constexpr int B0 = 2 * (LA - 1);
    if (g >= LA) ws_wait_vm<B0 + LA * SPS>();
    else if (g >= 4) ws_wait_vm<B0 + 4 * SPS>();
    else if (g >= 2) ws_wait_vm<B0 + 2 * SPS>();
    else ws_wait_vm<B0>();
  };
#ifdef WS_STAMP
  unsigned long long wsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt0 = 0, wt1 = 0;
#endif
  auto stage = [&](auto h_c, auto k_c, int ti, int m0) {
    constexpr int h = decltype(h_c)::value, k = decltype(k_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    WS_T(wt0);
    wait_stage((ti * 2 + h) * KH + k);
    WS_T(wt1); WS_ACC(0, wt1, wt0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WS_T(wt0); WS_ACC(1, wt0, wt1);
    __builtin_amdgcn_sched_barrier(0);
    issue_stage();
    WS_T(wt1); WS_ACC(2, wt1, wt0);
    __builtin_amdgcn_sched_barrier(0);
    // the half being drained: h = 0 drains the previous slab's second half, h = 1 this slab's first half
    const int mrow = h == 0 ? m0 - nstreams * 128 + 64 : m0;
    const int m0d = h == 0 ? m0 - nstreams * 128 : m0;             // first row of the drained half's slab
    const int bnd = STATS ? (m0d / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;
    const bool live = h == 1 || ti > 0;
    const unsigned char* ab = smem + cslot * STG + l15 * 256;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {                               // two batches of 2 k-sub-steps x 4 pixel blocks
      bf16x8 fa[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(ab + i * 4096 + (((4 * (2 * qb + q) + lg) ^ l15) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[4 * k + 2 * qb + q][j], fa[q][i], acc[h][i][j], 0, 0, 0);
      if (qb == 0) drain(std::integral_constant<int, h ^ 1>{}, std::integral_constant<int, 2 * k>{}, mrow, bnd, live);
      else drain(std::integral_constant<int, h ^ 1>{}, std::integral_constant<int, 2 * k + 1>{}, mrow, bnd, live);
      // one MFMA, then a few of the drain's VALU instructions, and so on; its store at the end
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, STATS ? 4 : 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k == KH - 1) {
      // the drained half is empty now: its set accumulates the next-but-one half from zero
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h ^ 1][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (STATS && h == 0) emit_stats(m0d, ti > 0);                 // both halves of the previous slab are out
    }
    cslot = cslot == NS - 1 ? 0 : cslot + 1;
    WS_T(wt0); WS_ACC(3, wt0, wt1);
#ifdef WS_STAMP
    if (h == 1 && k == KH - 1 && ti >= 2) wsum[7] += 1;
#endif
  };

  // ---- PP: waves 0-3 and 4-7 (one of each per SIMD) half a stage apart - the LOAD segment of one group (16 fragment reads, the next
  // LDS-DMA pieces, the counted wait for the NEXT stage's pieces) beside the MFMA segment (32 MFMAs with the drain between them) of the
  // other; two barriers per stage, as conv_ws_kernel's WS_PP form.  At the wait inside stage g the pieces of stage g + 1 have at least
  // 2 (LA - 1) pieces + min(g, LA - 1) x SPS stores behind them. ----
  auto wait_next = [&](int g) {
    constexpr int B0 = 2 * (LA - 1);
    if (g >= LA - 1) ws_wait_vm<B0 + (LA - 1) * SPS>();
    else if (g >= 4) ws_wait_vm<B0 + 4 * SPS>();
    else if (g >= 2) ws_wait_vm<B0 + 2 * SPS>();
    else ws_wait_vm<B0>();
  };
  auto stage_pp = [&](auto h_c, auto k_c, int ti, int m0) {
    constexpr int h = decltype(h_c)::value, k = decltype(k_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    WS_T(wt0);
    const unsigned char* ab = smem + cslot * STG + l15 * 256;
    bf16x8 fa[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(ab + i * 4096 + (((4 * q + lg) ^ l15) << 4));
    __builtin_amdgcn_sched_barrier(0);
    issue_stage();
    WS_T(wt1); WS_ACC(2, wt1, wt0);
    __builtin_amdgcn_sched_barrier(0);
    wait_next((ti * 2 + h) * KH + k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WS_T(wt0); WS_ACC(0, wt0, wt1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WS_T(wt1); WS_ACC(1, wt1, wt0);
    __builtin_amdgcn_sched_barrier(0);
    const int mrow = h == 0 ? m0 - nstreams * 128 + 64 : m0;
    const int m0d = h == 0 ? m0 - nstreams * 128 : m0;
    const int bnd = STATS ? (m0d / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;
    const bool live = h == 1 || ti > 0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[4 * k + 2 * qb + q][j], fa[2 * qb + q][i], acc[h][i][j], 0, 0, 0);
      if (qb == 0) drain(std::integral_constant<int, h ^ 1>{}, std::integral_constant<int, 2 * k>{}, mrow, bnd, live);
      else drain(std::integral_constant<int, h ^ 1>{}, std::integral_constant<int, 2 * k + 1>{}, mrow, bnd, live);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, STATS ? 4 : 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k == KH - 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h ^ 1][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (STATS && h == 0) emit_stats(m0d, ti > 0);
    }
    cslot = cslot == NS - 1 ? 0 : cslot + 1;
    WS_T(wt0); WS_ACC(3, wt0, wt1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WS_T(wt1); WS_ACC(4, wt1, wt0);
    __builtin_amdgcn_sched_barrier(0);
#ifdef WS_STAMP
    if (h == 1 && k == KH - 1 && ti >= 2) wsum[7] += 1;
#endif
  };
  if constexpr (PP) {
    const int grp = wave >> 2;
    ws_wait_vm<2 * (LA - 1)>();             // my pieces of stage 0 have landed
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int ti = 0; ti < nmy; ++ti) {
      const int m0 = (stream + ti * nstreams) * 128;
      stage_pp(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ti, m0);
      stage_pp(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, ti, m0);
      stage_pp(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, ti, m0);
      stage_pp(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, ti, m0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else
  for (int ti = 0; ti < nmy; ++ti) {
    const int m0 = (stream + ti * nstreams) * 128;
    stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ti, m0);
    stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, ti, m0);
    stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, ti, m0);
    stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, ti, m0);
  }
  // the last slab's second half
  {
    const int m0 = (stream + (nmy - 1) * nstreams) * 128;
    const int bnd = STATS ? (m0 / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;
    drain(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, m0 + 64, bnd, true);
    drain(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, m0 + 64, bnd, true);
    drain(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, m0 + 64, bnd, true);
    drain(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, m0 + 64, bnd, true);
    if (STATS) emit_stats(m0, true);
  }
#ifdef WS_STAMP
  if (a.bias && lane == 0) {
    unsigned long long* o = (unsigned long long*)a.bias + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = wsum[i];
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

