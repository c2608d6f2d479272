// PROTOTYPE (round 5, not part of the library): conv_ws4_kernel - conv_ws_kernel's short-K 1x1 class on TWO independent four-wave workgroups per
// CU ("2 pixel halves x 4 channel quarters, the halves half a tile apart").  Outputs bit-identical to conv_ws_kernel, race screen of
// 20 launches clean, no scratch, 244-256 VGPRs; MEASURED SLOWER and therefore not shipped - profiles/r05_conv_ws4.txt:
//     256 -> 1024 plain 120.6 us against 104.2 (conv_ws_kernel, same process), statistics 142.5 against 124.3; 64 -> 256 statistics 109 against 72;
//     the start delay of a CU's second workgroup (0 ... 6144 cycles) changes nothing.
// About 6 % of the gap is tile quantisation (1057 tiles x 4 panels on 512 workgroups: 9 rounds for 8.26), ~4 % the harness's second-variant bias, the
// statistics forms pay twice the cross-lane reductions (a wave sums 64 rows instead of 128 before its 16-lane reduction).  What it shows: decoupling the
// two wave groups does NOT buy the overlap the lockstep stamps suggested - the class sits on its fill + store skeleton (92 us without any MFMA).
// To build it into the harness again: paste the kernel in front of "Shapes this kernel takes" in css_amd/csrc/conv_ws.hip, the dispatch block in front of
// css_launch_conv_ws and the routing lines at the top of that function's body, add `int ws_stagger;` to ConvArgs (launchers.h).
// ---------------------------------------------------------------------------------------------------------------------------------
// conv_ws4_kernel (round 5; K <= 256 without addend): the same weight-stationary class on TWO INDEPENDENT WORKGROUPS PER CU.
//
// What the stamps of conv_ws_kernel say (profiles/r03_conv_ws_where_the_time_goes.txt): its eight waves do everything in lockstep - all read
// fragments, all multiply, all convert and store - so the matrix pipe, the LDS and the vector-memory path are used in turns: 6.3 us per
// (128-pixel tile, panel) against 1.7 us of MFMAs; every wave reads the WHOLE pixel stage (512 KiB of LDS reads per tile).  Schedules that
// shift the two wave groups of ONE workgroup against each other (WS_PP, conv_ws3) did not pay: s_barrier couples them stage by stage.
// Here (2 pixel halves x 4 channel quarters, the halves half a tile apart):
//   * a workgroup is FOUR waves (256 threads) and owns a 256-channel panel; wave w holds channels 64 w .. 64 w + 63 for all of K as MFMA
//     operands (K = 256: 32 fragments = 128 VGPRs - what the K = 512 instance of conv_ws_kernel carries);
//   * its pixel stream runs in HALF tiles of 64 pixels: a stage is 64 pixels x 64 channels (8 KiB), a wave reads it once (8 ds_read_b128)
//     for 32 MFMAs - HALF the LDS bytes per MFMA of conv_ws_kernel; the two halves of a 128-pixel tile follow each other in the same
//     workgroup (same accumulator registers), so a wave still owns its 128-row statistics slab: the first half's 16-lane sums wait in 2 KiB
//     of LDS for the second half's;
//   * TWO such workgroups share a CU (74 KiB of LDS and 256 VGPRs each: __launch_bounds__(256, 2)); they take alternate tiles of the pixel
//     stream conv_ws_kernel's ONE workgroup walks, have barriers of their own and therefore drift freely against each other: one converts and
//     stores while the other multiplies.  The second workgroup starts a.ws_stagger x 64 cycles late (half a tile time) so that they begin out of phase;
//   * outputs are BIT-IDENTICAL to conv_ws_kernel (same MFMA instruction on the same K blocks in the same order); the statistics slabs are the
//     same sums in another order: (sum of rows 0..63 over 16 lanes) + (sum of rows 64..127 over 16 lanes) instead of one 8-term lane sum.
// vmcnt bookkeeping as above: the wait for "my two pieces of stage g" counts 2 (LA - 1) younger pieces + 8 stores per half tile finished since
// the pieces were issued (LA = NT half tiles); the statistics stores of every second half tile are NOT counted (a smaller count only waits for more).
template <int KS, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_ws4_kernel(const ConvArgs a) {
  constexpr int NT = KS == 4 ? 2 : (KS == 2 ? 4 : 6);            // half tiles of look-ahead
  constexpr int BH = 64, BN = 256, LA = NT * KS, NS = LA + 1;    // (one barrier per stage: the slot read ONE stage ago is free)
  constexpr int STG = BH * 128;                                  // one stage: 64 pixels x 128 bytes (64 channels)
  constexpr int NST = 8, NPC = 2;
  static_assert(NPC * (LA - 1) + NT * NST <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STG + (STATS ? 2048 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- schedule: virtual CU v = conv_ws_kernel's workgroup; its two workgroups here take alternate tiles of v's stream ----
  const int ncu = gridDim.x >> 1, c8 = ncu >> 3;
  const int v = blockIdx.x % ncu, half_wg = blockIdx.x / ncu;
  const int xcd = v & 7, idx8 = v >> 3;
  const int np = a.Cd / BN;
  const int spx = c8 / np;
  const int panel = idx8 % np, stream = xcd * spx + idx8 / np, nstreams = 8 * spx;
  const int mt_total = (a.M + 127) / 128;
  const int nstr = stream < mt_total ? (mt_total - stream + nstreams - 1) / nstreams : 0;      // tiles of the stream
  const int nmy = (nstr - half_wg + 1) >> 1;                                                   // ... of which mine (positions half_wg, half_wg + 2, ..)
  if (nmy <= 0) return;
  const int tstep = 2 * nstreams, t0 = stream + half_wg * nstreams;                            // my tiles: t0 + ti * tstep
  const int n0w = panel * BN + wave * 64;

  unsigned long long src_p = (unsigned long long)a.src, wt_p = (unsigned long long)a.wt;
  int src_n = (int)a.src_bytes, wt_n = (int)a.wt_bytes;
  asm volatile("" : "+s"(src_p), "+s"(wt_p), "+s"(src_n), "+s"(wt_n));
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_p, 0, src_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wt_p, 0, wt_n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);

  // ---- my weights: MFMA operand fragments for channels n0w + 16 j + (lane & 15), k = 32 q + 8 (lane >> 4) .. + 7 ----
  bf16x8 fw[2 * KS][4];
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = (unsigned)(n0w + 16 * j + l15) * (unsigned)a.Ktot * 2u + (unsigned)(32 * q + 8 * lg) * 2u;
      fw[q][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }
  if (half_wg && a.ws_stagger > 0) {           // the CU's second workgroup starts half a tile late (the weights are on their way meanwhile)
    for (int i = 0; i < a.ws_stagger; i += 64) __builtin_amdgcn_s_sleep(1);      // (s_sleep 1 = 64 cycles)
  }

  // ---- issue side: stage (tile it_ti, half it_h, slice it_k) = 8 pieces of 1 KiB, two per wave: rows 16 wave + 8 i + (lane >> 3) of the half, the
  // 16-byte chunk stored at position lane & 7 of row r is source chunk (lane & 7) ^ ((r >> 1) & 7) (conflict-free fragment reads) ----
  const int prow = wave * 16 + (lane >> 3);
  const int cch0 = (lane & 7) ^ ((lane >> 4) & 3);
  const unsigned lds2 = (unsigned)a.lds * 2u;
  int it_ti = 0, it_h = 0, it_k = 0, islot = 0;
  unsigned rowoff[2];
  auto issue_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = (t0 + it_ti * tstep) * 128 + it_h * BH + prow + 8 * i;
      rowoff[i] = (it_ti < nmy && m < a.M) ? (unsigned)m * lds2 + (unsigned)((cch0 ^ ((i & 1) << 2)) * 16) : WS_OOB;
    }
  };
  auto issue_stage = [&]() {
    unsigned char* const sa = smem + islot * STG + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) ws_dma16(rs_a, sa + i * 1024, rowoff[i] != WS_OOB ? rowoff[i] + (unsigned)(it_k * 128) : WS_OOB);
    islot = islot == NS - 1 ? 0 : islot + 1;
    if (++it_k == KS) {
      it_k = 0;
      if (++it_h == 2) { it_h = 0; ++it_ti; }
      issue_rows();
    }
  };
  issue_rows();
#pragma unroll
  for (int s = 0; s < LA; ++s) issue_stage();
  // the weights are needed from here on: one counted wait now (the pieces above stay in flight)
#pragma unroll
  for (int q = 0; q < 2 * KS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fw[q][j]));

  // ---- consumer ----
  f32x4 acc[4][4];        // [pixel tile i: pixels 16 i + (lane & 15) of the half][channel tile j: channels 16 j + 4 (lane >> 4) + reg]
  const int sw = (l15 >> 1) & 7;
  int cslot = 0;
  const int nl = n0w + 16 * (lg & 1) + 8 * (lg >> 1);            // first of my 8 channels in a store of channel-tile pair 0 (after the lane swap)
  float* const carry = reinterpret_cast<float*>(smem + NS * STG) + (wave * 4 + lg) * 32;      // [j][os 4 | oq 4] of the first half (lanes l15 == 0)

  auto epilogue = [&](int m0h, int h) {        // rows m0h .. m0h + 63 (half h of the 128-row tile m0h - 64 h)
    const int bnd = STATS ? ((m0h - BH * h) / a.stat_Mg + 1) * a.stat_Mg : 0x7fffffff;     // rows >= bnd: next statistics group of the SLAB's first row (stage 2 sums them)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0h + 16 * i + l15;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        unsigned lo0 = ws_pack2(acc[i][2 * pr][0], acc[i][2 * pr][1]), hi0 = ws_pack2(acc[i][2 * pr][2], acc[i][2 * pr][3]);
        unsigned lo1 = ws_pack2(acc[i][2 * pr + 1][0], acc[i][2 * pr + 1][1]), hi1 = ws_pack2(acc[i][2 * pr + 1][2], acc[i][2 * pr + 1][3]);
        ws_swap16(lo0, lo1);
        ws_swap16(hi0, hi1);
        ws_u32x4 vv = {lo0, hi0, lo1, hi1};
        __builtin_amdgcn_raw_buffer_store_b128(vv, rs_d, (int)(m < a.M ? ((unsigned)m * (unsigned)a.ldd + (unsigned)(nl + 32 * pr)) * 2u : WS_OOB), 0, WS_STORE_AUX);
      }
    }
    if (STATS) {
      // sum and sum of squares of the bf16-ROUNDED outputs (what the batch norm will read), two values per instruction; the row test only in a
      // half that straddles a statistics-group boundary.  Half 0 parks its 16-lane sums in LDS, half 1 adds them and stores the 128-row slab.
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(a.stats, 0, (int)a.stat_bytes, 0x00020000);
      const unsigned base = (unsigned)(m0h >> 7) * 2u * (unsigned)a.Cd * 4u;
      const bool whole = m0h + BH <= bnd;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ws_f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
        auto accum = [&](bool test) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 t = acc[i][j];
            asm volatile("" : "+v"(t));       // opaque: otherwise the packed values of the store loop stay alive (CSE) across the epilogue
            const unsigned lo = ws_pack2(t[0], t[1]), hi = ws_pack2(t[2], t[3]);
            ws_f32x2 v01 = {ws_lo(lo), ws_hi(lo)}, v23 = {ws_lo(hi), ws_hi(hi)};
            if (test && !(m0h + 16 * i + l15 < bnd)) { v01 = ws_f32x2{0.f, 0.f}; v23 = ws_f32x2{0.f, 0.f}; }   // (rows >= M hold zeros already)
            s01 += v01; s23 += v23;
            q01 += v01 * v01; q23 += v23 * v23;
          }
        };
        if (whole) accum(false);
        else accum(true);
        ws_f32x4 os = {ws_row16_sum(s01[0]), ws_row16_sum(s01[1]), ws_row16_sum(s23[0]), ws_row16_sum(s23[1])};
        ws_f32x4 oq = {ws_row16_sum(q01[0]), ws_row16_sum(q01[1]), ws_row16_sum(q23[0]), ws_row16_sum(q23[1])};
        if (h == 0) {
          if (l15 == 0) {
            *reinterpret_cast<ws_f32x4*>(carry + j * 8) = os;
            *reinterpret_cast<ws_f32x4*>(carry + j * 8 + 4) = oq;
          }
        } else {
          const ws_f32x4 cs = *reinterpret_cast<const ws_f32x4*>(carry + j * 8), cq = *reinterpret_cast<const ws_f32x4*>(carry + j * 8 + 4);
          os = cs + os;
          oq = cq + oq;
          const int n = n0w + 16 * j + 4 * lg;
          const bool lane_ok = l15 == 0;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, os), rs_s, (int)(lane_ok ? base + (unsigned)n * 4u : WS_OOB), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, oq), rs_s, (int)(lane_ok ? base + (unsigned)(a.Cd + n) * 4u : WS_OOB), 0, 0);
        }
      }
    }
  };
  auto wait_stage = [&](int e) {               // e = half tiles finished so far
    if constexpr (NT == 2) {
      if (e >= 2) ws_wait_vm<NPC * (LA - 1) + 2 * NST>();
      else if (e == 1) ws_wait_vm<NPC * (LA - 1) + NST>();
      else ws_wait_vm<NPC * (LA - 1)>();
    } else if constexpr (NT == 4) {
      if (e >= 4) ws_wait_vm<NPC * (LA - 1) + 4 * NST>();
      else if (e == 3) ws_wait_vm<NPC * (LA - 1) + 3 * NST>();
      else if (e == 2) ws_wait_vm<NPC * (LA - 1) + 2 * NST>();
      else if (e == 1) ws_wait_vm<NPC * (LA - 1) + NST>();
      else ws_wait_vm<NPC * (LA - 1)>();
    } else {
      if (e >= 6) ws_wait_vm<NPC * (LA - 1) + 6 * NST>();
      else if (e == 5) ws_wait_vm<NPC * (LA - 1) + 5 * NST>();
      else if (e == 4) ws_wait_vm<NPC * (LA - 1) + 4 * NST>();
      else if (e == 3) ws_wait_vm<NPC * (LA - 1) + 3 * NST>();
      else if (e == 2) ws_wait_vm<NPC * (LA - 1) + 2 * NST>();
      else if (e == 1) ws_wait_vm<NPC * (LA - 1) + NST>();
      else ws_wait_vm<NPC * (LA - 1)>();
    }
  };

  for (int ti = 0; ti < nmy; ++ti) {
    const int m0 = (t0 + ti * tstep) * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        __builtin_amdgcn_sched_barrier(0);
        wait_stage(2 * ti + h);                 // my two pieces of this stage have landed
        __builtin_amdgcn_s_barrier();           // everybody's pieces have landed; everybody is done reading the previous stage
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue_stage();                          // stage + LA into the slot read one stage ago
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ab = smem + cslot * STG + l15 * 128;
        // all fragment reads of a batch first (back to back), then its MFMAs back to back.  K = 256 holds 128 registers of weights: one
        // 32-channel half of the stage at a time (4 reads, 16 MFMAs); the narrower instances read the whole stage (8 reads, 32 MFMAs)
        constexpr int HB = KS == 4 ? 1 : 2;
#pragma unroll
        for (int hb = 0; hb < 2; hb += HB) {
          bf16x8 fa[HB][4];
#pragma unroll
          for (int hh = 0; hh < HB; ++hh)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[hh][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048 + (((4 * (hb + hh) + lg) ^ sw) << 4));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int hh = 0; hh < HB; ++hh) {
            const int q = 2 * k + hb + hh;
            if (q == 0) {
              const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[hh][i], z, 0, 0, 0);
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[q][j], fa[hh][i], acc[i][j], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cslot = cslot == NS - 1 ? 0 : cslot + 1;
      }
      __builtin_amdgcn_sched_barrier(0);
      epilogue(m0 + BH * h, h);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ghost pieces (stages past my last tile) must have landed before the LDS is released
}


// ---- dispatch (conv_ws.hip, in front of css_launch_conv_ws) ----
static int g_ws4 = -1;         // -1: read CSS_WS4 on first use (0: conv_ws_kernel for every shape - the round-4 kernel, A/B and parity tests)
static int g_ws4_stagger = 0;
void css_conv_ws4_set(int on, int stagger) { g_ws4 = on ? 1 : 0; if (stagger >= 0) g_ws4_stagger = stagger; }     // (scripts/ws_bench.hip)
static bool ws4_on() {
  if (g_ws4 < 0) {
    const char* e = getenv("CSS_WS4");
    g_ws4 = (e && e[0] == '0') ? 0 : 1;
    const char* s = getenv("CSS_WS4_STAGGER");
    g_ws4_stagger = s ? atoi(s) : 1536;        // cycles the second workgroup of a CU waits before its first LDS-DMA piece
  }
  return g_ws4 != 0;
}
template <int KS>
static void launch_ws4(const ConvArgs& a, int n_cu, hipStream_t st) {
  const dim3 g(2 * n_cu), b(256);
  if (a.stats) hipLaunchKernelGGL((conv_ws4_kernel<KS, true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_ws4_kernel<KS, false>), g, b, 0, st, a);
}

// ---- routing (first lines of css_launch_conv_ws after the byte counts) ----
  if (!a.addend && a.Cs <= 256 && ws4_on()) {      // two four-wave workgroups per CU (conv_ws4_kernel)
    a.ws_stagger = g_ws4_stagger;
    if (a.Cs == 256) launch_ws4<4>(a, n_cu, st);
    else if (a.Cs == 128) launch_ws4<2>(a, n_cu, st);
    else launch_ws4<1>(a, n_cu, st);
    return;
  }
