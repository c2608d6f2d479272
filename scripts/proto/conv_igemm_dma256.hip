// ARCHIVED (round 4): conv_igemm_dma256_kernel, the first 256x256-tile LDS-DMA forward / dgrad kernel (round 1-2), removed
// from css_amd/csrc/conv.hip when conv_igemm_p8_kernel / conv_igemm_pp_kernel covered every shape it served (the remaining ones - strided
// 3x3, bias - run on conv_igemm_dma_kernel<256,128>).  Kept as the A/B reference: paste behind conv_igemm_dma_kernel in conv.hip to build it
// (it uses that file's Mma<>, bload16, dma16, store_wave_tile).  Not part of the product, not compiled.
#if 0
// --------------------------------------------------------------------------
// 256x256x32 variant of the LDS-DMA kernel for Cout >= 256: 8 waves as 2 (pixels) x 4 (channels), 128x64 outputs per wave,
// FOUR 32 KiB LDS stages (three K tiles = 96 KiB in flight per CU, as in the 256x128x64 kernel).  Per FLOP it moves 2/3 of
// the global->LDS bytes and 3/4 of the LDS->register bytes of that kernel - the two walls its ablations showed (loads-only
// and compute-only ceilings) - at the price of a coarser tile grid (the launcher sends leftovers to the 128x128 kernel).
//  * LDS rows are 32 bf16 = 64 B, unpadded; 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3) (conflict-free
//    ds_read_b128: 16 consecutive rows of one chunk column cover all 16 slots of a 256-byte bank row).
//  * thread t owns position t & 3 of rows (t >> 2) + 128 i; a DMA wave-instruction covers 16 rows.
//  * sync per K tile: s_waitcnt vmcnt(8) -> s_barrier -> issue tile kt+3 -> 16 MFMAs per wave on tile kt.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_igemm_dma256_kernel(const ConvArgs a) {
  using T = bf16_t;
  constexpr int BM = 256, BN = 256, BK = 32, VEC = 8, NST = 4;
  constexpr int A_IT = 2, B_IT = 2;                    // DMA wave-instructions per thread per stage (rows t>>2 + 128 i)
  constexpr int ROWB = BK * 2;                         // 64 bytes per LDS row
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int WTM = 128, WTN = 64, TM = 4, TN = 2, CSTR = WTN + VEC;
  static_assert(8 * 64 * CSTR * 2 <= NST * ST_BYTES, "epilogue staging (one 64-row half per wave) fits");
  // one LDS object (see conv_igemm_dma_kernel); tail: per wave [arrival counter + pad | slot | slot] for the BN statistics
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES + 8 * 12 * 64 * 4];
  float* sstat = reinterpret_cast<float*>(smem + NST * ST_BYTES);
  if (a.stats && threadIdx.x < 8) reinterpret_cast<int*>(sstat)[threadIdx.x * 12 * 64] = 0;   // ordered by the main loop's barriers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nt_n = (a.Cd + BN - 1) / BN;
  const int ntiles = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int m0 = a.m_begin + (logical / nt_n) * BM, n0 = (logical % nt_n) * BN;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);

  const int prow = tid >> 2;                                   // tile row of this thread's chunks (+ 128 i)
  const int cch = (tid & 3) ^ ((tid >> 4) & 3);                // source chunk (8 channels) this thread fetches: p ^ ((r>>2)&3)

  int a_base[A_IT], a_h[A_IT], a_w[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + prow + i * 128;
    if (m < a.M) {
      const int hw = a.Hd * a.Wd;
      const int n_img = m / hw;
      const int rem = m - n_img * hw;
      const int hd = rem / a.Wd;
      const int wd = rem - hd * a.Wd;
      a_base[i] = n_img * a.Hs * a.Ws;
      if (a.mode == 0) {
        a_h[i] = hd * a.stride - a.pad;
        a_w[i] = wd * a.stride - a.pad;
      } else {
        a_h[i] = hd + a.pad;
        a_w[i] = wd + a.pad;
      }
    } else {
      a_base[i] = 0;
      a_h[i] = -0x40000000;
      a_w[i] = -0x40000000;
    }
  }
  unsigned b_off[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + prow + i * 128;
    b_off[i] = n < a.Cd ? (unsigned)n * (unsigned)a.Ktot * 2u : OOB;
  }
  int kc = cch * VEC, tr = 0, ts = 0;
  while (kc >= a.Cs) {
    kc -= a.Cs;
    if (++ts == a.S) { ts = 0; ++tr; }
  }
  int kglob = cch * VEC;

  // all-padding kernel rows are skipped, block-uniformly (see conv_igemm_dma_kernel)
  unsigned tr_mask = 0xffffffffu;
  int nk = (a.Ktot + BK - 1) / BK;
  if (a.Cs % BK == 0 && a.R > 1 && a.R < 32) {
    const int hw = a.Hd * a.Wd;
    const int mlast = min(m0 + BM, a.M) - 1;
    const int i0 = m0 / hw, i1 = mlast / hw;
    const int h0 = (m0 - i0 * hw) / a.Wd, h1 = (mlast - i1 * hw) / a.Wd;
    if (i1 - i0 <= 1) {
      const int alo = h0, ahi = i1 == i0 ? h1 : a.Hd - 1;
      const int blo = i1 == i0 ? h0 : 0, bhi = h1;
      tr_mask = 0;
      int cnt = 0;
      for (int r = 0; r < a.R; ++r) {
        bool v;
        if (a.mode == 0) {
          const int o = r * a.dil - a.pad;
          v = (alo * a.stride + o <= a.Hs - 1 && ahi * a.stride + o >= 0) || (blo * a.stride + o <= a.Hs - 1 && bhi * a.stride + o >= 0);
        } else {
          const int o = a.pad - r * a.dil;
          v = (alo + o <= (a.Hs - 1) * a.stride && ahi + o >= 0) || (blo + o <= (a.Hs - 1) * a.stride && bhi + o >= 0);
        }
        if (v) { tr_mask |= 1u << r; ++cnt; }
      }
      nk = cnt * a.S * (a.Cs / BK);
      while (tr < a.R && !((tr_mask >> tr) & 1)) { ++tr; kglob += a.S * a.Cs; }
    }
  }

  unsigned offA[A_IT];
  auto tap_offsets = [&]() {
    const bool tap_ok = tr < a.R;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int hs, ws;
      bool ok = tap_ok;
      if (a.mode == 0) {
        hs = a_h[i] + tr * a.dil;
        ws = a_w[i] + ts * a.dil;
      } else {
        const int th = a_h[i] - tr * a.dil, tw = a_w[i] - ts * a.dil;
        ok = ok && th >= 0 && tw >= 0;
        if (a.stride == 2) {
          ok = ok && !((th | tw) & 1);
          hs = th >> 1;
          ws = tw >> 1;
        } else {
          hs = th;
          ws = tw;
        }
      }
      ok = ok && (unsigned)hs < (unsigned)a.Hs && (unsigned)ws < (unsigned)a.Ws;
      const unsigned off = (unsigned)((a_base[i] + hs * a.Ws + ws) * a.lds + kc) * 2u;
      offA[i] = ok ? off : OOB;
    }
  };
  tap_offsets();
  // wave-uniform LDS destinations: instruction i of wave w covers chunks [i*512 + w*64, +64) of the A (or B) image
  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * ST_BYTES + wave * 1024;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) dma16(rs_a, sa + i * 8192, offA[i]);
    const unsigned kb = kglob < a.Ktot ? (unsigned)kglob * 2u : OOB;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma16(rs_b, sb + i * 8192, (b_off[i] | kb) & OOB ? OOB : b_off[i] + kb);
    kglob += BK;
    kc += BK;
    if (kc >= a.Cs) {
      do {
        kc -= a.Cs;
        if (++ts == a.S) {
          ts = 0;
          ++tr;
          while (tr < a.R && !((tr_mask >> tr) & 1)) { ++tr; kglob += a.S * a.Cs; }
        }
      } while (kc >= a.Cs);
      tap_offsets();
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) offA[i] += (offA[i] & OOB) ? 0u : (unsigned)(BK * 2);
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int xr = (l31 >> 2) & 3;
  int koff[2];                                         // swizzled byte offset of k-step ks inside a row
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (((2 * ks + lh) ^ xr) << 4);
  const int a_row = (wm * WTM + l31) * ROWB, b_row = A_BYTES + (wn * WTN + l31) * ROWB;
  auto compute = [&](int stage) {
    const unsigned char* sbase = smem + stage * ST_BYTES;
    bf16x8 fw[2][TN], fa[2][TM];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const bf16x8*>(sbase + b_row + i * 32 * ROWB + koff[ks]);
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[ks][j] = *reinterpret_cast<const bf16x8*>(sbase + a_row + j * 32 * ROWB + koff[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  issue(0);
  issue(1);
  issue(2);
  int st_c = 0, st_i = 3;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(st_i);                       // tile kt+3 (past the end: all-OOB = zeros into a free stage)
    compute(st_c);
    st_c = (st_c + 1) & 3;
    st_i = (st_i + 1) & 3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ghost DMAs must land before the stages are reused below
  __builtin_amdgcn_s_barrier();

  // ---- epilogue, one 64-pixel half of the wave tile at a time: accumulators -> wave-private LDS -> 16-byte row stores
  T* Cw = reinterpret_cast<T*>(smem) + wave * (64 * CSTR);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = i * 32 + 8 * q + 4 * lh;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn * WTN + nl + e;
            bv[e] = n < a.Cd ? a.bias[n] : 0.f;
          }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          T* p = Cw + (jj * 32 + l31) * CSTR + nl;
          union { T e[4]; uint2 u2; } pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk.e[e] = (T)(acc[i][2 * half + jj][4 * q + e] + bv[e]);
          *reinterpret_cast<uint2*>(p) = pk.u2;
        }
      }
    }
    __syncthreads();
    store_wave_tile<T, 64, WTN, CSTR, BN, true>(a, Cw, m0 + wm * WTM + half * 64, n0 + wn * WTN, half, lane, sstat + wave * 12 * WTN);
    __syncthreads();
  }
}

#endif
