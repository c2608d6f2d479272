// ARCHIVED (round 4): conv_wgrad_dma256_kernel<STAG>, the one-barrier-per-step 256x256 weight-gradient kernels (lockstep /
// half-step stagger) that conv_wgrad_p8_kernel replaced in round 3 (profiles/r03_wgrad_p8.txt).  Kept as the A/B reference: paste in front of
// conv_wgrad_p8_kernel in css_amd/csrc/conv_wgrad.hip to build it (it uses that file's dma16_lds, raw_rsrc, wg_frag_sw).  Not compiled.
#if 0
// STAG (r03): the two waves of a SIMD (wn = 0 / 1) run the same program with one barrier per step, i.e. in lockstep: both read their
// fragments, then both queue for the SIMD's one matrix pipe.  With STAG the second cout half defers the MFMAs of each step's second
// 16-pixel half by one step (its fragments stay in registers across the barrier): after a barrier it multiplies while the first half
// reads, then reads while the first half multiplies (MI355X_MICROARCH.md, Two waves per SIMD, item 9).  Same MFMA order per
// accumulator, so the result is bit-identical.
template <bool STAG>
__global__ __launch_bounds__(512) void conv_wgrad_dma256_kernel(const WgradArgs a) {
  constexpr int BN_ = 256, BKC = 256, BP = 32, NST = 4;
  constexpr int T_BYTES = BP * 512, ST_BYTES = 2 * T_BYTES;     // Y tile then X tile
  constexpr int WTN = 128, WTK = 64, TN = 4, TK = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  // XCD-aware order as in conv_wgrad_kernel: all tiles of one pixel slice run on one XCD
  const int per_z = a.tiles_k * a.tiles_n;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
  const int zz = (j8 / per_z) * 8 + xcd, t = j8 % per_z;
  if (zz >= a.splits) return;
  const int k0 = (t % a.tiles_k) * BKC, n0 = (t / a.tiles_k) * BN_;
  const int m_begin = zz * a.m_per_split;
  const int m_end = min(a.M, m_begin + a.m_per_split);
  const int nit = (m_end - m_begin + BP - 1) / BP;
  if (nit <= 0) return;

  const int prow = tid >> 5;                                             // tile row of this thread's chunks (+ 16 i)
  const int schunk = ((((tid & 31) >> 2) ^ (prow & 3)) << 2) | (tid & 3);  // source 16-byte column of those chunks
  const int kcol = k0 + schunk * 8;
  const bool k_ok = kcol < a.Ktot;
  const int tap = k_ok ? kcol / a.Cs : 0;
  const int xc = k_ok ? kcol - tap * a.Cs : 0;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const int dh = tr * a.dil - a.pad, dw_ = ts * a.dil - a.pad;
  const int ncol = n0 + schunk * 8;
  const bool n_ok = ncol < a.Cd;

  const u32x4 rs_x = raw_rsrc(a.x, a.x_bytes), rs_y = raw_rsrc(a.dy, a.dy_bytes);
  // Issue side.  Every thread walks two pixel rows (prow, prow + 16) through the slice in steps of BP = 32 pixels; the step is a
  // mixed-radix addition on (image, hd, wd) with one carry per digit, and the byte offsets into x and dy move by constants picked
  // by the carries - a handful of full-rate VALU instructions per row where the first version re-derived (image, hd, wd) with two
  // magic divisions, 64-bit multiply-adds and three 32-bit multiplies per row and step (~110 instructions per step and wave next to
  // 16 MFMAs).  Rows >= m_end, padding taps and tail columns land as zeros (out-of-range offset).
  const int q_w = (int)fdiv((uint32_t)BP, a.fd_w), d_w = BP - q_w * a.Wd;            // BP = (d_n * Hd + d_h) * Wd + d_w
  const int d_n = (int)fdiv((uint32_t)BP, a.fd_hw), d_h = q_w - d_n * a.Hd;
  const int xrow = a.ldx * 2;                                                       // bytes per source pixel
  const int sx_w = a.stride * xrow, sx_h = a.stride * a.Ws * xrow, sx_n = a.Hs * a.Ws * xrow;
  const int D0 = d_n * sx_n + d_h * sx_h + d_w * sx_w;                               // no carry
  const int Dw = sx_h - a.Wd * sx_w, Dh = sx_n - a.Hd * sx_h;                        // extra when wd / hd wrap
  const int ystep = BP * a.ldy * 2;
  int r_m[2], r_hs[2], r_ws[2];          // row, source coordinates of my tap (may be outside the image: padding)
  unsigned r_xo[2], r_yo[2];             // byte offsets of my 16-byte chunk in x and dy
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_begin + prow + i * 16;
    const uint32_t n_img = fdiv((uint32_t)m, a.fd_hw);
    const uint32_t rem = (uint32_t)m - n_img * a.fd_hw.d;
    const uint32_t hd = fdiv(rem, a.fd_w);
    const uint32_t wd = rem - hd * a.fd_w.d;
    r_m[i] = m;
    r_hs[i] = (int)hd * a.stride + dh;
    r_ws[i] = (int)wd * a.stride + dw_;
    r_xo[i] = (unsigned)(((int)n_img * a.Hs * a.Ws + r_hs[i] * a.Ws + r_ws[i]) * a.ldx + xc) * 2u;
    r_yo[i] = (unsigned)(m * a.ldy + ncol) * 2u;
  }
  const int hs_hi = (a.Hd - 1) * a.stride + dh, ws_hi = (a.Wd - 1) * a.stride + dw_;   // source coordinate of the last output row / column
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void*)smem) + (unsigned)wave * 1024u;
  auto issue = [&](int stage) {          // the next pixel block of the slice -> stage; advances the walk
    const unsigned sy = lds0 + (unsigned)stage * ST_BYTES, sx = sy + T_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16_lds(rs_y, sy + i * 8192, (n_ok && r_m[i] < m_end) ? r_yo[i] : OOB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = k_ok && r_m[i] < m_end && (unsigned)r_hs[i] < (unsigned)a.Hs && (unsigned)r_ws[i] < (unsigned)a.Ws;
      dma16_lds(rs_x, sx + i * 8192, ok ? r_xo[i] : OOB);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r_m[i] += BP;
      r_yo[i] += (unsigned)ystep;
      int ws = r_ws[i] + d_w * a.stride, hs = r_hs[i] + d_h * a.stride;
      int dx = D0;
      const bool cw = ws > ws_hi;
      ws -= cw ? a.Wd * a.stride : 0;
      hs += cw ? a.stride : 0;
      dx += cw ? Dw : 0;
      const bool ch = hs > hs_hi;
      hs -= ch ? a.Hd * a.stride : 0;
      dx += ch ? Dh : 0;
      r_ws[i] = ws;
      r_hs[i] = hs;
      r_xo[i] += (unsigned)dx;
    }
  };

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 fy[2][TN], fx[2][TK];          // fragments of the two 16-pixel halves of a stage
  auto read_frags = [&](int stage, int ks) {
    const unsigned char* Yb = smem + stage * ST_BYTES;
    const unsigned char* Xb = Yb + T_BYTES;
#pragma unroll
    for (int i = 0; i < TN; ++i) fy[ks][i] = wg_frag_sw(Yb, ks * 16, wn * WTN + i * 32, lane);
#pragma unroll
    for (int j = 0; j < TK; ++j) fx[ks][j] = wg_frag_sw(Xb, ks * 16, wk * WTK + j * 32, lane);
  };
  auto mma = [&](int ks) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[ks][i], fx[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  issue(0);
  issue(1);
  issue(2);
  int st_c = 0, st_i = 3;
  if (STAG && wn == 1) {
    for (int it = 0; it < nit; ++it) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (it > 0) mma(1);                 // second half of the previous step (fragments kept across the barrier)
      __builtin_amdgcn_sched_barrier(0);
      read_frags(st_c, 0);
      read_frags(st_c, 1);                // (must be complete before the next barrier: the stage is refilled after it)
      __builtin_amdgcn_sched_barrier(0);
      issue(st_i);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      st_c = (st_c + 1) & 3;
      st_i = (st_i + 1) & 3;
    }
    mma(1);
  } else
  for (int it = 0; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // fragment reads first, then the LDS-DMA issue and the walk's ALU work while the reads are in flight, then the MFMAs (issue ahead
    // of the reads: +2 % time; a software pipeline inside the wave - reads of one 16-pixel half under the MFMAs of the other, with the
    // barrier between them - +4 %: the two waves of a SIMD already cover each other, profiles/r02_conv_ablation.txt section 6)
    read_frags(st_c, 0);
    read_frags(st_c, 1);
    __builtin_amdgcn_sched_barrier(0);
    issue(st_i);                        // step it+3 (past m_end: all-OOB = zeros into a free stage)
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    mma(1);
    st_c = (st_c + 1) & 3;
    st_i = (st_i + 1) & 3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ghost DMAs must have landed before the workgroup's LDS is released
  // D[row -> n][col -> k]: one 128-byte fp32 segment per half-wave per accumulator register
  const int l31 = lane & 31, lh = lane >> 5;
  if (a.ws) {
    // partial tile of this pixel slice -> its own 256 x 256 fp32 slab of the workspace with PLAIN stores (wgrad_slab_reduce_kernel
    // adds the slices up in a fixed order): fp32 atomics run at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md) - 504 workgroups x
    // 256 KiB took longer than the MFMAs of a layer-3 weight gradient - plain stores of the same shape at ~6 TB/s
    float* slab = a.ws + ((size_t)t * a.splits + zz) * (256 * 256);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int kl = wk * WTK + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nl = wn * WTN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[nl * 256 + kl] = acc[i][j][r];
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

#endif
