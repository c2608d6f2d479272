// Implicit-GEMM convolution for NHWC activations on gfx950 (CDNA4):
//   conv_igemm_kernel     : forward and data-gradient, register-staged (fp32 parity path, channel counts that are not whole K tiles)
//   conv_igemm_dma_kernel : the same through LDS-DMA (narrow layers, leftover rows of the persistent kernels)
//   css_launch_conv       : the ONE dispatch over these, conv_ws.hip (short-K 1x1), conv_p8.hip / conv_pp.hip (persistent 256x256 tiles)
// (weight gradient: conv_wgrad.hip)
//
// Replaces the cuDNN convolutions the reference reaches through nn.Conv2d in
//   generalframeworks/networks/resnet.py:24-40,119-139 (Bottleneck 1x1 / 3x3 dilated),
//   generalframeworks/networks/deeplabv3/aspp.py:17-72 (ASPP 1x1 + dilated 3x3),
//   generalframeworks/networks/deeplabv3/deeplabv3.py:115-133,151-169 (decoder heads).
//
// Data layout: activations [N][H][W][ld] (channels innermost, ld >= C), weights
// [Cout][R][S][Cin] (= torch channels_last physical layout of an nn.Conv2d weight).
// GEMM view: Out[M = N*Ho*Wo][Cout] = A[M][K = R*S*Cin] * W[Cout][K]^T, both operands
// K-contiguous, so every MFMA fragment is one 16-byte LDS read.
//
// MFMA: v_mfma_f32_32x32x16_bf16 (bf16 in, fp32 accumulate) for the throughput path and
// v_mfma_f32_32x32x2_f32 (exact fp32) for the parity path. The weight tile is the MFMA
// "A" operand and the activation tile the "B" operand, so a lane owns one output pixel and
// four consecutive output channels per accumulator quad: the epilogue packs those and goes
// through LDS to full 16-byte row-contiguous stores.
#include "common.h"
#include "launchers.h"
#include <cstdlib>

// --------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KS = 16;  // reduction elements consumed per MFMA
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag load(const bf16_t* row_k, int h) {
    return *reinterpret_cast<const frag*>(row_k + h * 8);
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KS = 2;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* row_k, int h) { return row_k[h]; }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// Sum of v over the lanes {l : l % CV == lane % CV} of the wave, result in every lane.  row_ror DPP adds inside 16-lane
// rows, then v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows.  The swaps are inline asm: the
// __builtin_amdgcn_permlane{16,32}_swap builtins of this toolchain return the same register for both results.
template <int CV>
__device__ __forceinline__ float lanes_sum(float v) {
  static_assert(CV == 4 || CV == 8 || CV == 16, "lane groups");
  if constexpr (CV <= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  if constexpr (CV <= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  float w = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  v += w;
  w = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return v + w;
}

// --------------------------------------------------------------------------
// Shared tail of the epilogue: one wave moves its staged [WTM][WTN] tile from LDS to global memory in 16-byte row
// pieces.  Fused into that pass, per launch option:
//  * a.addend  - the tensor added element-wise before the store (dgrad: the residual branch's gradient, which autograd
//                would otherwise add in a separate pass over both tensors; same rounding: bf16 + bf16 in fp32, rounded once)
//  * a.stats   - batch-norm statistics of exactly the values stored (the bf16-rounded ones bn_stats would read back):
//                per-channel (sum, sum of squares) of every 128-row SLAB of the output, written to
//                stats[slab][2][Cd] (fp32, plain stores - one workgroup owns a slab).  Statistics groups (forward passes
//                batched along M, stat_Mg rows each) need not be slab-aligned: a slab's row holds only the rows of the group
//                its FIRST row belongs to; the (< 128) rows past a group boundary are summed from the stored tensor by
//                css_bn_reduce_finalize_slabs (bn.hip), which adds everything per group in fp64.
// sstat: LDS area of this wave's (slab, wave column) pair: an arrival counter (zeroed at kernel start) and one slot
// [2 parts][2 stats][WTN] per wave row (part 1 = rows of the next group).  wml: which of the slab's two wave rows this is.
// --------------------------------------------------------------------------
template <typename T, int WTM, int WTN, int CSTR, int BN, bool STATS>
__device__ __forceinline__ void store_wave_tile(const ConvArgs& a, const T* Cw, int mrow0, int n0w, int wml, int lane, float* sstat) {
  constexpr int VEC = 16 / sizeof(T), CV = WTN / VEC, RSTEP = 64 / CV;
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst);
  const T* __restrict__ addp = reinterpret_cast<const T*>(a.addend);
  const bool vec_ok = (a.ldd % VEC) == 0 && ((reinterpret_cast<uintptr_t>(a.dst) & 15) == 0) &&
                      (!addp || ((a.ld_add % VEC) == 0 && (reinterpret_cast<uintptr_t>(a.addend) & 15) == 0));
  const int cv = lane % CV, r0 = lane / CV;
  const int n = n0w + cv * VEC;
  const bool want = STATS && a.stats != nullptr;
  float s[VEC], q[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = q[e] = 0.f;
  static_assert(WTM % RSTEP == 0, "static trip count");
  constexpr int NIT = WTM / RSTEP;
  auto accumulate = [&](const Vec16<T>& v) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float f = v.f(e);
      s[e] += f;
      q[e] += f * f;
    }
  };
  if (vec_ok && n0w + WTN <= a.Cd && mrow0 + WTM <= a.M) {
    // interior wave tile (wave-uniform test): branch-free, every LDS read / addend load issued before the first store
    Vec16<T> v[NIT], r[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it].load(Cw + (r0 + it * RSTEP) * CSTR + cv * VEC);
    if (addp) {
      unsigned mk[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        r[it].load(addp + (size_t)(mrow0 + r0 + it * RSTEP) * a.ld_add + n);
        mk[it] = a.add_mask ? a.add_mask[(size_t)(mrow0 + r0 + it * RSTEP) * (a.Cd / VEC) + n / VEC] : 0xFFFFu;
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[it].set(e, v[it].f(e) + keep_if_bit(r[it].f(e), mk[it], e));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it].store(dst + (size_t)(mrow0 + r0 + it * RSTEP) * a.ldd + n);
    if (want) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) accumulate(v[it]);
    }
  } else {
    // edge tiles: M / Cout tails, unaligned leading dimensions
    for (int it = 0; it < NIT; ++it) {
      const int row = r0 + it * RSTEP;
      const int m = mrow0 + row;
      if (m >= a.M || n >= a.Cd) continue;
      Vec16<T> v;
      v.load(Cw + row * CSTR + cv * VEC);
      T* o = dst + (size_t)m * a.ldd + n;
      if (vec_ok && n + VEC <= a.Cd) {
        if (addp) {
          Vec16<T> r;
          r.load(addp + (size_t)m * a.ld_add + n);
          const unsigned mk = a.add_mask ? a.add_mask[(size_t)m * (a.Cd / VEC) + n / VEC] : 0xFFFFu;
#pragma unroll
          for (int e = 0; e < VEC; ++e) v.set(e, v.f(e) + keep_if_bit(r.f(e), mk, e));
        }
        v.store(o);
      } else {
        // statically indexed on purpose: a run-time index into the vector sends it through a stack object (which the
        // compiler places in LDS) on EVERY path of this loop
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (n + e < a.Cd) {
            if (addp) v.set(e, v.f(e) + ElemT<T>::to_f(addp[(size_t)m * a.ld_add + n + e]));      // (no mask here: css_conv2d_dgrad_add_masked requires Cd % VEC == 0)
            o[e] = v.e[e];
          }
        }
      }
      if (want) accumulate(v);
    }
  }
  if constexpr (STATS) {
    if (want) {
      const int bnd = ((mrow0 & ~127) / a.stat_Mg + 1) * a.stat_Mg;   // rows >= bnd belong to the next statistics group
      const bool straddle = mrow0 < bnd && mrow0 + WTM > bnd && bnd < a.M;   // wave-uniform, at most one wave row per group
      // sum over the 64/CV lanes that own the same channel vector (lane = row*CV + cv): VALU only (DPP row rotates, then the
      // gfx950 row-swap instructions) - a ds_bpermute butterfly plus LDS atomics cost ~4 us per tile here
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s[e] = lanes_sum<CV>(s[e]);
        q[e] = lanes_sum<CV>(q[e]);
      }
      float s1[VEC], q1[VEC];   // the share of the NEXT group
      if (straddle) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s1[e] = q1[e] = 0.f;
        for (int row = r0; row < WTM; row += RSTEP) {
          const int m = mrow0 + row;
          if (m < bnd || m >= a.M || n >= a.Cd) continue;
          Vec16<T> v;
          v.load(Cw + row * CSTR + cv * VEC);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float f = v.f(e);
            s1[e] += f;
            q1[e] += f * f;
          }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          s1[e] = lanes_sum<CV>(s1[e]);
          q1[e] = lanes_sum<CV>(q1[e]);
        }
      } else {
        const float all_next = mrow0 >= bnd ? 1.f : 0.f;   // the whole wave tile lies in the next group (or none of it)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s1[e] = all_next * s[e]; q1[e] = all_next * q[e]; }
      }
      // Two wave rows make a slab.  The first of the pair to get here parks its sums in its LDS slot; the second adds them
      // to its own and writes the slab's row(s) of a.stats: no block-wide barrier, no second pass.
      float* mine = sstat + 4 * WTN;            // sstat = this pair's area: [counter .. pad][slot 0][slot 1]; slot = [4][WTN]
      float* slot = mine + (wml ? 4 * WTN : 0);
      if (r0 == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          slot[0 * WTN + cv * VEC + e] = s[e] - s1[e];
          slot[1 * WTN + cv * VEC + e] = q[e] - q1[e];
          slot[2 * WTN + cv * VEC + e] = s1[e];
          slot[3 * WTN + cv * VEC + e] = q1[e];
        }
      }
      int second = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // slot stores stay ahead of the arrival count
      if (lane == 0) second = atomicAdd(reinterpret_cast<int*>(sstat), 1);
      second = __builtin_amdgcn_readfirstlane(second);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (second && (mrow0 & ~127) < a.M) {
        const float* other = mine + (wml ? 0 : 4 * WTN);
        const int row0 = mrow0 & ~127;
        for (int i = lane; i < 4 * WTN; i += 64) {
          const int col = i % WTN, stat = (i / WTN) & 1, part = i / (2 * WTN);
          const int nn = n0w + col;
          if (nn >= a.Cd) continue;
          // part 0 only: the rows of a slab past a statistics-group boundary are summed from the stored tensor by stage 2
          if (part == 0) a.stats[((size_t)(row0 >> 7) * 2 + stat) * a.Cd + nn] = slot[i] + other[i];
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int BK, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  using MT = Mma<T>;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int KV = BK / VEC;       // 16-B vectors per tile row
  constexpr int STR = BK + VEC;      // LDS row stride in elements (pad = one vector: conflict-free b128 reads)
  constexpr int RPP = 256 / KV;      // tile rows covered per pass of the 256 threads
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int CSTR = WTN + VEC;
  constexpr int AB_ELEMS = 2 * (BM + BN) * STR;
  constexpr int C_ELEMS = 4 * WTM * CSTR;
  constexpr int SM_ELEMS = AB_ELEMS > C_ELEMS ? AB_ELEMS : C_ELEMS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/thread mapping");
  __shared__ __attribute__((aligned(16))) T smem[SM_ELEMS];
  constexpr bool STATS = BM % 128 == 0;                    // slab statistics need whole slabs per workgroup
  // per (slab, wave column) pair of waves: [counter + pad: 4*WTN floats][slot wave row 0: 4*WTN][slot wave row 1: 4*WTN]
  __shared__ float sstat[STATS ? (BM / 128) * WAVES_N * 12 * (BN / WAVES_N) : 1];
  if (STATS && a.stats && threadIdx.x < (BM / 128) * WAVES_N)
    reinterpret_cast<int*>(sstat)[threadIdx.x * 12 * (BN / WAVES_N)] = 0;       // ordered by the main loop's barriers
  T* As = smem;
  T* Bs = smem + 2 * BM * STR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of
  // logical tiles, n-tile fastest: tiles that share an activation panel run on one XCD's L2 at the same time.
  const int nt_n = (a.Cd + BN - 1) / BN;
  const int ntiles = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int m0 = a.m_begin + (logical / nt_n) * BM, n0 = (logical % nt_n) * BN;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);

  const int kv = tid % KV;
  const int prow = tid / KV;

  // ---- per-row bookkeeping for this thread's A_IT activation rows -------
  int a_base[A_IT], a_h[A_IT], a_w[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + prow + i * RPP;
    if (m < a.M) {
      int hw = a.Hd * a.Wd;
      int n_img = m / hw;
      int rem = m - n_img * hw;
      int hd = rem / a.Wd;
      int wd = rem - hd * a.Wd;
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
  unsigned b_off[B_IT];   // byte offset of this thread's weight rows (OOB for rows >= Cout)
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + prow + i * RPP;
    b_off[i] = n < a.Cd ? (unsigned)n * (unsigned)a.Ktot * (unsigned)sizeof(T) : OOB;
  }
  // ---- running (tap, channel) position of this thread's vector column ---
  int kc = kv * VEC, tr = 0, ts = 0;
  while (kc >= a.Cs) {
    kc -= a.Cs;
    if (++ts == a.S) { ts = 0; ++tr; }
  }
  int kglob = kv * VEC;  // global k of this thread's vector (for the weight tile)

  // issue the loads of the NEXT K tile into (ra, rb) and advance the K position; never branches
  auto issue = [&](uint4 (&ra)[A_IT], uint4 (&rb)[B_IT]) {
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
      const unsigned off = (unsigned)((a_base[i] + hs * a.Ws + ws) * a.lds + kc) * (unsigned)sizeof(T);
#ifdef CSS_ABLATE_NOLOAD
      ra[i] = make_uint4(off, ok, 0, 0);
#else
      ra[i] = bload16(rs_a, ok ? off : OOB);
#endif
    }
    const unsigned kb = kglob < a.Ktot ? (unsigned)kglob * (unsigned)sizeof(T) : OOB;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
#ifdef CSS_ABLATE_NOLOAD
      rb[i] = make_uint4(b_off[i], kb, 0, 0);
#else
      rb[i] = bload16(rs_b, (b_off[i] | kb) & OOB ? OOB : b_off[i] + kb);
#endif
    kglob += BK;
    kc += BK;
    while (kc >= a.Cs) {
      kc -= a.Cs;
      if (++ts == a.S) { ts = 0; ++tr; }
    }
  };
  auto store_tiles = [&](const uint4 (&ra)[A_IT], const uint4 (&rb)[B_IT], int buf) {
    T* Ab = As + buf * BM * STR;
    T* Bb = Bs + buf * BN * STR;
#ifdef CSS_ABLATE_NOSTORE
#pragma unroll
    for (int i = 0; i < A_IT; ++i) asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
#pragma unroll
    for (int i = 0; i < B_IT; ++i) asm volatile("" ::"v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
#else
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      *reinterpret_cast<uint4*>(Ab + (prow + i * RPP) * STR + kv * VEC) = ra[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      *reinterpret_cast<uint4*>(Bb + (prow + i * RPP) * STR + kv * VEC) = rb[i];
#endif
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  auto compute = [&](int cur) {
    const T* Ab = As + cur * BM * STR + (wm * WTM + l31) * STR;
    const T* Bb = Bs + cur * BN * STR + (wn * WTN + l31) * STR;
#pragma unroll
    for (int ks = 0; ks < BK / MT::KS; ++ks) {
      typename MT::frag fw[TN], fa[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i] = MT::load(Bb + i * 32 * STR + ks * MT::KS, lh);
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = MT::load(Ab + j * 32 * STR + ks * MT::KS, lh);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = MT::mma(fw[i], fa[j], acc[i][j]);
    }
  };

  // ---- main loop: register-staged, two K tiles in flight (tiles past the last one are all-OOB loads = zeros) ----
  const int nk = (a.Ktot + BK - 1) / BK;
  uint4 ra0[A_IT], rb0[B_IT], ra1[A_IT], rb1[B_IT];
  issue(ra0, rb0);
  issue(ra1, rb1);
  store_tiles(ra0, rb0, 0);
  __syncthreads();
  for (int kt = 0;;) {
    issue(ra0, rb0);              // tile kt+2
    compute(0);                   // tile kt
    store_tiles(ra1, rb1, 1);     // tile kt+1
    __syncthreads();
    if (++kt >= nk) break;
    issue(ra1, rb1);              // tile kt+2
    compute(1);
    store_tiles(ra0, rb0, 0);
    __syncthreads();
    if (++kt >= nk) break;
  }

  // ---- epilogue: accumulators -> LDS (wave-private) -> 16-byte row stores
  // D[row -> n][col -> m]: lane owns pixel m = l31, channels 8*q + 4*lh + {0..3} per quad q.
  T* Cw = smem + wave * (WTM * CSTR);
#pragma unroll
  for (int i = 0; i < TN; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = i * 32 + 8 * q + 4 * lh;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int n = n0 + wn * WTN + nl + e;
          bv[e] = n < a.Cd ? a.bias[n] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        T* p = Cw + (j * 32 + l31) * CSTR + nl;
        union { T e[4]; uint2 u2; uint4 u4; } pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.e[e] = ElemT<T>::from_f(acc[i][j][4 * q + e] + bv[e]);
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(p) = pk.u2;
        else *reinterpret_cast<uint4*>(p) = pk.u4;
      }
    }
  }
  __syncthreads();
  static_assert(!STATS || WTM == 64, "two wave rows per slab");
  store_wave_tile<T, WTM, WTN, CSTR, BN, STATS>(a, Cw, m0 + wm * WTM, n0 + wn * WTN, wm & 1, lane,
                                                sstat + (STATS ? ((wm >> 1) * WAVES_N + wn) * 12 * WTN : 0));
}

// --------------------------------------------------------------------------
// bf16 throughput variant: 256x128x64 tile, 8 waves (4x2, 64x64 each), THREE LDS stages filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: global -> LDS without VGPR staging or ds_write), two K tiles in flight.
// The ablation of the register-staged kernel above (profiles/r01_conv_ablation.txt) showed global loads and LDS stores
// each costing ~25 %; this structure removes the stores and halves the weight-tile traffic per FLOP.
//  * LDS image of a stage: rows of 64 bf16 = 128 B, unpadded (a DMA wave-instruction writes 64 lanes x 16 B = 1 KiB
//    = 8 whole rows, lane-linear).  16-byte chunk c of row r is stored at position c ^ ((r >> 1) & 7): the swizzle is
//    applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads (conflict-free: the 16 lanes of
//    a read group then hit 16 distinct 16-byte slots of the 256-byte bank row).
//  * thread t owns chunk position t & 7 of rows (t >> 3) + 64 i: ((r >> 1) & 7) is the same for all of them, so one
//    running (tap, channel) state per thread serves every DMA it issues.
//  * sync per K tile: s_waitcnt vmcnt(6) (this wave's share of tile kt has landed, tile kt+1 may be in flight) ->
//    s_barrier (everyone's share landed, everyone left stage (kt-1)%3) -> issue tile kt+2 into that stage -> MFMAs on kt.
// --------------------------------------------------------------------------
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, void* lds_wave_base, unsigned off) {
#ifdef CSS_DMA_NOLOAD
  asm volatile("" ::"v"(off));
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, (int)off, 0, 0, 0);
#endif
}

// Template: BM x BN tile with NWM x NWN waves of 64 x (BN/NWN) outputs.  <256,128,4,2> is the kernel described above; the 4-wave
// instances <128,128,2,2> / <128,64,2,2> serve leftover rows and the narrow layers (rows per DMA pass = 8 per wave, so a thread's
// rows are prow + 8*NW*i and the swizzle term ((r >> 1) & 7) stays the same for all of them; vmcnt = A_IT + B_IT).
// NWK = 2 (r03, the leftover launches: <= one workgroup per CU, so a CU's four SIMDs hold ONE wave each and nothing hides the K loop's
// wait -> barrier -> issue -> read -> multiply chain, 0.96 us per K step): a second group of NWM x NWN waves in the same workgroup runs
// the second half of the K steps on a stage ring of its own - two waves per SIMD at different points of the chain, half as many steps
// each - and the halves are added through LDS before the (unchanged) epilogue.  fp32 sums of two halves instead of one chain: the
// result differs from NWK = 1 in the last bit (the parity tests compare both against torch-CPU).
template <int BM, int BN, int NWM, int NWN, int NST = 3, int NWK = 1>
__global__ __launch_bounds__(64 * NWM * NWN * NWK) void conv_igemm_dma_kernel(const ConvArgs a) {
  using T = bf16_t;
  constexpr int BK = 64, VEC = 8, NW = NWM * NWN, RP = 8 * NW;     // NST LDS stages, NST-1 K tiles in flight
  constexpr int A_IT = BM / RP, B_IT = BN / RP;        // DMA wave-instructions per thread per stage (rows t>>3 + RP i)
  static_assert(BM % RP == 0 && BN % RP == 0 && BM / NWM == 64 && RP % 16 == 0, "tile / wave mapping");
  constexpr int ROWB = BK * 2;                         // 128 bytes per LDS row
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int WTM = 64, WTN = BN / NWN, TM = 2, TN = WTN / 32, CSTR = WTN + VEC;
  constexpr int NPAIR = (BM / 128) * NWN;              // (slab, wave column) pairs of the statistics hand-over
  static_assert(NW * WTM * CSTR * 2 <= NST * ST_BYTES, "epilogue staging fits");
  // ONE LDS object on purpose: with a second __shared__ variable the LDS lowering tags every access with alias scopes and
  // the waitcnt pass then puts s_waitcnt vmcnt(0) in front of the fragment reads (it must assume the in-flight LDS-DMA
  // writes alias them), which serialises the two-tiles-in-flight pipeline (measured: 64 -> 90 ms per step).
#ifdef CSS_ABL_NOSSTAT
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NWK * NST * ST_BYTES];
  float* sstat = reinterpret_cast<float*>(smem);
  constexpr bool DSTATS = false;
#else
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NWK * NST * ST_BYTES + NPAIR * 12 * WTN * 4];
  float* sstat = reinterpret_cast<float*>(smem + NWK * NST * ST_BYTES);   // 4 wave pairs x [counter + pad | slot | slot], see store_wave_tile
  constexpr bool DSTATS = true;
  if (a.stats && threadIdx.x < NPAIR) reinterpret_cast<int*>(sstat)[threadIdx.x * 12 * WTN] = 0;   // ordered by the main loop's barriers
#endif

  const int kg = NWK > 1 ? (int)threadIdx.x / (64 * NW) : 0;                 // K group of this wave (wave-uniform)
  const int tid = (int)threadIdx.x - kg * (64 * NW), lane = tid & 63, wave = tid >> 6;     // thread / wave index inside its K group
  const int wm = wave / NWN, wn = wave % NWN;
  unsigned char* const sring = smem + kg * (NST * ST_BYTES);                  // this group's stage ring
  const int nt_n = (a.Cd + BN - 1) / BN;
  const int ntiles = gridDim.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int m0 = a.m_begin + (logical / nt_n) * BM, n0 = (logical % nt_n) * BN;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);

  const int prow = tid >> 3;                                   // tile row of this thread's chunks (+ 64 i)
  const int cch = (tid & 7) ^ ((tid >> 4) & 7);                // source chunk (8 channels) this thread fetches: p ^ ((r>>1)&7)

  int a_base[A_IT], a_h[A_IT], a_w[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + prow + i * RP;
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
    const int n = n0 + prow + i * RP;
    b_off[i] = n < a.Cd ? (unsigned)n * (unsigned)a.Ktot * 2u : OOB;
  }
  int kc = cch * VEC, tr = 0, ts = 0;
  while (kc >= a.Cs) {
    kc -= a.Cs;
    if (++ts == a.S) { ts = 0; ++tr; }
  }
  int kglob = cch * VEC;

  // Kernel rows that read only zero padding for EVERY pixel of this tile are skipped (block-uniform): with the ASPP
  // dilations 12/24/36 on a 65x65 map (aspp.py:36-38) a 256-pixel tile (4 image rows) misses the map with its upper or
  // lower kernel row 12 / 25 / 37 % of the time.  Needs tap boundaries on K-tile boundaries (Cs % BK == 0).
  unsigned tr_mask = 0xffffffffu;
  int nk = (a.Ktot + BK - 1) / BK;
  if (a.Cs % BK == 0 && a.R > 1 && a.R < 32) {
    const int hw = a.Hd * a.Wd;
    const int mlast = min(m0 + BM, a.M) - 1;
    const int i0 = m0 / hw, i1 = mlast / hw;
    const int h0 = (m0 - i0 * hw) / a.Wd, h1 = (mlast - i1 * hw) / a.Wd;
    if (i1 - i0 <= 1) {
      // image rows touched: [h0, h1] (one image) or [h0, Hd-1] and [0, h1] (two images)
      const int alo = h0, ahi = i1 == i0 ? h1 : a.Hd - 1;
      const int blo = i1 == i0 ? h0 : 0, bhi = h1;
      tr_mask = 0;
      int cnt = 0;
      for (int r = 0; r < a.R; ++r) {
        bool v;
        if (a.mode == 0) {
          const int o = r * a.dil - a.pad;    // hs = h*stride + o must fall in [0, Hs) for some h
          v = (alo * a.stride + o <= a.Hs - 1 && ahi * a.stride + o >= 0) || (blo * a.stride + o <= a.Hs - 1 && bhi * a.stride + o >= 0);
        } else {
          const int o = a.pad - r * a.dil;    // th = h + o must fall in [0, (Hs-1)*stride] for some h
          v = (alo + o <= (a.Hs - 1) * a.stride && ahi + o >= 0) || (blo + o <= (a.Hs - 1) * a.stride && bhi + o >= 0);
        }
        if (v) { tr_mask |= 1u << r; ++cnt; }
      }
      nk = cnt * a.S * (a.Cs / BK);
      while (tr < a.R && !((tr_mask >> tr) & 1)) { ++tr; kglob += a.S * a.Cs; }
    }
  }

  // K steps of this group: [kg * nkg, min(nk, (kg + 1) * nkg)); a later group starts at the tap / channel slice of its first step
  // (Cs is a whole number of K tiles for this kernel: a step never straddles two taps)
  const int nkg = (nk + NWK - 1) / NWK;
  int steps_left = nkg;
  if (NWK > 1) {
    const int s0 = kg * nkg;
    steps_left = max(0, min(nk, s0 + nkg) - s0);
    if (kg > 0) {
      const int ncs = a.Cs / BK, v = s0 / ncs, sl = s0 - v * ncs;
      for (int j = 0; j < v && tr < a.R; ++j)
        if (++ts == a.S) {
          ts = 0;
          ++tr;
          while (tr < a.R && !((tr_mask >> tr) & 1)) ++tr;
        }
      kc += sl * BK;
      kglob = (tr * a.S + ts) * a.Cs + kc;
    }
  }
  // Byte offsets of this thread's A rows for the CURRENT tap and channel position; within a tap they simply advance by
  // BK*2 bytes per K tile, so the (expensive) coordinate / bounds arithmetic runs only when the tap changes.
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
  // wave-uniform LDS destinations: instruction i of wave w covers chunks [i*64*NW + w*64, +64) of the stage image
  auto issue = [&](int stage) {
    unsigned char* sa = sring + stage * ST_BYTES + wave * 1024;
    const bool live = NWK == 1 || steps_left > 0;          // past this group's share: zeros into a free stage
    --steps_left;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) dma16(rs_a, sa + i * (NW * 1024), live ? offA[i] : OOB);
    const unsigned kb = (live && kglob < a.Ktot) ? (unsigned)kglob * 2u : OOB;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma16(rs_b, sb + i * (NW * 1024), (b_off[i] | kb) & OOB ? OOB : b_off[i] + kb);
    kglob += BK;
    kc += BK;
    if (kc >= a.Cs) {          // next tap (uniform across the block whenever Cs is a multiple of BK)
      do {
        kc -= a.Cs;
        if (++ts == a.S) {
          ts = 0;
          ++tr;
          while (tr < a.R && !((tr_mask >> tr) & 1)) { ++tr; kglob += a.S * a.Cs; }   // all-padding kernel rows
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
  const int xr = (l31 >> 1) & 7;
  int koff[4];                                         // swizzled byte offset of k-step ks inside a row
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = (((2 * ks + lh) ^ xr) << 4);
  const int a_row = (wm * WTM + l31) * ROWB, b_row = A_BYTES + (wn * WTN + l31) * ROWB;
  auto compute = [&](int stage) {
    const unsigned char* sbase = sring + stage * ST_BYTES;
    // fragments of k-step ks+1 are requested before the MFMAs of k-step ks (two register sets, statically indexed)
    bf16x8 fw[2][TN], fa[2][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) fw[0][i] = *reinterpret_cast<const bf16x8*>(sbase + b_row + i * 32 * ROWB + koff[0]);
#pragma unroll
    for (int j = 0; j < TM; ++j) fa[0][j] = *reinterpret_cast<const bf16x8*>(sbase + a_row + j * 32 * ROWB + koff[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
          fw[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(sbase + b_row + i * 32 * ROWB + koff[ks + 1]);
#pragma unroll
        for (int j = 0; j < TM; ++j)
          fa[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(sbase + a_row + j * 32 * ROWB + koff[ks + 1]);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks & 1][i], fa[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };

#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0) issue(s0);
  int st_c = 0, st_i = NST - 1;
  for (int kt = 0; kt < nkg; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (A_IT + B_IT)) : "memory");   // this wave's share of tile kt landed, later tiles may fly
    __builtin_amdgcn_s_barrier();
    issue(st_i);                       // tile kt+NST-1 (past the end: all-OOB = zeros into a free stage)
#ifndef CSS_DMA_NOCOMPUTE
    compute(st_c);
#endif
    st_c = st_c == NST - 1 ? 0 : st_c + 1;
    st_i = st_i == NST - 1 ? 0 : st_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ghost DMAs must land before the stages are reused below
  __builtin_amdgcn_s_barrier();

  if (NWK > 1) {
    // the second K group hands its partial sums over through ITS ring (free now); the first adds them and runs the epilogue alone
    static_assert(NWK == 1 || NW * TN * TM * 16 * 64 * 4 <= NST * ST_BYTES, "partial sums fit one ring");
    float* red = reinterpret_cast<float*>(smem + NST * ST_BYTES) + wave * (TN * TM * 16 * 64) + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * TM + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TM + j) * 16 + r) * 64];
    }
  }
  // ---- epilogue (as in conv_igemm_kernel): accumulators -> wave-private LDS -> 16-byte row stores
  T* Cw = reinterpret_cast<T*>(smem) + wave * (WTM * CSTR);
  if (kg == 0)
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
      for (int j = 0; j < TM; ++j) {
        T* p = Cw + (j * 32 + l31) * CSTR + nl;
        union { T e[4]; uint2 u2; } pk;
        if constexpr (sizeof(T) == 2) {          // (one v_cvt_pk_bf16_f32 per pair: common.h, pack2_bf16)
          pk.u2.x = pack2_bf16(acc[i][j][4 * q] + bv[0], acc[i][j][4 * q + 1] + bv[1]);
          pk.u2.y = pack2_bf16(acc[i][j][4 * q + 2] + bv[2], acc[i][j][4 * q + 3] + bv[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) pk.e[e] = (T)(acc[i][j][4 * q + e] + bv[e]);
        }
        *reinterpret_cast<uint2*>(p) = pk.u2;
      }
    }
  }
  __syncthreads();
  if (kg == 0)
    store_wave_tile<T, WTM, WTN, CSTR, BN, DSTATS>(a, Cw, m0 + wm * WTM, n0 + wn * WTN, wm & 1, lane, sstat + ((wm >> 1) * NWN + wn) * 12 * WTN);
}

// --------------------------------------------------------------------------
// host-side launchers (called from abi.cpp through these C++ entry points)
// --------------------------------------------------------------------------
// 128-row tiles (leftover rows of the big-tile kernels, layers with Cout <= 64): LDS-DMA instances; CSS_SMALL_DMA=0 selects the
// register-staged kernels instead (kept for the fp32 path and as the A/B reference)
static inline bool small64_nst2() {
  // (measured, alternating processes on one box: c2 116.0 -> 115.3 ms per step, c4 137.4 -> 135.9: profiles/r04_small64_nst2_ab.txt)
  static const int v = getenv("CSS_SMALL64_NST2") ? atoi(getenv("CSS_SMALL64_NST2")) : 1;
  return v != 0;
}
static void launch_small_n64(dim3 g, hipStream_t st, const ConvArgs& b, int n_cu = 0, bool leftover = false) {
  static const bool dma = !(getenv("CSS_SMALL_DMA") && atoi(getenv("CSS_SMALL_DMA")) == 0) && !getenv("CSS_NO_DMA_CONV");
  static const bool no_split = getenv("CSS_NO_SMALL_SPLITK") != nullptr;
  if (dma && b.Cs % 64 == 0 && !no_split && leftover && n_cu > 0 && (int)g.x <= n_cu && b.Ktot >= 8 * 64 && !b.bias) {
    // at most one workgroup per CU (the leftover rows of the persistent kernels): two K groups per workgroup
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, 2, 2, 3, 2>), g, dim3(512), 0, st, b);
  } else if (dma && b.Cs % 64 == 0 && small64_nst2() && n_cu > 0 && (int)g.x > 2 * n_cu) {
    // more than two workgroups per CU: two LDS stages (51 KiB) let THREE share a CU instead of two (the Cout <= 64 layers; CSS_SMALL64_NST2=0: off)
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, 2, 2, 2>), g, dim3(256), 0, st, b);
  } else if (dma && b.Cs % 64 == 0) {      // (channel counts that are not whole K tiles - the 7x7 stem - change tap inside a tile: register-staged)
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, 2, 2>), g, dim3(256), 0, st, b);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, 64, 64, 2, 2>), g, dim3(256), 0, st, b);
  }
}
static void launch_small_n128(dim3 g, hipStream_t st, const ConvArgs& b, int n_cu = 0) {
  static const bool dma = !(getenv("CSS_SMALL_DMA") && atoi(getenv("CSS_SMALL_DMA")) == 0) && !getenv("CSS_NO_DMA_CONV");
  // more workgroups than CUs: TWO LDS stages (72 KiB) instead of three (104 KiB) let two workgroups share a CU - two waves per SIMD at
  // different points of the wait -> barrier -> issue -> read -> multiply chain instead of one (CSS_SMALL_NST2=0/1 overrides)
  // (measured, alternating processes: c4 136.0 -> 133.3 ms per step - its leftover launches are 306 workgroups, two rounds of one per CU
  // before - and c2 113.3 -> 113.1)
  static const int nst2 = getenv("CSS_SMALL_NST2") ? atoi(getenv("CSS_SMALL_NST2")) : 1;
  if (dma && b.Cs % 64 == 0 && nst2 && n_cu > 0 && (int)g.x > n_cu) {
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 2>), g, dim3(256), 0, st, b);
  } else if (dma && b.Cs % 64 == 0) {
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2>), g, dim3(256), 0, st, b);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, 128, 64, 2, 2>), g, dim3(256), 0, st, b);
  }
}

// ONE statement of which kernel family a bf16 launch takes (css_launch_conv branches on it).  Every family lays its statistics slabs out
// per 256-row tile (two 128-row slabs): css_conv_tile_rows_ is what css_conv2d_forward_bnstats_tile_rows reports to the stage-2 kernel.
enum ConvPlan { PLAN_OTHER = 0, PLAN_WS = 1, PLAN_TILE256 = 3 };
static ConvPlan conv_plan(const ConvArgs& a, int dtype, int n_cu) {
  static const bool no_dma = getenv("CSS_NO_DMA_CONV") != nullptr, no_256 = getenv("CSS_NO_DMA256_CONV") != nullptr;
  if (dtype != CSS_BF16 || no_dma || no_256) return PLAN_OTHER;
  if (css_conv_ws_supported(a, n_cu)) return PLAN_WS;                    // conv_ws.hip: 128-row slabs, as the 256-row tiles
  if (a.Cd >= 256 && css_conv_pp_plan(a, n_cu) != 0) return PLAN_TILE256;   // persistent 256x256 tiles (conv_p8.hip / conv_pp.hip)
  return PLAN_OTHER;
}
int css_conv_tile_rows_(const ConvArgs& a, int dtype, int n_cu) { (void)a; (void)dtype; (void)n_cu; return 256; }

int css_launch_conv(const ConvArgs& a_in, int dtype, int n_cu, hipStream_t st, LaunchProf* prof) {
  auto P0 = [&](bool big, double share, bool ws = false) { if (prof) prof->begin(big, share, ws); };
  auto P1 = [&]() { if (prof) prof->end(); };
  ConvArgs a = a_in;
  a.m_begin = 0;
  {
    // CSS_CONV_NT (experiment, round 6): bit 1 = conv_ws_kernel's output stores non-temporal (bit 0, the same in conv_igemm_p8_kernel, sent its
    // residual-add instance to scratch - tests/test_host_cpu.py::test_conv_p8_kernel_isa - and was taken out again)
    static const int conv_nt = getenv("CSS_CONV_NT") ? atoi(getenv("CSS_CONV_NT")) & 3 : 0;
    a.st_nt = conv_nt;
  }
  if (a.M <= 0 || a.Cd <= 0) return CSS_OK;
  if (a.stats && (dtype != CSS_BF16 || a.stat_Mg < 128 || a.addend)) return CSS_ERR_ARG;   // slab statistics: bf16 forward only
  a.stat_nslab = cdiv(a.M, 128);
  a.stat_G = a.stats ? a.M / a.stat_Mg : 0;
  {
    const size_t esz = dtype == CSS_BF16 ? 2 : 4;
    const size_t sb = (size_t)a.N * a.Hs * a.Ws * a.lds * esz, wb = (size_t)a.Cd * a.Ktot * esz;
    if (sb >= 0x7FFFFFF0ull || wb >= 0x7FFFFFF0ull) return CSS_ERR_ARG;   // 32-bit buffer offsets
    a.src_bytes = (unsigned)sb;
    a.wt_bytes = (unsigned)wb;
  }
  if (dtype == CSS_BF16) {
    if (a.Cs % 8 || a.lds % 8 || (reinterpret_cast<uintptr_t>(a.src) & 15) || (reinterpret_cast<uintptr_t>(a.wt) & 15))
      return CSS_ERR_ARG;
    static const bool no_dma = getenv("CSS_NO_DMA_CONV") != nullptr;
    const ConvPlan plan = conv_plan(a, dtype, n_cu);
    if (!no_dma && css_conv_c64_supported(a, dtype)) {
      // 3x3 stride 1 on 64 input channels (layer 1's conv2 and its data gradient, the deep stem): the patch-in-LDS kernel, every row in one launch (conv_c64.hip)
      P0(false, 1.0);
      css_launch_conv_c64(a, n_cu, st);
      P1();
    } else if (plan == PLAN_WS) {
      // short-K 1x1 (conv3 of a Bottleneck forward, conv1 backward): weight-stationary kernel, every row in one launch (conv_ws.hip)
      P0(true, 1.0, true);
      css_launch_conv_ws(a, n_cu, st);
      P1();
    } else if (plan == PLAN_TILE256) {
      // 256x256 tiles: whole rounds of the chip on the big kernel, leftover rows on the 128x128 kernel
      const int nt_n = cdiv(a.Cd, 256), mt = cdiv(a.M, 256);
      int full_mt = mt;
      const double rounds = (double)mt * nt_n / n_cu;
      if (rounds > 1.0 && rounds - (long)rounds < 0.6 && (rounds - (long)rounds) > 1e-9) full_mt = (int)((long)rounds * n_cu / nt_n);
      if (full_mt > 0) {
        ConvArgs b = a;
        b.M = full_mt * 256 < a.M ? full_mt * 256 : a.M;
        P0(true, (double)(b.M - b.m_begin) / a.M);
        {
          // persistent, one workgroup per CU walking full_mt * nt_n tiles: the 8-phase K loop (conv_p8.hip) for every shape with at least
          // three 64-channel K steps per valid kernel row, the 32-channel-step loop (conv_pp.hip) for the few below that
          b.dst_bytes = (unsigned)((size_t)b.M * b.ldd * 2);
          const int tiles = full_mt * nt_n;
          if (css_conv_p8_supported(b)) css_launch_conv_p8(b, tiles < n_cu ? tiles : n_cu, st);
          else css_launch_conv_pp(b, tiles < n_cu ? tiles : n_cu, st);
        }
        P1();
      }
      if (full_mt < mt) {
        ConvArgs b = a;
        b.m_begin = full_mt * 256;
        P0(false, (double)(b.M - b.m_begin) / a.M);
        {
          // few leftover tiles: halve their width so that twice as many CUs share the (latency-bound) K loop
          static const int rem_mode = getenv("CSS_REM_N64") ? atoi(getenv("CSS_REM_N64")) : 1;
          const int wgs128 = cdiv(a.M - b.m_begin, 128) * cdiv(a.Cd, 128);
          if ((rem_mode == 1 && wgs128 * 2 <= n_cu) || (rem_mode == 2 && wgs128 <= n_cu))
            launch_small_n64(dim3(cdiv(a.M - b.m_begin, 128) * cdiv(a.Cd, 64)), st, b, n_cu, true);
          else
            launch_small_n128(dim3(wgs128), st, b, n_cu);
        }
        P1();
      }
    } else if (a.Cd > 64 && !no_dma) {
      // Big tiles (256x128, one workgroup per CU) only for whole rounds of the chip; the leftover rows go to the 128x128
      // kernel (4x as many, smaller tiles, two per CU) instead of paying a full extra round for a fraction of one.
      const int nt_n = cdiv(a.Cd, 128), mt = cdiv(a.M, 256);
      const int slots = n_cu;
      // Round 4: with two workgroups of the two-stage 128x128 kernel per CU the Cout = 65..255 layers run all their rows there - no
      // 256x128 launch + leftover launch pair (conv forward + dgrad kernels of the step: c4 69.7 -> 68.9 ms, c2 59.2 -> 58.7;
      // profiles/r04_n128_small_only_ab.txt).  CSS_N128_SMALL_ONLY=0: the 256x128 kernel for the whole rounds again.
      static const bool small_only = !(getenv("CSS_N128_SMALL_ONLY") && atoi(getenv("CSS_N128_SMALL_ONLY")) == 0);
      int full_mt = small_only ? 0 : mt;
      const double rounds = (double)mt * nt_n / slots;
      if (!small_only && rounds > 1.0 && rounds - (long)rounds < 0.6 && (rounds - (long)rounds) > 1e-9) {
        full_mt = (int)((long)rounds * slots / nt_n);      // m-tiles covered by whole rounds
      }
      if (full_mt > 0) {
        ConvArgs b = a;
        b.M = full_mt * 256 < a.M ? full_mt * 256 : a.M;
        P0(false, (double)(b.M - b.m_begin) / a.M);
        hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 128, 4, 2>), dim3(full_mt * nt_n), dim3(512), 0, st, b);
        P1();
      }
      if (full_mt < mt) {
        ConvArgs b = a;
        b.m_begin = full_mt * 256;
        P0(false, (double)(b.M - b.m_begin) / a.M);
        launch_small_n128(dim3(cdiv(a.M - b.m_begin, 128) * nt_n), st, b, n_cu);
        P1();
      }
    } else if (a.Cd > 64) {
      dim3 g(cdiv(a.M, 128) * cdiv(a.Cd, 128));
      P0(false, (double)(a.M - a.m_begin) / a.M);
      launch_small_n128(g, st, a, n_cu);
      P1();
    } else {
      dim3 g(cdiv(a.M, 128) * cdiv(a.Cd, 64));
      P0(false, (double)(a.M - a.m_begin) / a.M);
      launch_small_n64(g, st, a, n_cu);
      P1();
    }
  } else if (dtype == CSS_F32) {
    if (a.Cs % 4 || a.lds % 4 || (reinterpret_cast<uintptr_t>(a.src) & 15) || (reinterpret_cast<uintptr_t>(a.wt) & 15))
      return CSS_ERR_ARG;
    dim3 g(cdiv(a.M, 64) * cdiv(a.Cd, 64));
    P0(false, (double)(a.M - a.m_begin) / a.M);
    hipLaunchKernelGGL((conv_igemm_kernel<float, 64, 64, 16, 2, 2>), g, dim3(256), 0, st, a);
    P1();
  } else {
    return CSS_ERR_DTYPE;
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

