// Implicit-GEMM convolution for NHWC activations on gfx950 (CDNA4):
//   conv_igemm_kernel : forward and data-gradient (one kernel, two gather modes)
//   conv_wgrad_kernel : weight gradient (split over pixels, fp32 atomic accumulate)
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

// 16-byte buffer load: out-of-range offsets (>= num_records) return zeros in hardware, so the im2col zero padding,
// the M / Cout / K tails and the "ghost" prefetches past the last K tile need no branches, and the compiler can keep
// exact vmcnt counts for a prefetch distance of two tiles.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}


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
typedef __attribute__((address_space(3))) void lds_void;
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
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; all (k-column, cout) tiles of one pixel slice
  // are given to ONE XCD so that the slice of dY / X is fetched into a single L2 (measured before: 3.7x over-fetch).
  const int per_z = a.tiles_k * a.tiles_n;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
  const int zz = (j8 / per_z) * 8 + xcd, t = j8 % per_z;
  if (zz >= a.splits) return;
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
          slab[nl * BKC + kl] = acc[i][j][r];
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

// --------------------------------------------------------------------------
// conv_wgrad_p8_kernel (r03): conv_wgrad_dma256_kernel's tile, fragments, walk and epilogue on the phase structure of conv_p8.hip.
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
__global__ __launch_bounds__(512) void conv_wgrad_p8_kernel(const WgradArgs a) {
  constexpr int BN_ = 256, BKC = 256, BP = 32, NST = 4;
  constexpr int T_BYTES = BP * 512, ST_BYTES = 2 * T_BYTES;     // Y tile then X tile
  constexpr int WTN = 128, WTK = 64, TN = 4, TK = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int per_z = a.tiles_k * a.tiles_n;
  const int xcd = blockIdx.x & 7, j8 = blockIdx.x >> 3;
  const int zz = (j8 / per_z) * 8 + xcd, t = j8 % per_z;
  if (zz >= a.splits) return;
  const int k0 = (t % a.tiles_k) * BKC, n0 = (t / a.tiles_k) * BN_;
  const int m_begin = zz * a.m_per_split;
  const int m_end = min(a.M, m_begin + a.m_per_split);
  const int nit = (m_end - m_begin + BP - 1) / BP;
  if (nit <= 0) return;

  const int prow = tid >> 5;
  const int schunk = ((((tid & 31) >> 2) ^ (prow & 3)) << 2) | (tid & 3);
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
  const int d_n = (int)fdiv((uint32_t)BP, a.fd_hw), d_h = q_w - d_n * a.Hd;
  const int xrow = a.ldx * 2;
  const int sx_w = a.stride * xrow, sx_h = a.stride * a.Ws * xrow, sx_n = a.Hs * a.Ws * xrow;
  const int D0 = d_n * sx_n + d_h * sx_h + d_w * sx_w;
  const int Dw = sx_h - a.Wd * sx_w, Dh = sx_n - a.Hd * sx_h;
  const int ystep = BP * a.ldy * 2;
  int r_m[2], r_hs[2], r_ws[2];
  unsigned r_xo[2], r_yo[2];
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
  const int hs_hi = (a.Hd - 1) * a.stride + dh, ws_hi = (a.Wd - 1) * a.stride + dw_;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void*)smem) + (unsigned)wave * 1024u;
  // offsets of pixel row i as it stands (OOB: past the slice / padding / tail columns -> zeros), then the row moves on by BP pixels
  auto row_offsets = [&](int i, unsigned& vy, unsigned& vx) {
    vy = (n_ok && r_m[i] < m_end) ? r_yo[i] : OOB;
    const bool ok = k_ok && r_m[i] < m_end && (unsigned)r_hs[i] < (unsigned)a.Hs && (unsigned)r_ws[i] < (unsigned)a.Ws;
    vx = ok ? r_xo[i] : OOB;
    asm volatile("" : "+v"(vy), "+v"(vx));     // (pinned: the optimizer must not sink this into the load segment that uses it)
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
  };
  auto stage_half = [&](int stage, int i, unsigned vy, unsigned vx) {
    const unsigned sy = lds0 + (unsigned)stage * ST_BYTES + (unsigned)i * 8192u;
    dma16_lds(rs_y, sy, vy);
    dma16_lds(rs_x, sy + T_BYTES, vx);
  };

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: steps 0, 1, 2 (data phases 0..5); offsets of data phase 6 ready for the first phase's issue
  unsigned vy, vx;
#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      row_offsets(i, vy, vx);
      stage_half(st, i, vy, vx);
    }
  row_offsets(0, vy, vx);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // data phase 0 has landed
  __builtin_amdgcn_s_barrier();
  if (wn == 1) __builtin_amdgcn_s_barrier();            // the second cout half runs one barrier behind
  asm volatile("" ::: "memory");

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
  typedef __attribute__((address_space(3))) s16x4 wgp_lds_v4;
  auto frag = [&](unsigned ad) {
    union { struct { s16x4 a, b; } s; bf16x8 f; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgp_lds_v4*)(uintptr_t)ad);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wgp_lds_v4*)(uintptr_t)(ad + 4 * 512));
    return u.f;
  };
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

// dw[n][k] += sum over the pixel slices of ws[tile][slice][n - n0][k - k0] (fixed order: the weight gradient is bit-reproducible).
// grid = (64, tiles): block (bx, t) owns rows 4 bx .. 4 bx + 3 of tile t; thread = 4 consecutive k columns.
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int tiles_k,
                                                                int Cd, int Ktot) {
  const int t = blockIdx.y;
  const int k0 = (t % tiles_k) * 256, n0 = (t / tiles_k) * 256;
  const int nl = blockIdx.x * 4 + (threadIdx.x >> 6), kl = (threadIdx.x & 63) * 4;
  const int n = n0 + nl, k = k0 + kl;
  if (n >= Cd || k >= Ktot) return;
  const float4* p = reinterpret_cast<const float4*>(ws + (size_t)t * splits * (256 * 256) + nl * 256 + kl);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  int z = 0;
  for (; z + 3 < splits; z += 4) {        // four slabs in flight
    const float4 v0 = p[(size_t)(z + 0) * (256 * 256 / 4)], v1 = p[(size_t)(z + 1) * (256 * 256 / 4)];
    const float4 v2 = p[(size_t)(z + 2) * (256 * 256 / 4)], v3 = p[(size_t)(z + 3) * (256 * 256 / 4)];
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
    s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
  }
  for (; z < splits; ++z) {
    const float4 v0 = p[(size_t)z * (256 * 256 / 4)];
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
                                                                    int Cd, int Ktot) {
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
    const float4 v0 = p[(size_t)(z + 0) * SL], v1 = p[(size_t)(z + 1) * SL], v2 = p[(size_t)(z + 2) * SL], v3 = p[(size_t)(z + 3) * SL];
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
    s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
    s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
  }
  for (; z < splits; ++z) {
    const float4 v0 = p[(size_t)z * SL];
    s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
  }
  const float r[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  float* o = dw + (size_t)n * Ktot + k;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k + e < Ktot) o[e] += r[e];
}

// --------------------------------------------------------------------------
// host-side launchers (called from abi.cpp through these C++ entry points)
// --------------------------------------------------------------------------
// 128-row tiles (leftover rows of the big-tile kernels, layers with Cout <= 64): LDS-DMA instances; CSS_SMALL_DMA=0 selects the
// register-staged kernels instead (kept for the fp32 path and as the A/B reference)
static void launch_small_n64(dim3 g, hipStream_t st, const ConvArgs& b, int n_cu = 0) {
  static const bool dma = !(getenv("CSS_SMALL_DMA") && atoi(getenv("CSS_SMALL_DMA")) == 0) && !getenv("CSS_NO_DMA_CONV");
  static const bool no_split = getenv("CSS_NO_SMALL_SPLITK") != nullptr;
  if (dma && b.Cs % 64 == 0 && !no_split && n_cu > 0 && (int)g.x <= n_cu && b.Ktot >= 8 * 64 && !b.bias) {
    // at most one workgroup per CU (the leftover rows of the persistent kernels): two K groups per workgroup
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, 2, 2, 3, 2>), g, dim3(512), 0, st, b);
  } else if (dma && b.Cs % 64 == 0) {      // (channel counts that are not whole K tiles - the 7x7 stem - change tap inside a tile: register-staged)
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, 2, 2>), g, dim3(256), 0, st, b);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, 64, 64, 2, 2>), g, dim3(256), 0, st, b);
  }
}
static void launch_small_n128(dim3 g, hipStream_t st, const ConvArgs& b) {
  static const bool dma = !(getenv("CSS_SMALL_DMA") && atoi(getenv("CSS_SMALL_DMA")) == 0) && !getenv("CSS_NO_DMA_CONV");
  if (dma && b.Cs % 64 == 0) {
    hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2>), g, dim3(256), 0, st, b);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, 128, 64, 2, 2>), g, dim3(256), 0, st, b);
  }
}

// rows per statistics slab pair of a forward launch (see css_conv2d_forward_bnstats): 272 when the 272-row persistent tiling is used
// ONE statement of which tiling a bf16 launch takes (css_launch_conv branches on it, css_conv2d_forward_bnstats_tile_rows reports it: the
// statistics slabs bn_reduce_slabs_kernel reads are laid out by this number, so the two must not be able to drift apart)
enum ConvPlan { PLAN_OTHER = 0, PLAN_WS = 1, PLAN_PP272 = 2, PLAN_TILE256 = 3 };
static ConvPlan conv_plan(const ConvArgs& a, int dtype, int n_cu) {
  static const bool no_dma = getenv("CSS_NO_DMA_CONV") != nullptr, no_256 = getenv("CSS_NO_DMA256_CONV") != nullptr;
  if (dtype != CSS_BF16 || no_dma || no_256) return PLAN_OTHER;
  if (css_conv_ws_supported(a, n_cu)) return PLAN_WS;                    // conv_ws.hip: 128-row slabs, as the 256-row tiles
  if (a.Cd >= 256 && !a.add_mask && css_conv_pp_plan(a, n_cu) == 272) return PLAN_PP272;
  if (a.Cd >= 256) return PLAN_TILE256;
  return PLAN_OTHER;
}
int css_conv_tile_rows_(const ConvArgs& a, int dtype, int n_cu) { return conv_plan(a, dtype, n_cu) == PLAN_PP272 ? 272 : 256; }

int css_launch_conv(const ConvArgs& a_in, int dtype, int n_cu, hipStream_t st, LaunchProf* prof) {
  auto P0 = [&](bool big, double share, bool ws = false) { if (prof) prof->begin(big, share, ws); };
  auto P1 = [&]() { if (prof) prof->end(); };
  ConvArgs a = a_in;
  a.m_begin = 0;
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
    if (plan == PLAN_WS) {
      // short-K 1x1 (conv3 of a Bottleneck forward, conv1 backward): weight-stationary kernel, every row in one launch (conv_ws.hip)
      P0(true, 1.0, true);
      css_launch_conv_ws(a, n_cu, st);
      P1();
    } else if (plan == PLAN_PP272) {
      // 272-row tiles of the persistent kernel cover every row in whole rounds of the chip: one launch
      ConvArgs b = a;
      b.dst_bytes = (unsigned)((size_t)b.M * b.ldd * 2);
      const int tiles = cdiv(a.M, 272) * cdiv(a.Cd, 256);
      P0(true, 1.0);
      css_launch_conv_pp(b, 272, tiles < n_cu ? tiles : n_cu, st);
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
        if (css_conv_pp_plan(b, n_cu) != 0) {
          // second-generation kernel (conv_pp.hip): persistent, one workgroup per CU walking full_mt * nt_n tiles
          b.dst_bytes = (unsigned)((size_t)b.M * b.ldd * 2);
          const int tiles = full_mt * nt_n;
          if (css_conv_p8_supported(b)) css_launch_conv_p8(b, tiles < n_cu ? tiles : n_cu, st);        // 8-phase K loop (conv_p8.hip)
          else if (css_conv_pp64_supported(b)) css_launch_conv_pp64(b, tiles < n_cu ? tiles : n_cu, st);   // 128-byte rows (conv_pp64.hip)
          else css_launch_conv_pp(b, 256, tiles < n_cu ? tiles : n_cu, st);
        } else {
          hipLaunchKernelGGL(conv_igemm_dma256_kernel, dim3(full_mt * nt_n), dim3(512), 0, st, b);
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
            launch_small_n64(dim3(cdiv(a.M - b.m_begin, 128) * cdiv(a.Cd, 64)), st, b, n_cu);
          else
            launch_small_n128(dim3(wgs128), st, b);
        }
        P1();
      }
    } else if (a.Cd > 64 && !no_dma) {
      // Big tiles (256x128, one workgroup per CU) only for whole rounds of the chip; the leftover rows go to the 128x128
      // kernel (4x as many, smaller tiles, two per CU) instead of paying a full extra round for a fraction of one.
      const int nt_n = cdiv(a.Cd, 128), mt = cdiv(a.M, 256);
      const int slots = n_cu;
      int full_mt = mt;
      const double rounds = (double)mt * nt_n / slots;
      if (rounds > 1.0 && rounds - (long)rounds < 0.6 && (rounds - (long)rounds) > 1e-9) {
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
        launch_small_n128(dim3(cdiv(a.M - b.m_begin, 128) * nt_n), st, b);
        P1();
      }
    } else if (a.Cd > 64) {
      dim3 g(cdiv(a.M, 128) * cdiv(a.Cd, 128));
      P0(false, (double)(a.M - a.m_begin) / a.M);
      launch_small_n128(g, st, a);
      P1();
    } else {
      dim3 g(cdiv(a.M, 128) * cdiv(a.Cd, 64));
      P0(false, (double)(a.M - a.m_begin) / a.M);
      launch_small_n64(g, st, a);
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

// Pixel splits of the weight gradient: a multiple of 8 (one slice per XCD at a time, see the kernels) chosen so that the tiles
// an XCD owns (tiles per slice x slices per XCD) fill its 32 CUs x resident workgroups in whole rounds, with >= 4 iterations each.
void css_wgrad_plan_(int M, int Ktot, int Cd, int dtype, int n_cu, int* splits_out, int* mps_out) {
  static const bool no_256 = getenv("CSS_NO_DMA256_WGRAD") != nullptr;
  const bool big = dtype == CSS_BF16 && Cd >= 256 && Ktot >= 256 && !no_256;
  const int bn = dtype == CSS_BF16 ? (big ? 256 : 128) : 64, bkc = bn, bp = dtype == CSS_BF16 ? (big ? 32 : 64) : 16;
  const int tiles = cdiv(Ktot, bkc) * cdiv(Cd, bn);
  const int slots = (n_cu / 8) * (big ? 1 : 2);
  int best_k = 1;
  double best_eff = 0;
  for (int k = 1; k <= 64; ++k) {
    const int mps_k = cdiv(cdiv(M, 8 * k), bp) * bp;
    if (k > 1 && mps_k < 4 * bp) break;
    const int txcd = tiles * k;
    const double eff = (double)txcd / ((double)cdiv(txcd, slots) * slots);
    if (eff > best_eff + 1e-9) { best_eff = eff; best_k = k; }
    if (eff >= 0.93) break;
  }
  int splits = 8 * best_k;
  const int mps = cdiv(cdiv(M, splits), bp) * bp;
  splits = cdiv(M, mps);
  *splits_out = splits;
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
  const int bn = big ? 256 : (dtype == CSS_BF16 ? 128 : 64);        // (square tiles: css_launch_wgrad)
  return (size_t)cdiv(Ktot, bn) * cdiv(Cd, bn) * splits * ((size_t)bn * bn * sizeof(float));
}

int css_launch_wgrad(WgradArgs a, int dtype, int n_cu, hipStream_t st, LaunchProf* prof) {
  if (a.M <= 0) return CSS_OK;
  a.fd_hw = make_fastdiv((uint32_t)(a.Hd * a.Wd));
  a.fd_w = make_fastdiv((uint32_t)a.Wd);
  int bn, bkc, bp;
  static const bool no_256 = getenv("CSS_NO_DMA256_WGRAD") != nullptr;
  const bool big = dtype == CSS_BF16 && a.Cd >= 256 && a.Ktot >= 256 && !no_256;   // 256x256 LDS-DMA kernel, one workgroup per CU
  if (dtype == CSS_BF16) {
    bn = big ? 256 : 128; bkc = big ? 256 : 128; bp = big ? 32 : 64;
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
  dim3 g(a.tiles_k * a.tiles_n * cdiv(splits, 8) * 8);
  if (prof) prof->begin(big, 1.0, false);
  if ((size_t)a.tiles_k * a.tiles_n * a.splits * ((size_t)bn * bkc * sizeof(float)) > a.ws_bytes) a.ws = nullptr;   // atomics path (not reproducible)
  if (big) {
    // CSS_WGRAD_KERNEL: 0 = conv_wgrad_dma256_kernel (both cout halves in lockstep), 1 = the same with the second half one half-step
    // behind, 2 (default) = conv_wgrad_p8_kernel (two-phase steps)
    static const int which = getenv("CSS_WGRAD_KERNEL") ? atoi(getenv("CSS_WGRAD_KERNEL")) : (getenv("CSS_WGRAD_NOSTAGGER") ? 0 : 2);
    if (which == 0) hipLaunchKernelGGL(conv_wgrad_dma256_kernel<false>, g, dim3(512), 0, st, a);
    else if (which == 1) hipLaunchKernelGGL(conv_wgrad_dma256_kernel<true>, g, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(conv_wgrad_p8_kernel, g, dim3(512), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(64, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits, a.tiles_k, a.Cd, a.Ktot);
  } else if (dtype == CSS_BF16) {
    hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 128, 128, 64>), g, dim3(256), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL((wgrad_slab_reduce_gen_kernel<128, 128>), dim3(128 / 8, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits,
                         a.tiles_k, a.Cd, a.Ktot);
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<float, 64, 64, 16>), g, dim3(256), 0, st, a);
    if (a.ws)
      hipLaunchKernelGGL((wgrad_slab_reduce_gen_kernel<64, 64>), dim3(64 / 16, a.tiles_k * a.tiles_n), dim3(256), 0, st, a.ws, a.dw, a.splits,
                         a.tiles_k, a.Cd, a.Ktot);
  }
  if (prof) prof->end();
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
