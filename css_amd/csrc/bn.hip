// Train/eval BatchNorm for NHWC activations [M][ld] on gfx950 (HBM-bound).
// Replaces nn.BatchNorm2d / nn.SyncBatchNorm as used on the reference hot path
// (generalframeworks/networks/resnet.py:110-137, deeplabv3/aspp.py:21,32,49,62,
//  deeplabv3/deeplabv3.py:117,123,130; SyncBN conversion at mix_label.py:76):
//   bn_stats      per-channel sum / sum-of-squares (fp64 accumulators; cross-rank
//                 all-reduce of these 2C numbers replaces SyncBN's all_gather)
//   bn_finalize   mean / invstd / fused scale+shift, running-stat update
//   bn_apply      a = act(scale*y + shift [+ residual])
//   bn_bwd_reduce sum(dz), sum(dz*xhat) with dz = da * (a > 0)
//   bn_bwd_apply  dy = scale*(dz - mean(dz) - xhat*mean(dz*xhat)), optional residual grad
#include "common.h"
#include <type_traits>
#include <cstdlib>

// generic two-value per-channel reduction over rows ------------------------------------
// Stage 1: every block reduces its row range and STORES one partial row  partial[blockIdx.x][2][C]  (fp64, plain
// stores: 1000 blocks doing fp64 atomics onto the same 2C addresses ran at ~390 GB/s, the contended-atomic regime).
// Stage 2 (bn_reduce*_kernel below) sums the partial rows in fp64.
template <typename T, typename F>
__device__ __forceinline__ void channel_reduce2(F f, int Mg, int C, int rows_per_block, double* partial, bool rev = false) {
  // blockIdx.z = statistics group (one group per forward pass batched into the tensor); gridDim.x = row blocks per group
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const int TPC = CV < 256 ? CV : 256;
  const int RPB = 256 / TPC;
  const int tid = threadIdx.x;
  const int cvi = tid % TPC, rg = tid / TPC;
  const int cv = blockIdx.y * TPC + cvi;
  const bool active = rg < RPB && cv < CV;
  float s0[VEC], s1[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s0[e] = s1[e] = 0.f;
  const int gbase = blockIdx.z * Mg;
  const int bx = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;      // (bn_pass_order: last rows first)
  const int row0 = gbase + bx * rows_per_block;
  const int row1 = min(gbase + Mg, row0 + rows_per_block);
  if (active) {
    int r = row0 + rg;
    for (; r + 3 * RPB < row1; r += 4 * RPB) {   // four independent rows in flight (latency-bound otherwise)
      f(r, cv * VEC, s0, s1);
      f(r + RPB, cv * VEC, s0, s1);
      f(r + 2 * RPB, cv * VEC, s0, s1);
      f(r + 3 * RPB, cv * VEC, s0, s1);
    }
    for (; r < row1; r += RPB) f(r, cv * VEC, s0, s1);
  }
  __shared__ float red[2][256 * VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[0][tid * VEC + e] = s0[e];
    red[1][tid * VEC + e] = s1[e];
  }
  __syncthreads();
  if (rg == 0 && cv < CV) {
    double* p0 = partial + ((size_t)blockIdx.z * gridDim.x + bx) * 2 * C + cv * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      double a0 = 0, a1 = 0;
      for (int g = 0; g < RPB; ++g) {
        a0 += red[0][(g * TPC + cvi) * VEC + e];
        a1 += red[1][(g * TPC + cvi) * VEC + e];
      }
      p0[e] = a0;
      p0[C + e] = a1;
    }
  }
}

// ---- stage 2 ---------------------------------------------------------------------------------------------------------
// sums[g][j] = sum over the partial rows of group g, j in [0, 2C), in fp64.  These kernels are pure latency (a few MB read
// by C/8 blocks), so: 1024 threads = 8 channels x 128 row partitions, the partitions dealt over as many groups as divide
// 128 (all groups of a launch run concurrently), 8 independent loads in flight per thread, tree reduction in LDS.
// rows(g, lo, hi): partial rows [lo, hi) belong to group g.  emit(g, t0, t1, c) is called
// by ONE thread per channel, for g = 0..G-1 in order (the running statistics take their G momentum updates in order).
#ifndef S2_CH_V
#define S2_CH_V 8
#endif
constexpr int S2_CH = S2_CH_V, S2_NP = 1024 / S2_CH_V;
struct NoTail { __device__ __forceinline__ void operator()(int, int, int, int, double&, double&) const {} };
// tail(g, c, pl, P, t0, t1): optional extra contribution of partition pl (of P) to the sums of (group g, channel c)
// Latency is everything here (profiles/r03_bn_stage2_prefetch.txt): ONE memory round trip for the usual launch - all of a thread's partial
// rows (up to S2_U) are requested at once, the tail's rows right behind them, and only then the first value is used (the callers fetch
// gamma / beta / the running statistics before they call this, for the same reason); partitions are summed inside a wave with lane
// shuffles, the 16 waves through LDS with one barrier.
constexpr int S2_U = 12;
__device__ __forceinline__ double s2_shfl_xor(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}
template <typename PT, typename Rows, typename Emit, typename Tail = NoTail>
__device__ __forceinline__ void stage2_reduce(const PT* __restrict__ partial, int C, int G, Rows rows, Emit emit, Tail tail = Tail()) {
  static_assert(S2_CH == 8, "a wave = 8 partitions x 8 channels");
  __shared__ double red[2][16][S2_CH];
  const int cl = threadIdx.x & (S2_CH - 1), part = threadIdx.x / S2_CH, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * S2_CH + cl;
  const int cc = c < C ? c : C - 1;                           // (threads past the last channel read channel C-1 and are never emitted)
  const int GP = (G <= 16 && S2_NP % G == 0) ? G : 1;         // groups reduced concurrently (a wave's 8 partitions stay in one group)
  const int P = S2_NP / GP;                                   // partitions per group
  const int gslot = part / P, pl = part % P;
  for (int g0 = 0; g0 < G; g0 += GP) {
    const int g = g0 + gslot;
    int lo, hi;
    rows(g, lo, hi);
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    // first S2_U rows of this thread: every request goes out before the first use (rows past `hi` re-read row `lo` and count as zero)
    PT v0[S2_U], v1[S2_U];
    const int safe = lo < hi ? lo : 0;
#pragma unroll
    for (int u = 0; u < S2_U; ++u) {
      const int rr = lo + pl + u * P;
      const size_t src = (size_t)(rr < hi ? rr : safe) * 2 * C + cc;
      v0[u] = partial[src];
      v1[u] = partial[src + C];
    }
    double t0 = 0.0, t1 = 0.0;
    tail(g, cc, pl, P, t0, t1);
#pragma unroll
    for (int u = 0; u < S2_U; ++u) {
      const bool ok = lo + pl + u * P < hi;
      a0[u & 3] += ok ? (double)v0[u] : 0.0;
      a1[u & 3] += ok ? (double)v1[u] : 0.0;
    }
    for (int r = lo + pl + S2_U * P; r < hi; r += 4 * P) {      // (more than S2_U x P partial rows: the c4-sized tensors)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = r + u * P < hi;
        const size_t src = (size_t)(ok ? r + u * P : safe) * 2 * C + cc;
        const PT w0 = partial[src], w1 = partial[src + C];
        a0[u] += ok ? (double)w0 : 0.0;
        a1[u] += ok ? (double)w1 : 0.0;
      }
    }
    double s0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + t0, s1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + t1;
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {                          // over the wave's 8 partitions (lane = partition * 8 + channel)
      s0 += s2_shfl_xor(s0, m);
      s1 += s2_shfl_xor(s1, m);
    }
    if ((threadIdx.x & 63) < S2_CH) {
      red[0][wave][cl] = s0;
      red[1][wave][cl] = s1;
    }
    __syncthreads();
    if (part == 0 && c < C) {
      const int wpg = P / 8;                                    // waves per group
      for (int q = 0; q < GP; ++q) {
        double e0 = 0.0, e1 = 0.0;
        for (int w = 0; w < wpg; ++w) {
          e0 += red[0][q * wpg + w][cl];
          e1 += red[1][q * wpg + w][cl];
        }
        emit(g0 + q, e0, e1, c);
      }
    }
    __syncthreads();
  }
}
// train-mode finalize for G groups: per-group mean/invstd/scale/shift ([G][C]); the running statistics take the G
// momentum updates one after the other, exactly like G separate forward passes (mix_label.py:166 -> ddp_model.py:102-103,140-143)
__device__ __forceinline__ void bn_finalize_one(double a0, double a1, double count, float gam, float bet, float* running_mean, float* running_var,
                                                float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                                                int c) {
  double mean = a0 / count;
  double var = a1 / count - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean, fvar = (float)var;
  const float invstd = 1.0f / sqrtf(fvar + eps);
  mean_out[c] = fmean;
  invstd_out[c] = invstd;
  const float sc = gam * invstd;
  scale_out[c] = sc;
  shift_out[c] = bet - fmean * sc;
  if (running_mean) {
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}
// the same with gamma / beta / the running statistics already in registers (fetched before the reduction: their round trip hides
// behind the partial rows'); the running statistics are stored after every group's update
__device__ __forceinline__ void bn_finalize_reg(double a0, double a1, double count, float gam, float bet, float& rm, float& rv, float* running_mean,
                                                float* running_var, float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out,
                                                float* shift_out, int c) {
  double mean = a0 / count;
  double var = a1 / count - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean, fvar = (float)var;
  const float invstd = 1.0f / sqrtf(fvar + eps);
  mean_out[c] = fmean;
  invstd_out[c] = invstd;
  const float sc = gam * invstd;
  scale_out[c] = sc;
  shift_out[c] = bet - fmean * sc;
  if (running_mean) {
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    rm = (1.f - momentum) * rm + momentum * fmean;
    rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}
// grid = ceil(C/8), block 1024.  partial is [G][nrb][2][C] (fp64); sums is [G][2][C].
__global__ __launch_bounds__(1024) void bn_reduce_kernel(const double* __restrict__ partial, int nrb, int C, int G, double* __restrict__ sums,
                                                         float* __restrict__ g1, float* __restrict__ g0, int accumulate, double count_local) {
  double t0 = 0, t1 = 0;
  // SyncBN with per-rank pixel counts: the local count of every group rides behind the sums and is all-reduced with them
  if (sums && count_local > 0 && blockIdx.x == 0 && threadIdx.x < G) sums[(size_t)G * 2 * C + threadIdx.x] = count_local;
  const int ce = blockIdx.x * S2_CH + (int)threadIdx.x;          // the emitting threads (0..S2_CH-1): their channel
  float old0 = 0.f, old1 = 0.f;
  if (accumulate && g0 && threadIdx.x < S2_CH && ce < C) { old0 = g0[ce]; old1 = g1[ce]; }      // (before the reduction: see stage2_reduce)
  stage2_reduce(
      partial, C, G, [&](int g, int& lo, int& hi) { lo = g * nrb; hi = lo + nrb; },
      [&](int g, double a0, double a1, int c) {
        if (sums) { sums[(size_t)g * 2 * C + c] = a0; sums[(size_t)g * 2 * C + C + c] = a1; }
        t0 += a0; t1 += a1;
        if (g == G - 1 && g0) {   // BN backward: dbeta = sum(dz), dgamma = sum(dz*xhat), summed over the groups
          g0[c] = old0 + (float)t0;
          g1[c] = old1 + (float)t1;
        }
      });
}
// train-mode finalize fused with the sum (single rank): per-group mean/invstd/scale/shift ([G][C])
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const double* __restrict__ partial, int nrb, int G, double count,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* running_mean, float* running_var, float momentum, float eps,
                                                                  float* mean_out, float* invstd_out, float* scale_out, float* shift_out, int C) {
  const int ce = blockIdx.x * S2_CH + (int)threadIdx.x;
  float gam = 0.f, bet = 0.f, rm = 0.f, rv = 0.f;
  if (threadIdx.x < S2_CH && ce < C) {
    gam = gamma[ce]; bet = beta[ce];
    if (running_mean) { rm = running_mean[ce]; rv = running_var[ce]; }
  }
  stage2_reduce(
      partial, C, G, [&](int g, int& lo, int& hi) { lo = g * nrb; hi = lo + nrb; },
      [&](int g, double a0, double a1, int c) {
        bn_finalize_reg(a0, a1, count, gam, bet, rm, rv, running_mean, running_var, momentum, eps, mean_out + g * C, invstd_out + g * C,
                        scale_out + g * C, shift_out + g * C, c);
      });
}

// Stage 2 for the statistics the convolution epilogue emits (conv.hip: store_wave_tile, conv_pp.hip): partial is fp32
// [nslab][2][C], two rows ("slabs") per convolution tile of BM rows of the [G*Mg][C] tensor y: slab p covers rows
// (p >> 1) * BM + (p & 1) * (BM - 128) ... of the tile's first (BM - 128 rows) or second (128 rows) pixel half and holds the sums of
// those of its rows that belong to the statistics group of its FIRST row.  BM = 256 (128-row slabs: every convolution kernel of the library).
// Group g = the slabs that START inside it plus - when g*Mg is not a slab start - the rows g*Mg .. (next slab start) of y itself (< 128 rows,
// summed here from the bf16 tensor: the head of the group sits in a slab that started in the previous group).
// sums_out != null: write [G][2][C] sums (+ [G] row counts) only (SyncBN: all-reduced before bn_finalize); else finalize in place.
__device__ __forceinline__ int slab_start(int p, int BM) { return (p >> 1) * BM + (p & 1) * (BM - 128); }
__device__ __forceinline__ int first_slab_from(int row, int BM) {         // first slab whose start is >= row (rows < 2^31)
  const int t = BM == 256 ? row >> 8 : row / BM, rem = row - t * BM;
  return 2 * t + (rem == 0 ? 0 : (rem <= BM - 128 ? 1 : 2));
}
__global__ __launch_bounds__(1024) void bn_reduce_slabs_kernel(const float* __restrict__ partial, int nslab, int Mg, int G, double count,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* running_mean, float* running_var, float momentum, float eps,
                                                               float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                                                               double* sums_out, int C, const bf16_t* __restrict__ y, int ldy, int BM) {
  if (sums_out && blockIdx.x == 0 && threadIdx.x < G) sums_out[(size_t)G * 2 * C + threadIdx.x] = (double)Mg;   // local count, see bn_reduce_kernel
  const int ce = blockIdx.x * S2_CH + (int)threadIdx.x;
  float gam = 0.f, bet = 0.f, rm = 0.f, rv = 0.f;
  if (!sums_out && threadIdx.x < S2_CH && ce < C) {
    gam = gamma[ce]; bet = beta[ce];
    if (running_mean) { rm = running_mean[ce]; rv = running_var[ce]; }
  }
  stage2_reduce(
      partial, C, G,
      [&](int g, int& lo, int& hi) {
        const int b = g * Mg, e = b + Mg;
        lo = first_slab_from(b, BM);
        hi = first_slab_from(e, BM);
        if (hi > nslab) hi = nslab;
      },
      [&](int g, double a0, double a1, int c) {
        if (sums_out) {
          sums_out[(size_t)g * 2 * C + c] = a0;
          sums_out[(size_t)g * 2 * C + C + c] = a1;
        } else {
          bn_finalize_reg(a0, a1, count, gam, bet, rm, rv, running_mean, running_var, momentum, eps, mean_out + g * C, invstd_out + g * C,
                          scale_out + g * C, shift_out + g * C, c);
        }
      },
      [&](int g, int c, int pl, int P, double& t0, double& t1) {
        const int b = g * Mg;
        int e = slab_start(first_slab_from(b, BM), BM);      // == b when the group starts on a slab boundary
        if (e > b + Mg) e = b + Mg;
        for (int r = b + pl; r < e; r += P) {
          const float v = (float)y[(size_t)r * ldy + c];
          t0 += (double)v;
          t1 += (double)v * (double)v;
        }
      });
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y, int Mg, int C, int ld,
                                                       int rows_per_block, double* partial) {
  constexpr int VEC = 16 / sizeof(T);
  auto f = [&](int r, int c, float* s0, float* s1) {
    Vec16<T> v;
    v.load(y + (size_t)r * ld + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float x = v.f(e);
      s0[e] += x;
      s1[e] += x * x;
    }
  };
  channel_reduce2<T>(f, Mg, C, rows_per_block, partial);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int G, double count, const double* __restrict__ count_dev,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   float* mean_out, float* invstd_out, float* scale_out, float* shift_out, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  for (int g = 0; g < G; ++g)
    bn_finalize_one(sums[(size_t)g * 2 * C + c], sums[(size_t)g * 2 * C + C + c], count_dev ? count_dev[g] : count, gamma[c], beta[c], running_mean, running_var, momentum, eps,
                    mean_out + g * C, invstd_out + g * C, scale_out + g * C, shift_out + g * C, c);
}

// eval mode: scale/shift from running statistics
__global__ void bn_eval_coeff_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float eps, float* scale_out, float* shift_out, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.0f / sqrtf(running_var[c] + eps);
  float sc = gamma[c] * invstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - running_mean[c] * sc;
}

#define EW_UNROLL 4
// Elementwise BN kernels: thread t owns channel vector (blockIdx.y*TPC + t % TPC) for all of its rows, so the per-channel
// coefficients are loaded ONCE into registers (they used to be re-loaded for every element and made these kernels
// instruction-bound).  grid = (row blocks, channel blocks, groups).
template <typename T, bool MASK, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ res, int ldr,
                                                       T* __restrict__ out, int ldo, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int Mg, int C, int rows_per_block,
                                                       unsigned char* __restrict__ mask, int rev) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const int TPC = CV < 256 ? CV : 256, RPB = 256 / TPC;
  const int cvi = threadIdx.x % TPC, rg = threadIdx.x / TPC;
  const int cv = blockIdx.y * TPC + cvi;
  if (rg >= RPB || cv >= CV) return;
  const int c = cv * VEC, g = blockIdx.z;
  using P = Pairs<T>;
  f32x2 sc[P::NP], sh[P::NP];
#pragma unroll
  for (int p = 0; p < P::NP; ++p) {
    sc[p] = f32x2{scale[g * C + c + 2 * p], scale[g * C + c + 2 * p + 1]};
    sh[p] = f32x2{shift[g * C + c + 2 * p], shift[g * C + c + 2 * p + 1]};
  }
  const int gbase = g * Mg;
  // (experiment switch CSS_BN_NT, round 6: bits 1 / 2 / 3 of `rev` = non-temporal loads of y / of the residual / non-temporal stores of the output)
  const int nt = rev >> 1;
  rev &= 1;
  const int bx = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;      // (bn_pass_order: last rows first)
  const int row0 = gbase + bx * rows_per_block, row1 = min(gbase + Mg, row0 + rows_per_block);
  // EW_UNROLL rows per trip, all loads issued before the first use: a thread's trips are a serial chain of ~2 us memory
  // round trips, which (not bandwidth) bounded the 10-40 MB layers.
  for (int r = row0 + rg; r < row1; r += EW_UNROLL * RPB) {
    Vec16<T> v[EW_UNROLL], rr[EW_UNROLL];
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
      const int ru = r + u * RPB;
      if (ru < row1) {
        v[u].load(y + (size_t)ru * ldy + c, (nt & 1) != 0);
        if (RES) rr[u].load(res + (size_t)ru * ldr + c, (nt & 2) != 0);
      }
    }
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
      const int ru = r + u * RPB;
      if (ru < row1) {
        Vec16<T> o;
#pragma unroll
        for (int p = 0; p < P::NP; ++p) {
          f32x2 x = P::get(v[u], p) * sc[p] + sh[p];
          if (RES) x += P::get(rr[u], p);
          if (RELU) x = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
          P::set(o, p, x);
        }
        if (nt & 8) o.store_sc1(out + (size_t)ru * ldo + c);      // (CSS_BN_NT bit 3: write-through output stores, -0.5 ms per step)
        else o.store(out + (size_t)ru * ldo + c, (nt & 4) != 0);
        if (MASK) {      // ReLU mask for the backward passes: one byte per vector, of the STORED values (what reading `out` back would give)
          unsigned bits = 0;
#pragma unroll
          for (int e = 0; e < VEC; ++e) bits |= (o.f(e) > 0.f ? 1u : 0u) << e;
          mask[(size_t)ru * CV + cv] = (unsigned char)bits;
        }
      }
    }
  }
}

// Where the backward kernels take the ReLU mask from (a template parameter: these kernels sit close to instruction-bound, a run-time
// choice per element cost 20-45 % of their time):
enum { BN_NORELU = 0, BN_MASK_RECOMPUTE = 1, BN_MASK_ACT = 2, BN_MASK_BITS = 3 };
//   RECOMPUTE  layers without a residual: y * scale + shift > 0, exactly as the forward computed it (no extra read)
//   ACT        from the saved activation tensor `a`
//   BITS       residual layers: one byte per 16-byte vector written by bn_apply (bit e = element e > 0)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ a,
                                                            int lda, const T* __restrict__ y, int ldy,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int Mg, int C, int rows_per_block, double* partial,
                                                            const unsigned char* __restrict__ mask, int rev) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const int TPC = CV < 256 ? CV : 256;
  const int cv = min(blockIdx.y * TPC + (int)(threadIdx.x % TPC), CV - 1);   // this thread's channel vector (as in channel_reduce2)
  const int g = blockIdx.z, c0 = cv * VEC;
  using P = Pairs<T>;
  f32x2 mu[P::NP], is[P::NP], sc[P::NP], sh[P::NP];
#pragma unroll
  for (int p = 0; p < P::NP; ++p) {
    const int ch = g * C + c0 + 2 * p;
    mu[p] = f32x2{mean[ch], mean[ch + 1]};
    is[p] = f32x2{invstd[ch], invstd[ch + 1]};
    sc[p] = MODE == BN_MASK_RECOMPUTE ? f32x2{scale[ch], scale[ch + 1]} : f32x2{0.f, 0.f};
    sh[p] = MODE == BN_MASK_RECOMPUTE ? f32x2{shift[ch], shift[ch + 1]} : f32x2{0.f, 0.f};
  }
  const int nt = rev >> 1;       // (CSS_BN_NT_BWDR: bit 0 = non-temporal loads of the gradient, bit 1 = of y, bit 2 = of the activation)
  rev &= 1;
  auto f = [&](int r, int c, float* s0, float* s1) {
    Vec16<T> gv, av, yv;
    gv.load(da + (size_t)r * ldda + c, (nt & 1) != 0);
    yv.load(y + (size_t)r * ldy + c, (nt & 2) != 0);
    if (MODE == BN_MASK_ACT) av.load(a + (size_t)r * lda + c, (nt & 4) != 0);
    unsigned bits = 0;
    if (MODE == BN_MASK_BITS) bits = mask[(size_t)r * CV + cv];
#pragma unroll
    for (int p = 0; p < P::NP; ++p) {
      f32x2 dz = P::get(gv, p);
      const f32x2 yy = P::get(yv, p);
      if (MODE == BN_MASK_RECOMPUTE) {
        const f32x2 act = yy * sc[p] + sh[p];
        dz = f32x2{act[0] > 0.f ? dz[0] : 0.f, act[1] > 0.f ? dz[1] : 0.f};
      } else if (MODE == BN_MASK_ACT) {
        const f32x2 act = P::get(av, p);
        dz = f32x2{act[0] > 0.f ? dz[0] : 0.f, act[1] > 0.f ? dz[1] : 0.f};
      } else if (MODE == BN_MASK_BITS) {
        dz = f32x2{keep_if_bit(dz[0], bits, 2 * p), keep_if_bit(dz[1], bits, 2 * p + 1)};
      }
      const f32x2 xh = (yy - mu[p]) * is[p];
      f32x2 a0 = f32x2{s0[2 * p], s0[2 * p + 1]}, a1 = f32x2{s1[2 * p], s1[2 * p + 1]};
      a0 += dz;
      a1 += dz * xh;
      s0[2 * p] = a0[0]; s0[2 * p + 1] = a0[1];
      s1[2 * p] = a1[0]; s1[2 * p + 1] = a1[1];
    }
  };
  channel_reduce2<T>(f, Mg, C, rows_per_block, partial, rev != 0);
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ a, int lda,
                                                           const T* __restrict__ y, int ldy, T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dres, int lddr, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, double count, const double* __restrict__ count_dev,
                                                           int Mg, int C, int rows_per_block, const unsigned char* __restrict__ mask, int rev) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const int TPC = CV < 256 ? CV : 256, RPB = 256 / TPC;
  const int cvi = threadIdx.x % TPC, rg = threadIdx.x / TPC;
  const int cv = blockIdx.y * TPC + cvi;
  if (rg >= RPB || cv >= CV) return;
  const int c = cv * VEC, g = blockIdx.z;
  const float inv_n = (float)(1.0 / (count_dev ? count_dev[g] : count));
  using P = Pairs<T>;
  f32x2 mu[P::NP], is[P::NP], gi[P::NP], m1[P::NP], m2[P::NP], sc[P::NP], sh[P::NP];
#pragma unroll
  for (int p = 0; p < P::NP; ++p) {
    const int ch = g * C + c + 2 * p;
    mu[p] = f32x2{mean[ch], mean[ch + 1]};
    is[p] = f32x2{invstd[ch], invstd[ch + 1]};
    gi[p] = f32x2{gamma[c + 2 * p], gamma[c + 2 * p + 1]} * is[p];
    m1[p] = f32x2{(float)sums[(size_t)g * 2 * C + c + 2 * p], (float)sums[(size_t)g * 2 * C + c + 2 * p + 1]} * inv_n;          // sums are [G][2][C]
    m2[p] = f32x2{(float)sums[(size_t)g * 2 * C + C + c + 2 * p], (float)sums[(size_t)g * 2 * C + C + c + 2 * p + 1]} * inv_n;
    sc[p] = MODE == BN_MASK_RECOMPUTE ? f32x2{scale[ch], scale[ch + 1]} : f32x2{0.f, 0.f};
    sh[p] = MODE == BN_MASK_RECOMPUTE ? f32x2{shift[ch], shift[ch + 1]} : f32x2{0.f, 0.f};
  }
  const int gbase = g * Mg;
  const int nt = rev >> 1;       // (CSS_BN_NT_BWDA: bit 0 = non-temporal loads of the gradient, bit 1 = of y / the activation, bit 2 = non-temporal stores)
  rev &= 1;
  const int bx = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;      // (bn_pass_order: last rows first)
  const int row0 = gbase + bx * rows_per_block, row1 = min(gbase + Mg, row0 + rows_per_block);
  for (int r = row0 + rg; r < row1; r += EW_UNROLL * RPB) {
    Vec16<T> gv[EW_UNROLL], av[EW_UNROLL], yv[EW_UNROLL];
    unsigned bits[EW_UNROLL];
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
      const int ru = r + u * RPB;
      bits[u] = 0;
      if (ru < row1) {
        gv[u].load(da + (size_t)ru * ldda + c, (nt & 1) != 0);
        yv[u].load(y + (size_t)ru * ldy + c, (nt & 2) != 0);
        if (MODE == BN_MASK_ACT) av[u].load(a + (size_t)ru * lda + c, (nt & 2) != 0);
        if (MODE == BN_MASK_BITS) bits[u] = mask[(size_t)ru * CV + cv];
      }
    }
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
      const int ru = r + u * RPB;
      if (ru < row1) {
        Vec16<T> o, dr;
#pragma unroll
        for (int p = 0; p < P::NP; ++p) {
          f32x2 dz = P::get(gv[u], p);
          const f32x2 yy = P::get(yv[u], p);
          if (MODE == BN_MASK_RECOMPUTE) {
            const f32x2 act = yy * sc[p] + sh[p];
            dz = f32x2{act[0] > 0.f ? dz[0] : 0.f, act[1] > 0.f ? dz[1] : 0.f};
          } else if (MODE == BN_MASK_ACT) {
            const f32x2 act = P::get(av[u], p);
            dz = f32x2{act[0] > 0.f ? dz[0] : 0.f, act[1] > 0.f ? dz[1] : 0.f};
          } else if (MODE == BN_MASK_BITS) {
            dz = f32x2{keep_if_bit(dz[0], bits[u], 2 * p), keep_if_bit(dz[1], bits[u], 2 * p + 1)};
          }
          const f32x2 xh = (yy - mu[p]) * is[p];
          P::set(o, p, gi[p] * (dz - m1[p] - xh * m2[p]));
          P::set(dr, p, dz);
        }
        if (nt & 8) {
          o.store_sc1(dy + (size_t)ru * lddy + c);
          if (dres) dr.store_sc1(dres + (size_t)ru * lddr + c);
        } else {
          o.store(dy + (size_t)ru * lddy + c, (nt & 4) != 0);
          if (dres) dr.store(dres + (size_t)ru * lddr + c, (nt & 4) != 0);
        }
      }
    }
  }
}

// ---- launchers -----------------------------------------------------------
// Row order of the streaming passes.  Their input was written a moment ago by a kernel that walks the rows upwards, so what the 256 MB
// Infinity Cache still holds is the END of that tensor: a pass that starts at row 0 misses, and pushes the cached tail out with its own
// traffic before it gets there.  Bit 0: bn_apply walks the rows downwards (its producer is the convolution), bit 1: bn_bwd_reduce
// (producer: the data-gradient convolution), bit 2: bn_bwd_apply (producer of what the cache holds: bn_bwd_reduce itself - if that
// ran downwards, the cache holds the first rows and this pass should walk upwards).  CSS_BN_PASS_ORDER overrides.
static inline int bn_nt_bwd_reduce() {
  static const int v = getenv("CSS_BN_NT_BWDR") ? atoi(getenv("CSS_BN_NT_BWDR")) & 7 : 2;      // (y non-temporal; the gradient stays cached for bn_bwd_apply)
  return v;
}
static inline int bn_nt_bwd_apply() {
  static const int v = getenv("CSS_BN_NT_BWDA") ? atoi(getenv("CSS_BN_NT_BWDA")) & 15 : 11;     // (last readers of the gradient and of y; write-through stores)
  return v;
}
static inline int bn_pass_order() {
  static const int v = getenv("CSS_BN_PASS_ORDER") ? atoi(getenv("CSS_BN_PASS_ORDER")) : 1;      // (measured, profiles/r03_dres_mask_and_bn_order.txt: bit 0 -0.25 ms per step, bits 1 and 2 nothing)
  return v;
}

// Tensors are [M = G*Mg][C]: G statistics groups of Mg rows each (G forward passes batched into one tensor).
// rows per block: every block streams >= 64 KiB (so that the partial rows stay a few % of the tensor), at most ~1024 blocks
static inline int pick_rows_per_block(int Mg, int G, int C, int vec) {
  const int CV = C / vec, TPC = CV < 256 ? CV : 256, RPB = 256 / TPC;
  const int ybl = (CV + TPC - 1) / TPC;
  const long bytes = (long)Mg * C * (16 / vec);
  long want = bytes / (64 * 1024) / ybl;
  static const long total_cap = getenv("CSS_BN_RED_BLOCKS") ? atol(getenv("CSS_BN_RED_BLOCKS")) : 512;   // (one block per CU wins the stand-alone microbenchmark by up to 22 % but loses 0.4 ms in the step, where the tensors come from the Infinity Cache)
  const long cap = total_cap / ((long)ybl * G) > 0 ? total_cap / ((long)ybl * G) : 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  int rpb = cdiv(Mg, want);
  rpb = cdiv(rpb, RPB) * RPB;
  if (rpb < 4 * RPB) rpb = 4 * RPB;
  return rpb;
}
int css_bn_nrb_(int Mg, int G, int C, int dtype) {
  const int vec = dtype == CSS_BF16 ? 8 : 4;
  if (Mg <= 0) return 1;
  return cdiv(Mg, pick_rows_per_block(Mg, G, C, vec));
}

template <typename T>
static int bn_stats_T(const void* y, int Mg, int G, int C, int ld, double* partial, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (C % VEC || ld % VEC) return CSS_ERR_ARG;
  const int CV = C / VEC, TPC = CV < 256 ? CV : 256;
  const int rpb = pick_rows_per_block(Mg, G, C, VEC);
  dim3 g(cdiv(Mg, rpb), cdiv(CV, TPC), G);
  hipLaunchKernelGGL(bn_stats_kernel<T>, g, dim3(256), 0, st, (const T*)y, Mg, C, ld, rpb, partial);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_stats(const void* y, int Mg, int G, int C, int ld, double* partial, int dtype, hipStream_t st) {
  if (Mg <= 0 || G <= 0) return CSS_ERR_ARG;
  return dtype == CSS_BF16 ? bn_stats_T<bf16_t>(y, Mg, G, C, ld, partial, st)
         : dtype == CSS_F32 ? bn_stats_T<float>(y, Mg, G, C, ld, partial, st) : CSS_ERR_DTYPE;
}
int css_launch_bn_reduce(const double* partial, int nrb, int C, int G, double* sums, float* g1, float* g0, int accumulate, double count_local,
                         hipStream_t st) {
  hipLaunchKernelGGL(bn_reduce_kernel, dim3(cdiv(C, S2_CH)), dim3(1024), 0, st, partial, nrb, C, G, sums, g1, g0, accumulate, count_local);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_reduce_finalize(const double* partial, int nrb, int G, double count, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                                  float* shift, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(cdiv(C, S2_CH)), dim3(1024), 0, st, partial, nrb, G, count, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, invstd, scale, shift, C);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_reduce_slabs(const float* partial, int M, int Mg, int G, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                               float* shift, double* sums_out, int C, const void* y, int ldy, int tile_rows, hipStream_t st) {
  if (M <= 0 || Mg < 128 || G <= 0 || (long)Mg * G != M || !y || ldy < C || tile_rows != 256) return CSS_ERR_ARG;
  hipLaunchKernelGGL(bn_reduce_slabs_kernel, dim3(cdiv(C, S2_CH)), dim3(1024), 0, st, partial, 2 * cdiv(M, tile_rows), Mg, G, count, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, invstd, scale, shift, sums_out, C, (const bf16_t*)y, ldy, tile_rows);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_finalize(const double* sums, int G, double count, const double* count_dev, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                           float* shift, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sums, G, count, count_dev, gamma, beta, running_mean, running_var,
                     momentum, eps, mean, invstd, scale, shift, C);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                             float* shift, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, rm, rv, eps, scale, shift, C);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

static inline int ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// rows per block for the elementwise kernels: ~2048 blocks in total, >= EW_UNROLL rows per thread
static inline int pick_rows_ew(int Mg, int G, int C, int vec, int min_rows = EW_UNROLL, long total_ew = 2048) {
  const int CV = C / vec, TPC = CV < 256 ? CV : 256, RPB = 256 / TPC;
  const int ybl = (CV + TPC - 1) / TPC;
  long want = total_ew / ((long)ybl * G);
  if (want < 1) want = 1;
  int rpb = cdiv(Mg, want);
  rpb = cdiv(rpb, RPB) * RPB;
  if (rpb < min_rows * RPB) rpb = min_rows * RPB;
  return rpb;
}

template <typename T>
static int bn_apply_T(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale,
                      const float* shift, int M, int C, int relu, int Mg, unsigned char* mask, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (C % VEC || ldy % VEC || ldo % VEC || (res && ldr % VEC)) return CSS_ERR_ARG;
  const int CV = C / VEC, TPC = CV < 256 ? CV : 256, G = M / Mg;
  // (r02, templated kernel, in the step on one box: 256 blocks 18.1 ms per step, 2048 17.7, 8192 16.7)
  static const long apply_blocks = getenv("CSS_BN_APPLY_BLOCKS") ? atol(getenv("CSS_BN_APPLY_BLOCKS")) : 32768;
  static const int apply_nt = getenv("CSS_BN_NT") ? atoi(getenv("CSS_BN_NT")) & 15 : 11;     // (non-temporal loads of y and of the residual, write-through stores: bn_apply_kernel)
  const int rpb = pick_rows_ew(Mg, G, C, VEC, EW_UNROLL, apply_blocks);
  dim3 g(cdiv(Mg, rpb), cdiv(CV, TPC), G);
#define CSS_BN_APPLY_LAUNCH(MASK, RES, RELU)                                                                                               \
  hipLaunchKernelGGL((bn_apply_kernel<T, MASK, RES, RELU>), g, dim3(256), 0, st, (const T*)y, ldy, (const T*)res, ldr, (T*)out, ldo, scale, shift, \
                     Mg, C, rpb, mask, (bn_pass_order() & 1) | (apply_nt << 1))
  // (the elementwise kernels sit close to instruction-bound: mask / residual / ReLU are compile-time choices)
  if (mask) {
    if (!res || !relu) return CSS_ERR_ARG;                 // the bit mask exists for residual + ReLU layers
    CSS_BN_APPLY_LAUNCH(true, true, true);
  } else if (res) {
    if (relu) CSS_BN_APPLY_LAUNCH(false, true, true);
    else CSS_BN_APPLY_LAUNCH(false, true, false);
  } else {
    if (relu) CSS_BN_APPLY_LAUNCH(false, false, true);
    else CSS_BN_APPLY_LAUNCH(false, false, false);
  }
#undef CSS_BN_APPLY_LAUNCH
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
// bn_apply + ReLU + the 3x3 stride-2 pad-1 max pool that follows the stem's batch norm (resnet.py:186-190; torchvision's bn1 / relu / maxpool) in ONE pass:
// the normalised activation is read by nothing but the pool, so it is never written (round 5: -540 MB per forward pass at c2).  Thread = one pooled
// 16-byte vector; every tap is normalised in fp32, ROUNDED to T (what bn_apply would have stored), and compared exactly like maxpool_fwd_kernel (first
// maximum in (r, s) order wins, NaN propagates): pooled values and arg-max bytes are bit-identical to bn_apply followed by maxpool_fwd.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const T* __restrict__ y, T* __restrict__ out, uint8_t* __restrict__ arg,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, int N, int H, int W, int C,
                                                            int Ho, int Wo, int imgs_per_group, int relu) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * Ho * Wo * CV;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    size_t p = idx / CV;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho), n = (int)(p / Ho);
    const int g = n / imgs_per_group, c = cv * VEC;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = scale[g * C + c + e]; sh[e] = shift[g * C + c + e]; }
    Vec16<T> v[9];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s_;
        ok[r * 3 + s_] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const size_t off = ok[r * 3 + s_] ? ((size_t)(n * H + hi) * W + wi) * C + c : (size_t)c;      // (clamped: all nine requests go out up front)
        v[r * 3 + s_].load(y + off);
      }
    float best[VEC];
    uint8_t bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (!ok[t]) continue;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float f = v[t].f(e) * sc[e] + sh[e];                 // (the same expression as bn_apply_kernel: contracted to one fma there and here)
        if (relu) f = fmaxf(f, 0.f);
        f = ElemT<T>::to_f(ElemT<T>::from_f(f));             // the value bn_apply would have stored
        if (f > best[e] || f != f) { best[e] = f; bi[e] = (uint8_t)t; }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, best[e]);
    const size_t ob = ((size_t)(n * Ho + ho) * Wo + wo) * C + c;
    o.store(out + ob);
    if (arg) {
      union { uint8_t b[VEC]; typename std::conditional<VEC == 8, uint2, uint32_t>::type w; } pk;
#pragma unroll
      for (int e = 0; e < VEC; ++e) pk.b[e] = bi[e];
      *reinterpret_cast<decltype(pk.w)*>(arg + ob) = pk.w;
    }
  }
}
int css_launch_bn_apply_pool(const void* y, void* out, uint8_t* arg, const float* scale, const float* shift, int N, int H, int W, int C, int Ho, int Wo,
                             int G, int relu, int dtype, hipStream_t st) {
  if (N <= 0 || G <= 0 || N % G || H < 1 || W < 1 || Ho < 1 || Wo < 1 || (Ho - 1) * 2 - 1 >= H || (Wo - 1) * 2 - 1 >= W) return CSS_ERR_ARG;
  if (dtype != CSS_BF16 && dtype != CSS_F32) return CSS_ERR_DTYPE;
  const int vec = dtype == CSS_BF16 ? 8 : 4;
  if (C % vec || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return CSS_ERR_ARG;
  const size_t total = (size_t)N * Ho * Wo * (C / vec);
  const size_t nb = (total + 255) / 256;
  const dim3 g((unsigned)(nb < 1 ? 1 : (nb > 65536 ? 65536 : nb)));
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(bn_apply_pool_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)y, (bf16_t*)out, arg, scale, shift, N, H, W, C, Ho, Wo, N / G, relu);
  else
    hipLaunchKernelGGL(bn_apply_pool_kernel<float>, g, dim3(256), 0, st, (const float*)y, (float*)out, arg, scale, shift, N, H, W, C, Ho, Wo, N / G, relu);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_apply(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale,
                        const float* shift, int M, int C, int relu, int Mg, unsigned char* mask, int dtype, hipStream_t st) {
  if (M <= 0) return CSS_OK;
  if (Mg <= 0 || M % Mg) return CSS_ERR_ARG;
  return dtype == CSS_BF16 ? bn_apply_T<bf16_t>(y, ldy, res, ldr, out, ldo, scale, shift, M, C, relu, Mg, mask, st)
         : dtype == CSS_F32 ? bn_apply_T<float>(y, ldy, res, ldr, out, ldo, scale, shift, M, C, relu, Mg, mask, st) : CSS_ERR_DTYPE;
}

template <typename T>
static int bn_bwd_reduce_T(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, const float* mean,
                           const float* invstd, const float* scale, const float* shift, int Mg, int G, int C, int relu, double* partial,
                           const unsigned char* mask, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (C % VEC || ldda % VEC || ldy % VEC || (relu && a && lda % VEC) || (relu && !a && !mask && (!scale || !shift)) || (a && mask)) return CSS_ERR_ARG;
  const int CV = C / VEC, TPC = CV < 256 ? CV : 256;
  const int rpb = pick_rows_per_block(Mg, G, C, VEC);
  dim3 g(cdiv(Mg, rpb), cdiv(CV, TPC), G);
  const int mode = !relu ? BN_NORELU : mask ? BN_MASK_BITS : a ? BN_MASK_ACT : BN_MASK_RECOMPUTE;
#define CSS_BN_RED_LAUNCH(MODE)                                                                                                      \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MODE>), g, dim3(256), 0, st, (const T*)da, ldda, (const T*)a, lda, (const T*)y, ldy, mean, invstd, \
                     scale, shift, Mg, C, rpb, partial, mask, ((bn_pass_order() >> 1) & 1) | (bn_nt_bwd_reduce() << 1))
  if (mode == BN_NORELU) CSS_BN_RED_LAUNCH(BN_NORELU);
  else if (mode == BN_MASK_RECOMPUTE) CSS_BN_RED_LAUNCH(BN_MASK_RECOMPUTE);
  else if (mode == BN_MASK_ACT) CSS_BN_RED_LAUNCH(BN_MASK_ACT);
  else CSS_BN_RED_LAUNCH(BN_MASK_BITS);
#undef CSS_BN_RED_LAUNCH
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_bwd_reduce(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, const float* mean,
                             const float* invstd, const float* scale, const float* shift, int Mg, int G, int C, int relu, double* partial,
                             const unsigned char* mask, int dtype, hipStream_t st) {
  if (Mg <= 0 || G <= 0) return CSS_ERR_ARG;
  return dtype == CSS_BF16 ? bn_bwd_reduce_T<bf16_t>(da, ldda, a, lda, y, ldy, mean, invstd, scale, shift, Mg, G, C, relu, partial, mask, st)
         : dtype == CSS_F32 ? bn_bwd_reduce_T<float>(da, ldda, a, lda, y, ldy, mean, invstd, scale, shift, Mg, G, C, relu, partial, mask, st)
                            : CSS_ERR_DTYPE;
}

template <typename T>
static int bn_bwd_apply_T(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, void* dy, int lddy,
                          void* dres, int lddr, const float* mean, const float* invstd, const float* gamma, const double* sums,
                          const float* scale, const float* shift, double count, const double* count_dev, int M, int C, int relu, int Mg,
                          const unsigned char* mask, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (C % VEC || ldda % VEC || ldy % VEC || lddy % VEC || (relu && a && lda % VEC) || (dres && lddr % VEC) ||
      (relu && !a && !mask && (!scale || !shift)) || (a && mask))
    return CSS_ERR_ARG;
  const int CV = C / VEC, TPC = CV < 256 ? CV : 256, G = M / Mg;
  // amortise the 7-coefficient prologue (measured: 44 -> 30 us at 135200x128), and ONE block per CU: with 3-5 streams per block the
  // backward kernel is 5-25 % faster on 256 blocks than on 2048 (bn_bench.py; the forward apply kernel is the opposite)
  // (round 6, after the non-temporal loads: 1024 blocks here, 32768 in bn_apply, 512 in bn_bwd_reduce: -1.7 ms per c2 step together, profiles/r06_bn_blocks_ab.txt)
  static const long bwd_blocks = getenv("CSS_BN_BWD_EW_BLOCKS") ? atol(getenv("CSS_BN_BWD_EW_BLOCKS")) : 1024;
  const int rpb = pick_rows_ew(Mg, G, C, VEC, 16, bwd_blocks);
  dim3 g(cdiv(Mg, rpb), cdiv(CV, TPC), G);
  const int mode = !relu ? BN_NORELU : mask ? BN_MASK_BITS : a ? BN_MASK_ACT : BN_MASK_RECOMPUTE;
#define CSS_BN_APP_LAUNCH(MODE)                                                                                                       \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MODE>), g, dim3(256), 0, st, (const T*)da, ldda, (const T*)a, lda, (const T*)y, ldy, (T*)dy, lddy, \
                     (T*)dres, lddr, mean, invstd, gamma, sums, scale, shift, count, count_dev, Mg, C, rpb, mask, ((bn_pass_order() >> 2) & 1) | (bn_nt_bwd_apply() << 1))
  if (mode == BN_NORELU) CSS_BN_APP_LAUNCH(BN_NORELU);
  else if (mode == BN_MASK_RECOMPUTE) CSS_BN_APP_LAUNCH(BN_MASK_RECOMPUTE);
  else if (mode == BN_MASK_ACT) CSS_BN_APP_LAUNCH(BN_MASK_ACT);
  else CSS_BN_APP_LAUNCH(BN_MASK_BITS);
#undef CSS_BN_APP_LAUNCH
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_bwd_apply(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, void* dy, int lddy,
                            void* dres, int lddr, const float* mean, const float* invstd, const float* gamma, const double* sums,
                            const float* scale, const float* shift, double count, const double* count_dev, int M, int C, int relu, int Mg,
                            const unsigned char* mask, int dtype, hipStream_t st) {
  if (M <= 0) return CSS_OK;
  if (Mg <= 0 || M % Mg) return CSS_ERR_ARG;
  return dtype == CSS_BF16
             ? bn_bwd_apply_T<bf16_t>(da, ldda, a, lda, y, ldy, dy, lddy, dres, lddr, mean, invstd, gamma, sums, scale, shift, count, count_dev, M, C, relu, Mg, mask, st)
         : dtype == CSS_F32
             ? bn_bwd_apply_T<float>(da, ldda, a, lda, y, ldy, dy, lddy, dres, lddr, mean, invstd, gamma, sums, scale, shift, count, count_dev, M, C, relu, Mg, mask, st)
             : CSS_ERR_DTYPE;
}
