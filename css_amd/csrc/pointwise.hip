// HBM-bound data-movement kernels of the network path (NHWC):
//   max-pool 3x3/s2 (deeplabv3.py:153), bilinear resize align_corners=True
//   (deeplabv3.py:164, ddp_model.py:111,113,141,144), global average pool + broadcast
//   (aspp.py:27-38), channel concat / slice (aspp.py:71, deeplabv3.py:165-166),
//   NCHW<->NHWC image staging, dtype casts and weight re-layout, fused SGD(nesterov)+EMA
//   (mix_label.py:96-97,194-195; ddp_model.py:93-97).
#include "common.h"
#include <type_traits>
#include <cstdlib>

static inline int ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
#define GRID_STRIDE(idx, total) \
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

// ---- max pool -----------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, uint8_t* __restrict__ arg,
                                                          int N, int H, int W, int C, int Ho, int Wo, int ks, int stride,
                                                          int pad) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * Ho * Wo * CV;
  GRID_STRIDE(idx, total) {
    const int cv = (int)(idx % CV);
    size_t p = idx / CV;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho), n = (int)(p / Ho);
    float best[VEC];
    uint8_t bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int r = 0; r < ks; ++r) {
      const int hi = ho * stride - pad + r;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int s = 0; s < ks; ++s) {
        const int wi = wo * stride - pad + s;
        if ((unsigned)wi >= (unsigned)W) continue;
        Vec16<T> v;
        v.load(x + ((size_t)(n * H + hi) * W + wi) * C + cv * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = v.f(e);
          if (f > best[e] || f != f) { best[e] = f; bi[e] = (uint8_t)(r * ks + s); }
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, best[e]);
    const size_t ob = ((size_t)(n * Ho + ho) * Wo + wo) * C + cv * VEC;
    o.store(out + ob);
    if (arg) {          // (one VEC-byte store: ob is a multiple of VEC)
      union { uint8_t b[VEC]; typename std::conditional<VEC == 8, uint2, uint32_t>::type w; } pk;
#pragma unroll
      for (int e = 0; e < VEC; ++e) pk.b[e] = bi[e];
      *reinterpret_cast<decltype(pk.w)*>(arg + ob) = pk.w;
    }
  }
}

// gather form: each input element sums the gradients of the windows whose arg-max it is
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int ks,
                                                          int stride, int pad) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * H * W * CV;
  GRID_STRIDE(idx, total) {
    const int cv = (int)(idx % CV);
    size_t p = idx / CV;
    const int wi = (int)(p % W);
    p /= W;
    const int hi = (int)(p % H), n = (int)(p / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int r = 0; r < ks; ++r) {
      const int th = hi + pad - r;
      if (th < 0 || th % stride) continue;
      const int ho = th / stride;
      if (ho >= Ho) continue;
      for (int s = 0; s < ks; ++s) {
        const int tw = wi + pad - s;
        if (tw < 0 || tw % stride) continue;
        const int wo = tw / stride;
        if (wo >= Wo) continue;
        const size_t ob = ((size_t)(n * Ho + ho) * Wo + wo) * C + cv * VEC;
        Vec16<T> g;
        g.load(dout + ob);
        const uint8_t want = (uint8_t)(r * ks + s);
        union { uint8_t b[VEC]; typename std::conditional<VEC == 8, uint2, uint32_t>::type w; } pk;      // (one VEC-byte load instead of VEC)
        pk.w = *reinterpret_cast<const decltype(pk.w)*>(arg + ob);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (pk.b[e] == want) acc[e] += g.f(e);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(dx + idx * VEC);
  }
}

// ---- bilinear, align_corners=True ---------------------------------------
// src index of dst d: f = d * (S-1)/(D-1) (0 if D == 1); i0 = (int)f, i1 = min(i0+1, S-1), w1 = f - i0
struct Lin { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin lin_coord(int d, int S, float scale) {
  Lin l;
  float f = scale * (float)d;
  l.i0 = (int)f;
  if (l.i0 > S - 1) l.i0 = S - 1;
  l.i1 = l.i0 + (l.i0 < S - 1 ? 1 : 0);
  l.w1 = f - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ out, int ldo, int N,
                                                           int Hs, int Ws, int C, int Hd, int Wd, float sh, float sw) {
  const size_t total = (size_t)N * Hd * Wd * C;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % C);
    size_t p = idx / C;
    const int wd = (int)(p % Wd);
    p /= Wd;
    const int hd = (int)(p % Hd), n = (int)(p / Hd);
    const Lin ly = lin_coord(hd, Hs, sh), lx = lin_coord(wd, Ws, sw);
    const TI* b = x + (size_t)n * Hs * Ws * ldx + c;
    const float v00 = (float)b[((size_t)ly.i0 * Ws + lx.i0) * ldx], v01 = (float)b[((size_t)ly.i0 * Ws + lx.i1) * ldx];
    const float v10 = (float)b[((size_t)ly.i1 * Ws + lx.i0) * ldx], v11 = (float)b[((size_t)ly.i1 * Ws + lx.i1) * ldx];
    const float v = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
    out[((size_t)(n * Hd + hd) * Wd + wd) * ldo + c] = (TO)v;
  }
}

// gather form of the adjoint: each source element collects from the destination range that references it
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const TI* __restrict__ dout, int ldo, TO* __restrict__ dx, int ldx, int N,
                                                           int Hs, int Ws, int C, int Hd, int Wd, float sh, float sw) {
  const size_t total = (size_t)N * Hs * Ws * C;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % C);
    size_t p = idx / C;
    const int ws = (int)(p % Ws);
    p /= Ws;
    const int hs = (int)(p % Hs), n = (int)(p / Hs);
    int h_lo, h_hi, w_lo, w_hi;
    if (sh > 0.f) {
      h_lo = max(0, (int)floorf((float)(hs - 1) * ish) - 1);
      h_hi = min(Hd - 1, (int)ceilf((float)(hs + 1) * ish) + 1);
    } else { h_lo = 0; h_hi = Hd - 1; }
    if (sw > 0.f) {
      w_lo = max(0, (int)floorf((float)(ws - 1) * isw) - 1);
      w_hi = min(Wd - 1, (int)ceilf((float)(ws + 1) * isw) + 1);
    } else { w_lo = 0; w_hi = Wd - 1; }
    float acc = 0.f;
    for (int hd = h_lo; hd <= h_hi; ++hd) {
      const Lin ly = lin_coord(hd, Hs, sh);
      float wy = 0.f;
      if (ly.i0 == hs) wy += ly.w0;
      if (ly.i1 == hs) wy += ly.w1;
      if (wy == 0.f) continue;
      const TI* row = dout + ((size_t)(n * Hd + hd) * Wd) * ldo + c;
      for (int wd = w_lo; wd <= w_hi; ++wd) {
        const Lin lx = lin_coord(wd, Ws, sw);
        float wx = 0.f;
        if (lx.i0 == ws) wx += lx.w0;
        if (lx.i1 == ws) wx += lx.w1;
        if (wx != 0.f) acc += wy * wx * (float)row[(size_t)wd * ldo];
      }
    }
    dx[((size_t)(n * Hs + hs) * Ws + ws) * ldx + c] = (TO)acc;
  }
}

// 16-byte vector forms (same arithmetic per element) for the channel-rich maps: the ASPP feature map up-sampled into the
// decoder is 0.27 GB per pass and the scalar kernels above moved it at 0.6 TB/s
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_vec_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, int N, int Hs,
                                                               int Ws, int C, int Hd, int Wd, float sh, float sw) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * Hd * Wd * CV;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % CV) * VEC;
    size_t p = idx / CV;
    const int wd = (int)(p % Wd);
    p /= Wd;
    const int hd = (int)(p % Hd), n = (int)(p / Hd);
    const Lin ly = lin_coord(hd, Hs, sh), lx = lin_coord(wd, Ws, sw);
    const T* b = x + (size_t)n * Hs * Ws * ldx + c;
    Vec16<T> v00, v01, v10, v11, o;
    v00.load(b + ((size_t)ly.i0 * Ws + lx.i0) * ldx);
    v01.load(b + ((size_t)ly.i0 * Ws + lx.i1) * ldx);
    v10.load(b + ((size_t)ly.i1 * Ws + lx.i0) * ldx);
    v11.load(b + ((size_t)ly.i1 * Ws + lx.i1) * ldx);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      o.set(e, ly.w0 * (lx.w0 * v00.f(e) + lx.w1 * v01.f(e)) + ly.w1 * (lx.w0 * v10.f(e) + lx.w1 * v11.f(e)));
    o.store(out + ((size_t)(n * Hd + hd) * Wd + wd) * ldo + c);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_vec_kernel(const T* __restrict__ dout, int ldo, T* __restrict__ dx, int ldx, int N, int Hs,
                                                               int Ws, int C, int Hd, int Wd, float sh, float sw) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * Hs * Ws * CV;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % CV) * VEC;
    size_t p = idx / CV;
    const int ws = (int)(p % Ws);
    p /= Ws;
    const int hs = (int)(p % Hs), n = (int)(p / Hs);
    int h_lo, h_hi, w_lo, w_hi;
    if (sh > 0.f) {
      h_lo = max(0, (int)floorf((float)(hs - 1) * ish) - 1);
      h_hi = min(Hd - 1, (int)ceilf((float)(hs + 1) * ish) + 1);
    } else { h_lo = 0; h_hi = Hd - 1; }
    if (sw > 0.f) {
      w_lo = max(0, (int)floorf((float)(ws - 1) * isw) - 1);
      w_hi = min(Wd - 1, (int)ceilf((float)(ws + 1) * isw) + 1);
    } else { w_lo = 0; w_hi = Wd - 1; }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int hd = h_lo; hd <= h_hi; ++hd) {
      const Lin ly = lin_coord(hd, Hs, sh);
      float wy = 0.f;
      if (ly.i0 == hs) wy += ly.w0;
      if (ly.i1 == hs) wy += ly.w1;
      if (wy == 0.f) continue;
      const T* row = dout + ((size_t)(n * Hd + hd) * Wd) * ldo + c;
      for (int wd = w_lo; wd <= w_hi; ++wd) {
        const Lin lx = lin_coord(wd, Ws, sw);
        float wx = 0.f;
        if (lx.i0 == ws) wx += lx.w0;
        if (lx.i1 == ws) wx += lx.w1;
        if (wx != 0.f) {
          Vec16<T> v;
          v.load(row + (size_t)wd * ldo);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += wy * wx * v.f(e);
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(dx + ((size_t)(n * Hs + hs) * Ws + ws) * ldx + c);
  }
}

// ---- global average pool / broadcast ------------------------------------
// out[n][c] = scale * sum_{hw} x[n][hw][c]     (scale = 1/HW for the pool, 1 for the broadcast adjoint)
template <typename T>
__global__ __launch_bounds__(256) void spatial_sum_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int HW, int C,
                                                          float scale) {
  // block = 8 channel-vectors (16 B each) x 32 pixel partitions; 4 independent loads in flight per thread
  constexpr int VEC = 16 / sizeof(T);
  const int n = blockIdx.y;
  const int cv = blockIdx.x * 8 + (threadIdx.x & 7);
  const int part = threadIdx.x >> 3;
  const int c = cv * VEC;
  float acc[4][VEC];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[u][e] = 0.f;
  if (c < C) {
    const T* base = x + (size_t)n * HW * ldx + c;
    int p = part;
    for (; p + 96 < HW; p += 128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        Vec16<T> v;
        v.load(base + (size_t)(p + 32 * u) * ldx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[u][e] += v.f(e);
      }
    }
    for (; p < HW; p += 32) {
      Vec16<T> v;
      v.load(base + (size_t)p * ldx);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[0][e] += v.f(e);
    }
  }
  __shared__ float red[32][8 * VEC + 1];
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[part][(threadIdx.x & 7) * VEC + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
  __syncthreads();
  if (threadIdx.x < 8 * VEC) {
    const int cc = blockIdx.x * 8 * VEC + threadIdx.x;
    if (cc < C) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x];
      out[(size_t)n * C + cc] = (T)(t * scale);
    }
  }
}
// out[n][hw][c] = scale * x[n][c]
template <typename T>
__global__ __launch_bounds__(256) void spatial_bcast_kernel(const T* __restrict__ x, T* __restrict__ out, int ldo, int N, int HW,
                                                            int C, float scale) {
  const size_t total = (size_t)N * HW * C;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % C);
    const size_t p = idx / C;
    const int n = (int)(p / HW);
    out[p * ldo + c] = (T)((float)x[(size_t)n * C + c] * scale);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void spatial_bcast_vec_kernel(const T* __restrict__ x, T* __restrict__ out, int ldo, int N, int HW, int C,
                                                                float scale) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * HW * CV;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % CV) * VEC;
    const size_t p = idx / CV;
    const int n = (int)(p / HW);
    Vec16<T> v, o;
    v.load(x + (size_t)n * C + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, v.f(e) * scale);
    o.store(out + p * ldo + c);
  }
}

// ---- column sum: out[c] += sum_m x[m][c]  (bias gradients) ---------------
// stage 1: one partial row ws[blockIdx.y][C] per block of rows_per_block rows (plain stores); stage 2 adds the rows in block order.
// (fp32 atomics onto out[] would add in arrival order: the bias gradients would differ in their last bits from run to run.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, long M, int C, long rows_per_block, float* __restrict__ ws) {
  // C <= 256 (the class logits: C = 21 / 19): RG = 256 / C rows per pass, thread t = (row t / C, channel t % C) - with ld == C the RG x C
  // threads of a pass read one contiguous run (the per-64-channel form left 43 of 64 lanes idle at C = 21: 197 us for 22 MB)
  const int RG = 256 / C, tid = threadIdx.x;
  const int c = tid % C, rg = tid / C;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc = 0.f;
  if (rg < RG) {
    long r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG)        // four rows in flight
      acc += ((float)x[(size_t)r * ld + c] + (float)x[(size_t)(r + RG) * ld + c]) +
             ((float)x[(size_t)(r + 2 * RG) * ld + c] + (float)x[(size_t)(r + 3 * RG) * ld + c]);
    for (; r < r1; r += RG) acc += (float)x[(size_t)r * ld + c];
  }
  __shared__ float red[256];
  red[tid] = rg < RG ? acc : 0.f;
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int g = 0; g < RG; ++g) t += red[g * C + tid];       // (row groups in order)
    ws[(size_t)blockIdx.y * C + tid] = t;
  }
}
// any C (> 256 without the vector layout): 64 channels per block column, four row partitions
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, int ld, long M, int C, long rows_per_block, float* __restrict__ ws) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc = 0.f;
  if (c < C)
    for (long r = r0 + part; r < r1; r += 4) acc += (float)x[(size_t)r * ld + c];
  __shared__ float red[4][64];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && c < C)
    ws[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// the same on 16-byte vectors (C and ld multiples of the vector, C / VEC <= 256 and a power of two or a divisor of 256): thread = one
// channel vector of every (256 / CV)-th row, four rows in flight
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int ld, long M, int C, long rows_per_block, float* __restrict__ ws) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC, RG = 256 / CV;
  const int cv = threadIdx.x % CV, rg = threadIdx.x / CV;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (rg < RG) {
    long r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG) {
      Vec16<T> v0, v1, v2, v3;
      v0.load(x + (size_t)r * ld + cv * VEC); v1.load(x + (size_t)(r + RG) * ld + cv * VEC);
      v2.load(x + (size_t)(r + 2 * RG) * ld + cv * VEC); v3.load(x + (size_t)(r + 3 * RG) * ld + cv * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += (v0.f(e) + v1.f(e)) + (v2.f(e) + v3.f(e));
    }
    for (; r < r1; r += RG) {
      Vec16<T> v0;
      v0.load(x + (size_t)r * ld + cv * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += v0.f(e);
    }
  }
  __shared__ float red[256 * VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = rg < RG ? acc[e] : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    const int v = i / VEC, e = i - v * VEC;
    float t = 0.f;
    for (int g = 0; g < RG; ++g) t += red[(g * CV + v) * VEC + e];        // (row groups in order)
    ws[(size_t)blockIdx.y * C + i] = t;
  }
}
// out[c] += the partial rows in row order: 16 partitions x 64 channels per block, combined in a fixed tree
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ ws, int nrows, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int r = part; r < nrows; r += 16) acc += ws[(size_t)r * C + c];
  __shared__ float red[16][64];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
    out[c] += t;
  }
}

// ---- strided channel copy (concat / slice), optional cast ---------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy_channels_kernel(const TI* __restrict__ src, int lds, TO* __restrict__ dst, int ldd, size_t M,
                                                            int C) {
  const size_t total = M * (size_t)C;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % C);
    const size_t r = idx / C;
    dst[r * ldd + c] = (TO)(float)src[r * lds + c];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_vec_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd,
                                                                size_t M, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = M * (size_t)CV;
  GRID_STRIDE(idx, total) {
    const int cv = (int)(idx % CV);
    const size_t r = idx / CV;
    *reinterpret_cast<uint4*>(dst + r * ldd + cv * VEC) = *reinterpret_cast<const uint4*>(src + r * lds + cv * VEC);
  }
}

// ---- NCHW fp32 image -> NHWC (channel-padded) T -------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int HW,
                                                           int Cpad) {
  const size_t total = (size_t)N * HW * Cpad;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % Cpad);
    const size_t p = idx / Cpad;
    const int n = (int)(p / HW);
    const int hw = (int)(p - (size_t)n * HW);
    out[idx] = c < C ? (T)x[((size_t)n * C + c) * HW + hw] : (T)0.f;
  }
}

// the network input (C = 3 planes, Cpad = one 16-byte vector per pixel): thread = one pixel, plane reads coalesced across the wave, one
// 16-byte store (the element-per-thread kernel above: 70 us per 16 x 3 x 513^2 image batch, 2-byte stores and two 64-bit divisions each)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_vec_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int HW) {
  constexpr int VEC = 16 / sizeof(T);
  const size_t total = (size_t)N * HW;
  GRID_STRIDE(p, total) {
    const int n = (int)(p / HW);
    const int hw = (int)(p - (size_t)n * HW);
    Vec16<T> o;
#pragma unroll
    for (int c = 0; c < VEC; ++c) o.set(c, c < C ? x[((size_t)n * C + c) * HW + hw] : 0.f);
    o.store(out + p * VEC);
  }
}

// ---- weight re-layout ----------------------------------------------------
// master fp32 [Cout][taps][Cin] -> T [Cout][taps][CinPad]  (fwd layout; CinPad >= Cin zero-filled)
template <typename T>
__global__ __launch_bounds__(256) void weight_fwd_layout_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int taps,
                                                                int Cin, int CinPad) {
  const size_t total = (size_t)Cout * taps * CinPad;
  GRID_STRIDE(idx, total) {
    const int c = (int)(idx % CinPad);
    const size_t kt = idx / CinPad;
    out[idx] = c < Cin ? (T)w[kt * Cin + c] : (T)0.f;
  }
}
// master fp32 [Cout][taps][Cin] -> T [Cin][taps][Cout]   (dgrad layout)
template <typename T>
__global__ __launch_bounds__(256) void weight_dgrad_layout_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int taps,
                                                                  int Cin) {
  __shared__ float tile[32][33];
  // grid: (ceil(Cin/32), ceil(Cout/32), taps); block 32x8
  const int t = blockIdx.z;
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int k = k0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (k < Cout && c < Cin) ? w[((size_t)k * taps + t) * Cin + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, k = k0 + threadIdx.x;
    if (c < Cin && k < Cout) out[((size_t)c * taps + t) * Cout + k] = (T)tile[threadIdx.x][j];
  }
}

// All dgrad layouts of a flat parameter buffer in ONE launch.  desc[l] = {src_off, dst_off, Cout, taps, Cin, first_tile}
// (element offsets into flat / out); block b handles 32x32 tile (b - first_tile) of the layer that owns it.
template <typename T>
__global__ __launch_bounds__(256) void weight_dgrad_layout_batched_kernel(const float* __restrict__ flat, T* __restrict__ out,
                                                                          const long* __restrict__ desc, int n_layers) {
  __shared__ float tile[32][33];
  const long b = blockIdx.x;
  int lo = 0, hi = n_layers - 1;
  while (lo < hi) {   // last layer with first_tile <= b
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 6 + 5] <= b) lo = mid; else hi = mid - 1;
  }
  const long* d = desc + lo * 6;
  const float* w = flat + d[0];
  T* o = out + d[1];
  const int Cout = (int)d[2], taps = (int)d[3], Cin = (int)d[4];
  long tix = b - d[5];
  const int tc = (Cin + 31) / 32, tk = (Cout + 31) / 32;
  const int c0 = (int)(tix % tc) * 32; tix /= tc;
  const int k0 = (int)(tix % tk) * 32;
  const int t = (int)(tix / tk);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int k = k0 + j, c = c0 + tx;
    tile[j][tx] = (k < Cout && c < Cin) ? w[((size_t)k * taps + t) * Cin + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, k = k0 + tx;
    if (c < Cin && k < Cout) o[((size_t)c * taps + t) * Cout + k] = (T)tile[tx][j];
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ out, size_t n) {
  GRID_STRIDE(idx, n) out[idx] = (TO)(float)x[idx];
}

// ---- fused SGD(nesterov, weight decay) + EMA teacher update ---------------
// torch.optim.SGD semantics (mix_label.py:96-97): d = g + wd*p; buf = first ? d : mom*buf + d;
// p -= lr * (d + mom*buf); then ema = decay*ema + (1-decay)*p (ddp_model.py:93-97).
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                      float* __restrict__ ema, size_t n, float lr, float momentum, float wd,
                                                      int first, float decay, float grad_scale, const float* __restrict__ skip_flag) {
  // skip_flag (device, optional): non-zero = the gradients of this step are known to be invalid (the ranks agreed on a violated
  // bucket plan, train_step.py) - weights, momentum and the EMA teacher stay untouched, decided on the device without a host round trip
  if (skip_flag && *skip_flag != 0.f) return;
  const size_t n4 = n / 4;
  GRID_STRIDE(idx, n4) {
    float4 pv = reinterpret_cast<float4*>(p)[idx];
    const float4 gv = reinterpret_cast<const float4*>(g)[idx];
    float4 bv = first ? make_float4(0, 0, 0, 0) : reinterpret_cast<float4*>(buf)[idx];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = ge[e] * grad_scale + wd * pe[e];
      be[e] = first ? d : momentum * be[e] + d;
      pe[e] -= lr * (d + momentum * be[e]);
    }
    reinterpret_cast<float4*>(p)[idx] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(buf)[idx] = make_float4(be[0], be[1], be[2], be[3]);
    if (ema) {
      float4 ev = reinterpret_cast<float4*>(ema)[idx];
      ev.x = decay * ev.x + (1.f - decay) * pe[0];
      ev.y = decay * ev.y + (1.f - decay) * pe[1];
      ev.z = decay * ev.z + (1.f - decay) * pe[2];
      ev.w = decay * ev.w + (1.f - decay) * pe[3];
      reinterpret_cast<float4*>(ema)[idx] = ev;
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float d = g[i] * grad_scale + wd * p[i];
    const float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    const float pn = p[i] - lr * (d + momentum * b);
    p[i] = pn;
    if (ema) ema[i] = decay * ema[i] + (1.f - decay) * pn;
  }
}
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, size_t n, float decay) {
  GRID_STRIDE(idx, n) ema[idx] = decay * ema[idx] + (1.f - decay) * p[idx];
}

// ---- launchers -----------------------------------------------------------
#define DISPATCH_T(dtype, CALL)                  \
  do {                                           \
    if ((dtype) == CSS_BF16) { using T = bf16_t; CALL; } \
    else if ((dtype) == CSS_F32) { using T = float; CALL; } \
    else return CSS_ERR_DTYPE;                   \
  } while (0)

// The network's only pooling (3x3, stride 2, pad 1: resnet.py:190 / torchvision's maxpool) without the generic kernel's data-dependent loop: an input
// row is in the windows of output row (hi + 1) / 2 (tap 0 if hi is odd, tap 1 if even) and, for odd hi, of (hi - 1) / 2 (tap 2); the same for columns -
// at most 2 x 2 windows.  All four (gradient vector, arg-max bytes) pairs are requested up front (predicated, clamped addresses), then summed in the
// generic kernel's order (r ascending, s ascending: bit-identical results).  r05: 250 -> see DESIGN.md 3e (the generic form ran at 1.35 TB/s: a chain of
// dependent loads behind divergent branches, two integer divisions per tap).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ arg, T* __restrict__ dx, int N, int H,
                                                               int W, int C, int Ho, int Wo) {
  constexpr int VEC = 16 / sizeof(T);
  const int CV = C / VEC;
  const size_t total = (size_t)N * H * W * CV;
  GRID_STRIDE(idx, total) {
    const int cv = (int)(idx % CV);
    size_t p = idx / CV;
    const int wi = (int)(p % W);
    p /= W;
    const int hi = (int)(p % H), n = (int)(p / H);
    // candidate (output index, tap) pairs per axis, ascending tap: a = ((x + 1) >> 1, tap (x & 1) ? 0 : 1), b = ((x - 1) >> 1, tap 2) for odd x only
    const int ho_[2] = {(hi + 1) >> 1, (hi - 1) >> 1}, wo_[2] = {(wi + 1) >> 1, (wi - 1) >> 1};
    const int tr[2] = {(hi & 1) ? 0 : 1, 2}, ts[2] = {(wi & 1) ? 0 : 1, 2};
    const bool okh[2] = {ho_[0] < Ho, (hi & 1) && ho_[1] >= 0 && ho_[1] < Ho}, okw[2] = {wo_[0] < Wo, (wi & 1) && wo_[1] >= 0 && wo_[1] < Wo};
    Vec16<T> g[2][2];
    typedef typename std::conditional<VEC == 8, uint2, uint32_t>::type argw_t;
    union AB { uint8_t b[VEC]; argw_t w; } ab[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = okh[i] && okw[j];
        const size_t ob = ok ? ((size_t)(n * Ho + ho_[i]) * Wo + wo_[j]) * C + cv * VEC : (size_t)cv * VEC;      // (clamped: always a valid address)
        g[i][j].load(dout + ob);
        ab[i][j].w = *reinterpret_cast<const argw_t*>(arg + ob);
        if (!ok) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) ab[i][j].b[e] = 0xFF;
        }
      }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint8_t want = (uint8_t)(tr[i] * 3 + ts[j]);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (ab[i][j].b[e] == want) acc[e] += g[i][j].f(e);
      }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(dx + idx * VEC);
  }
}

int css_launch_maxpool_fwd(const void* x, void* out, uint8_t* arg, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride,
                           int pad, int dtype, hipStream_t st) {
  DISPATCH_T(dtype, {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC) return CSS_ERR_ARG;
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid((size_t)N * Ho * Wo * (C / VEC))), dim3(256), 0, st, (const T*)x, (T*)out,
                       arg, N, H, W, C, Ho, Wo, ks, stride, pad);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_maxpool_bwd(const void* dout, const uint8_t* arg, void* dx, int N, int H, int W, int C, int Ho, int Wo, int ks,
                           int stride, int pad, int dtype, hipStream_t st) {
  DISPATCH_T(dtype, {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC) return CSS_ERR_ARG;
    static const bool generic = getenv("CSS_MAXPOOL_BWD_GENERIC") != nullptr;
    if (ks == 3 && stride == 2 && pad == 1 && !generic)
      hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<T>, dim3(ew_grid((size_t)N * H * W * (C / VEC))), dim3(256), 0, st, (const T*)dout, arg, (T*)dx, N, H, W,
                         C, Ho, Wo);
    else
      hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid((size_t)N * H * W * (C / VEC))), dim3(256), 0, st, (const T*)dout, arg,
                         (T*)dx, N, H, W, C, Ho, Wo, ks, stride, pad);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

static inline float ac_scale(int S, int D) { return D > 1 ? (float)(S - 1) / (float)(D - 1) : 0.f; }

// dtype_in/out: CSS_F32 or CSS_BF16 (any combination)
int css_launch_bilinear(const void* x, int ldx, void* out, int ldo, int N, int Hs, int Ws, int C, int Hd, int Wd, int dtype_in,
                        int dtype_out, int backward, hipStream_t st) {
  const float sh = ac_scale(Hs, Hd), sw = ac_scale(Ws, Wd);
#define BIL(TI, TO)                                                                                                        \
  do {                                                                                                                     \
    if (!backward)                                                                                                         \
      hipLaunchKernelGGL((bilinear_fwd_kernel<TI, TO>), dim3(ew_grid((size_t)N * Hd * Wd * C)), dim3(256), 0, st,          \
                         (const TI*)x, ldx, (TO*)out, ldo, N, Hs, Ws, C, Hd, Wd, sh, sw);                                  \
    else /* x = dout [N,Hd,Wd,ldx], out = dx [N,Hs,Ws,ldo] */                                                              \
      hipLaunchKernelGGL((bilinear_bwd_kernel<TI, TO>), dim3(ew_grid((size_t)N * Hs * Ws * C)), dim3(256), 0, st,          \
                         (const TI*)x, ldx, (TO*)out, ldo, N, Hs, Ws, C, Hd, Wd, sh, sw);                                  \
  } while (0)
  if (dtype_in == dtype_out && (dtype_in == CSS_BF16 || dtype_in == CSS_F32)) {      // same type, vectorisable layout: 16-byte form
    const int vec = dtype_in == CSS_BF16 ? 8 : 4;
    if (C % vec == 0 && ldx % vec == 0 && ldo % vec == 0 && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15)) {
      DISPATCH_T(dtype_in, {
        if (!backward)
          hipLaunchKernelGGL(bilinear_fwd_vec_kernel<T>, dim3(ew_grid((size_t)N * Hd * Wd * (C / vec))), dim3(256), 0, st, (const T*)x, ldx, (T*)out,
                             ldo, N, Hs, Ws, C, Hd, Wd, sh, sw);
        else
          hipLaunchKernelGGL(bilinear_bwd_vec_kernel<T>, dim3(ew_grid((size_t)N * Hs * Ws * (C / vec))), dim3(256), 0, st, (const T*)x, ldx, (T*)out,
                             ldo, N, Hs, Ws, C, Hd, Wd, sh, sw);
      });
      CSS_CHECK_LAUNCH();
      return CSS_OK;
    }
  }
  if (dtype_in == CSS_F32 && dtype_out == CSS_F32) BIL(float, float);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_BF16) BIL(bf16_t, bf16_t);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_F32) BIL(bf16_t, float);
  else if (dtype_in == CSS_F32 && dtype_out == CSS_BF16) BIL(float, bf16_t);
  else return CSS_ERR_DTYPE;
#undef BIL
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_spatial_sum(const void* x, int ldx, void* out, int N, int HW, int C, float scale, int dtype, hipStream_t st) {
  DISPATCH_T(dtype, {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC || ldx % VEC) return CSS_ERR_ARG;
    hipLaunchKernelGGL(spatial_sum_kernel<T>, dim3(cdiv(C, 8 * VEC), N), dim3(256), 0, st, (const T*)x, ldx, (T*)out, HW, C, scale);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_spatial_bcast(const void* x, void* out, int ldo, int N, int HW, int C, float scale, int dtype, hipStream_t st) {
  DISPATCH_T(dtype, {
    constexpr int VEC = 16 / sizeof(T);
    if (C % VEC == 0 && ldo % VEC == 0 && !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15))
      hipLaunchKernelGGL(spatial_bcast_vec_kernel<T>, dim3(ew_grid((size_t)N * HW * (C / VEC))), dim3(256), 0, st, (const T*)x, (T*)out, ldo, N,
                         HW, C, scale);
    else
      hipLaunchKernelGGL(spatial_bcast_kernel<T>, dim3(ew_grid((size_t)N * HW * C)), dim3(256), 0, st, (const T*)x, (T*)out, ldo, N, HW,
                         C, scale);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_copy_channels(const void* src, int lds, void* dst, int ldd, long M, int C, int dtype_in, int dtype_out,
                             hipStream_t st) {
  if (M <= 0 || C <= 0) return CSS_OK;
  if (dtype_in == dtype_out) {
    const int vec = dtype_in == CSS_BF16 ? 8 : 4;
    const bool al = !(C % vec) && !(lds % vec) && !(ldd % vec) && !(reinterpret_cast<uintptr_t>(src) & 15) &&
                    !(reinterpret_cast<uintptr_t>(dst) & 15);
    if (al) {
      DISPATCH_T(dtype_in, {
        hipLaunchKernelGGL(copy_channels_vec_kernel<T>, dim3(ew_grid((size_t)M * (C / vec))), dim3(256), 0, st, (const T*)src, lds,
                           (T*)dst, ldd, (size_t)M, C);
      });
      CSS_CHECK_LAUNCH();
      return CSS_OK;
    }
  }
#define CPY(TI, TO)                                                                                                   \
  hipLaunchKernelGGL((copy_channels_kernel<TI, TO>), dim3(ew_grid((size_t)M * C)), dim3(256), 0, st, (const TI*)src, lds, \
                     (TO*)dst, ldd, (size_t)M, C)
  if (dtype_in == CSS_F32 && dtype_out == CSS_F32) CPY(float, float);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_BF16) CPY(bf16_t, bf16_t);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_F32) CPY(bf16_t, float);
  else if (dtype_in == CSS_F32 && dtype_out == CSS_BF16) CPY(float, bf16_t);
  else return CSS_ERR_DTYPE;
#undef CPY
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

constexpr long COLSUM_RPB = 2048;
size_t css_colsum_ws_bytes_(long M, int C) { return (size_t)cdiv(M > 0 ? M : 1, COLSUM_RPB) * (size_t)C * sizeof(float); }
int css_launch_colsum(const void* x, int ld, long M, int C, float* out, float* ws, int dtype, hipStream_t st) {
  if (M <= 0) return CSS_OK;
  if (!ws) return CSS_ERR_WORKSPACE;
  const int nrows = cdiv(M, COLSUM_RPB);
  DISPATCH_T(dtype, {
    constexpr int VEC = 16 / sizeof(T);
    const int CV = C / VEC;
    if (C % VEC == 0 && ld % VEC == 0 && CV <= 256 && 256 % CV == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
      hipLaunchKernelGGL(colsum_vec_kernel<T>, dim3(1, nrows), dim3(256), 0, st, (const T*)x, ld, M, C, COLSUM_RPB, ws);
    else if (C <= 256)
      hipLaunchKernelGGL(colsum_kernel<T>, dim3(1, nrows), dim3(256), 0, st, (const T*)x, ld, M, C, COLSUM_RPB, ws);
    else
      hipLaunchKernelGGL(colsum_wide_kernel<T>, dim3(cdiv(C, 64), nrows), dim3(256), 0, st, (const T*)x, ld, M, C, COLSUM_RPB, ws);
  });
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, st, ws, nrows, C, out);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_nchw_to_nhwc(const float* x, void* out, int N, int C, int HW, int Cpad, int dtype, hipStream_t st) {
  DISPATCH_T(dtype, {
    if (Cpad * (int)sizeof(T) == 16 && C <= Cpad && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
      hipLaunchKernelGGL(nchw_to_nhwc_vec_kernel<T>, dim3(ew_grid((size_t)N * HW)), dim3(256), 0, st, x, (T*)out, N, C, HW);
    else
      hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(ew_grid((size_t)N * HW * Cpad)), dim3(256), 0, st, x, (T*)out, N, C, HW, Cpad);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_weight_layout(const float* w, void* out, int Cout, int taps, int Cin, int CinPad, int dgrad, int dtype,
                             hipStream_t st) {
  DISPATCH_T(dtype, {
    if (!dgrad) {
      hipLaunchKernelGGL(weight_fwd_layout_kernel<T>, dim3(ew_grid((size_t)Cout * taps * CinPad)), dim3(256), 0, st, w, (T*)out, Cout,
                         taps, Cin, CinPad);
    } else {
      hipLaunchKernelGGL(weight_dgrad_layout_kernel<T>, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(32, 8), 0, st, w, (T*)out,
                         Cout, taps, Cin);
    }
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_weight_dgrad_layout_batched(const float* flat, void* out, const long* desc, int n_layers, long total_tiles, int dtype,
                                           hipStream_t st) {
  if (n_layers <= 0 || total_tiles <= 0) return CSS_OK;
  if (total_tiles > 0x7fffffffL) return CSS_ERR_ARG;
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(weight_dgrad_layout_batched_kernel<T>, dim3((unsigned)total_tiles), dim3(256), 0, st, flat, (T*)out, desc, n_layers);
  });
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_cast(const void* x, void* out, long n, int dtype_in, int dtype_out, hipStream_t st) {
  if (n <= 0) return CSS_OK;
#define CST(TI, TO) hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(ew_grid((size_t)n)), dim3(256), 0, st, (const TI*)x, (TO*)out, (size_t)n)
  if (dtype_in == CSS_F32 && dtype_out == CSS_BF16) CST(float, bf16_t);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_F32) CST(bf16_t, float);
  else if (dtype_in == CSS_F32 && dtype_out == CSS_F32) CST(float, float);
  else if (dtype_in == CSS_BF16 && dtype_out == CSS_BF16) CST(bf16_t, bf16_t);
  else return CSS_ERR_DTYPE;
#undef CST
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_sgd_ema(float* p, const float* g, float* buf, float* ema, long n, float lr, float momentum, float wd, int first,
                       float decay, float grad_scale, const float* skip_flag, hipStream_t st) {
  if (n <= 0) return CSS_OK;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf) |
       reinterpret_cast<uintptr_t>(ema)) & 15)
    return CSS_ERR_ARG;
  hipLaunchKernelGGL(sgd_ema_kernel, dim3(ew_grid((size_t)n / 4 + 1)), dim3(256), 0, st, p, g, buf, ema, (size_t)n, lr, momentum, wd,
                     first, decay, grad_scale, skip_flag);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ema(float* ema, const float* p, long n, float decay, hipStream_t st) {
  if (n <= 0) return CSS_OK;
  hipLaunchKernelGGL(ema_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, st, ema, p, (size_t)n, decay);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
