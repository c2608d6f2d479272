// Pixel-wise cross-entropy family on [B][H][W][K] fp32 logits (channels innermost), HBM-bound.
// Replaces (reference file:line)
//   nn.CrossEntropyLoss(ignore_index=-1)                   mix_label.py:81,169           (mode 0)
//   Attention_Threshold_Loss.forward                       generalframeworks/loss/loss.py:53-64  (mode 1)
//   ProbOhemCrossEntropy2d.forward                         generalframeworks/loss/loss.py:19-46  (mode 0 + keep threshold
//                                                          from a radix select instead of the argsort at :35-36)
// Tiles of 256 pixels are staged through LDS so that global traffic is fully coalesced and each thread then
// walks its pixel's K logits at an odd LDS stride.
#include "common.h"

constexpr int CE_MAXK = 32;
// per-image accumulators: S = sum of losses, NPOS = #(loss > 0), NVALID = #(counted pixels), NCONF = #(conf >= thr).
// All four are 64-bit INTEGERS (the counts as they are, S in units of 2^-32): integer adds commute, so the workgroups' atomics give the same
// bits whatever order they arrive in - the loss statistics (and through `coef` every gradient of the step) are run-to-run reproducible.
// A partial sum is converted once per wave (tile kernel) or per pixel (ce_kernel): |error| <= 2^-32 per conversion, ~1e-10 of a loss of 1.
enum { ST_S = 0, ST_NPOS = 1, ST_NVALID = 2, ST_NCONF = 3 };
constexpr float CE_FIX = 4294967296.f;               // 2^32
constexpr double CE_UNFIX = 1.0 / 4294967296.0;
typedef unsigned long long ce_acc_t;                  // (two's complement: losses are >= 0 up to rounding, negative partials wrap correctly)
// A NON-FINITE pixel loss (diverged logits) must surface as NaN like the reference's fp32 mean does (float -> integer conversion of
// NaN is 0 on AMDGPU, so the fixed-point sum stayed finite while the network was NaN): a non-finite partial marks the image's NVALID word -
// counts live in its low 40 bits (an image has < 2^40 pixels); a workgroup counts its poisoned partials above them in LDS (<= 256 per tile)
// and ce_flush ORs ONE bit (bit 40) into the global word, so the mark saturates instead of wrapping - integer add + bit OR, still
// order-independent.  ce_finalize turns a non-zero high part into a NaN loss (every mode ends there: OHEM's final loss too); ohem_init only
// masks the high part off the count (in mode 0 coef[b] stays the finite 1 / NV while the loss is NaN).
constexpr ce_acc_t CE_POISON = 1ull << 40;
constexpr ce_acc_t CE_COUNT_MASK = CE_POISON - 1;
__device__ __forceinline__ void ce_flush(ce_acc_t* dst, int which, ce_acc_t v) {
  if (which != ST_NVALID) { if (v) atomicAdd(dst, v); return; }
  if (v & CE_COUNT_MASK) atomicAdd(dst, v & CE_COUNT_MASK);
  if (v >> 40) atomicOr(dst, CE_POISON);
}
__device__ __forceinline__ ce_acc_t ce_fix(float v) { return isfinite(v) ? (ce_acc_t)(long long)(v * CE_FIX) : (ce_acc_t)0; }

template <bool BWD>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ label,
                                                 const float* __restrict__ conf, float conf_thr, const float* __restrict__ keep_thr,
                                                 int K, size_t P, int HW, ce_acc_t* __restrict__ stats, float* __restrict__ gtprob_out,
                                                 const float* __restrict__ coef, const float* __restrict__ gscale, int pos_only,
                                                 float* __restrict__ dlogits) {
  __shared__ float tile[256 * CE_MAXK];
  __shared__ ce_acc_t sacc[2][4];
  const int tid = threadIdx.x;
  const int KS = K | 1;  // odd LDS stride -> conflict-free per-pixel walks
  for (size_t p0 = (size_t)blockIdx.x * 256; p0 < P; p0 += (size_t)gridDim.x * 256) {
    const int np = (int)min((size_t)256, P - p0);
    __syncthreads();
    if (tid < 8) sacc[tid >> 2][tid & 3] = 0;
    const float* src = logits + p0 * K;
    for (int i = tid; i < np * K; i += 256) {
      const int pp = i / K, k = i - pp * K;
      tile[pp * KS + k] = src[i];
    }
    __syncthreads();
    const int b_first = (int)(p0 / HW);
    if (tid < np) {
      const size_t p = p0 + tid;
      const int b = (int)(p / HW);
      const int64_t lab = label[p];
      float* row = tile + tid * KS;
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, row[k]);
      float se = 0.f;
      for (int k = 0; k < K; ++k) se += __expf(row[k] - mx);
      bool valid = lab >= 0 && lab < K, nonfin = false;
      float loss = 0.f, gtp = 1.f;
      if (valid) {
        const float xg = row[lab];
        loss = logf(se) + mx - xg;
        gtp = __expf(xg - mx) / se;
        nonfin = !isfinite(loss);
        if (keep_thr && !(gtp <= *keep_thr)) { valid = false; loss = 0.f; }
      }
      if (!BWD) {
        if (gtprob_out) gtprob_out[p] = (lab >= 0 && lab < K) ? gtp : 1.f;
        if (stats) {
          const int s = b - b_first;
          if (nonfin) atomicAdd(&sacc[s][ST_NVALID], CE_POISON);
          if (valid) {
            atomicAdd(&sacc[s][ST_S], ce_fix(loss));
            atomicAdd(&sacc[s][ST_NVALID], (ce_acc_t)1);
            if (loss > 0.f) atomicAdd(&sacc[s][ST_NPOS], (ce_acc_t)1);
          }
          if (conf && conf[p] >= conf_thr) atomicAdd(&sacc[s][ST_NCONF], (ce_acc_t)1);
        }
      } else {
        float c = 0.f;
        if (valid && (!pos_only || loss > 0.f)) c = coef[b] * (*gscale);
        const float inv = 1.f / se;
        for (int k = 0; k < K; ++k) {
          float g = 0.f;
          if (c != 0.f) g = c * (__expf(row[k] - mx) * inv - (k == lab ? 1.f : 0.f));
          row[k] = g;
        }
      }
    }
    __syncthreads();
    if (!BWD) {
      if (stats && tid < 8) {
        const int s = tid >> 2, b = b_first + s;
        ce_flush(&stats[(size_t)b * 4 + (tid & 3)], tid & 3, sacc[s][tid & 3]);
      }
    } else {
      float* dst = dlogits + p0 * K;
      for (int i = tid; i < np * K; i += 256) {
        const int pp = i / K, k = i - pp * K;
        dst[i] = tile[pp * KS + k];
      }
    }
  }
}

// mode 0: mean CE over counted pixels.  mode 1: Attention_Threshold_Loss weighting (loss.py:56,60).
__global__ void ce_finalize_kernel(const ce_acc_t* __restrict__ acc, int B, int mode, float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // (the fixed-point / integer accumulators as doubles: exact for every count and for sums below 2^21)
  auto stat = [&](int i) {
    if ((i & 3) == ST_NVALID) return (double)(acc[i] & CE_COUNT_MASK);
    if ((i & 3) != ST_S) return (double)acc[i];
    const double v = (double)(long long)acc[i] * CE_UNFIX;
    return (acc[(i & ~3) + ST_NVALID] >> 40) ? (double)NAN : v;          // a non-finite pixel loss in this image: NaN, like the reference
  };
  double S = 0, NV = 0, NP = 0, WS = 0;
  for (int b = 0; b < B; ++b) {
    S += stat(b * 4 + ST_S);
    NV += stat(b * 4 + ST_NVALID);
    NP += stat(b * 4 + ST_NPOS);
  }
  if (mode == 0) {
    *loss = (float)(S / NV);
    for (int b = 0; b < B; ++b) coef[b] = (float)(1.0 / NV);
  } else {
    for (int b = 0; b < B; ++b) {
      // weighting = #(logits >= thr) / #(label >= 0): fp32 division like the reference (int64 / float32)
      const float wgt = (float)stat(b * 4 + ST_NCONF) / (float)stat(b * 4 + ST_NVALID);
      if (stat(b * 4 + ST_NPOS) > 0) WS += (double)wgt * stat(b * 4 + ST_S);
      coef[b] = stat(b * 4 + ST_NPOS) > 0 ? (float)((double)wgt / NP) : 0.f;
    }
    *loss = (float)(WS / NP);  // NP == 0 -> NaN like torch.mean of an empty selection
  }
}

// ---- OHEM threshold: k-th smallest ground-truth probability by 4-pass radix select -------------
struct OhemState {
  unsigned prefix;       // bits decided so far
  unsigned k;            // rank still to find inside the current bucket
  unsigned hist[256];
  float thr;             // result: keep pixels with gtprob <= thr
  unsigned active;       // 0: min_kept > num_valid -> plain CE (loss.py:28-29)
};
__global__ void ohem_init_kernel(OhemState* s, const ce_acc_t* __restrict__ stats, int B, long P, int min_kept) {
  if (threadIdx.x >= 256) return;
  s->hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    double nv = 0;
    for (int b = 0; b < B; ++b) nv += (double)(stats[b * 4 + ST_NVALID] & CE_COUNT_MASK);
    s->prefix = 0;
    long k = (long)min((long)P, (long)min_kept) - 1;
    s->k = (unsigned)(k < 0 ? 0 : k);
    s->active = ((double)min_kept > nv || nv <= 0 || min_kept <= 0) ? 0u : 1u;
    s->thr = INFINITY;
  }
}
__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* __restrict__ v, size_t P, OhemState* s, int shift) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = s->prefix;
  const unsigned hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned u = __float_as_uint(v[i]);
    if ((u & hi_mask) == (prefix & hi_mask)) atomicAdd(&h[(u >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&s->hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void ohem_pick_kernel(OhemState* s, int shift, float thresh) {
  if (threadIdx.x != 0) return;
  unsigned k = s->k, d = 0;
  for (; d < 256; ++d) {
    const unsigned c = s->hist[d];
    if (k < c) break;
    k -= c;
  }
  if (d > 255) d = 255;
  s->k = k;
  s->prefix |= d << shift;
  for (int i = 0; i < 256; ++i) s->hist[i] = 0;
  if (shift == 0) {
    const float kth = __uint_as_float(s->prefix);
    // loss.py:33-39: threshold = max(thresh, kth) ; kept = prob <= threshold
    s->thr = s->active ? (kth > thresh ? kth : thresh) : INFINITY;
  }
}

// ---- the same losses straight from the LOW-RESOLUTION logits --------------------------------------------------------
// The reference up-samples the student's [B,K,h,w] logits to the label size before the loss (ddp_model.py:141,144:
// F.interpolate(bilinear, align_corners=True)); at c2 that is 2 x 354 MB of fp32 written, read by the loss, and the same
// again for the gradient.  These kernels interpolate on the fly: the forward stages a tile's interpolated logits in LDS
// exactly like ce_kernel; the backward applies the adjoint of the interpolation inside the workgroup (LDS atomics on the
// tile's footprint in the small map) and flushes the footprint with global fp32 atomics (footprints of neighbouring tiles
// overlap by one small pixel).  Needs an up-sampling factor >= 2 (footprint of a 32x32 tile <= 18x18 small pixels).
constexpr int CES_TW = 32, CES_TH = 32, CES_FP = 18;

template <typename T>
__device__ __forceinline__ float ces_interp(const T* __restrict__ small, int ld, int h, int w, int b, int y, int x, float sh, float sw, int k) {
  float fy = sh * (float)y, fx = sw * (float)x;
  int y0 = (int)fy, x0 = (int)fx;
  if (y0 > h - 1) y0 = h - 1;
  if (x0 > w - 1) x0 = w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
  const size_t r0 = (size_t)(b * h + y0) * w, r1 = (size_t)(b * h + y1) * w;
  return wy0 * (wx0 * ElemT<T>::to_f(small[(r0 + x0) * ld + k]) + wx1 * ElemT<T>::to_f(small[(r0 + x1) * ld + k])) +
         wy1 * (wx0 * ElemT<T>::to_f(small[(r1 + x0) * ld + k]) + wx1 * ElemT<T>::to_f(small[(r1 + x1) * ld + k]));
}

template <typename T>
__global__ __launch_bounds__(256) void ce_small_fwd_kernel(const T* __restrict__ small, int ld, int h, int w, float sh, float sw,
                                                           const int64_t* __restrict__ label, const float* __restrict__ conf, float conf_thr,
                                                           const float* __restrict__ keep_thr, int K, size_t P, int H, int W,
                                                           ce_acc_t* __restrict__ stats, float* __restrict__ gtprob_out) {
  __shared__ ce_acc_t sacc[2][4];
  const int tid = threadIdx.x;
  const int HW = H * W;
  for (size_t p0 = (size_t)blockIdx.x * 256; p0 < P; p0 += (size_t)gridDim.x * 256) {
    __syncthreads();
    if (tid < 8) sacc[tid >> 2][tid & 3] = 0;
    __syncthreads();
    const int b_first = (int)(p0 / HW);
    const size_t p = p0 + tid;
    if (p < P) {
      const int b = (int)(p / HW);
      const int rem = (int)(p - (size_t)b * HW), y = rem / W, x = rem - y * W;
      const int64_t lab = label[p];
      // two passes over the classes (interpolating twice is cheaper than K registers with a run-time K)
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, ces_interp(small, ld, h, w, b, y, x, sh, sw, k));
      float se = 0.f, xg = 0.f;
      for (int k = 0; k < K; ++k) {
        const float v = ces_interp(small, ld, h, w, b, y, x, sh, sw, k);
        se += __expf(v - mx);
        if (k == lab) xg = v;
      }
      bool valid = lab >= 0 && lab < K, nonfin = false;
      float loss = 0.f, gtp = 1.f;
      if (valid) {
        loss = logf(se) + mx - xg;
        gtp = __expf(xg - mx) / se;
        nonfin = !isfinite(loss);
        if (keep_thr && !(gtp <= *keep_thr)) { valid = false; loss = 0.f; }
      }
      if (gtprob_out) gtprob_out[p] = (lab >= 0 && lab < K) ? gtp : 1.f;
      if (stats) {
        const int s = b - b_first;
        if (nonfin) atomicAdd(&sacc[s][ST_NVALID], CE_POISON);
        if (valid) {
          atomicAdd(&sacc[s][ST_S], ce_fix(loss));
          atomicAdd(&sacc[s][ST_NVALID], (ce_acc_t)1);
          if (loss > 0.f) atomicAdd(&sacc[s][ST_NPOS], (ce_acc_t)1);
        }
        if (conf && conf[p] >= conf_thr) atomicAdd(&sacc[s][ST_NCONF], (ce_acc_t)1);
      }
    }
    __syncthreads();
    if (stats && tid < 8) {
      const int s = tid >> 2, b = b_first + s;
      ce_flush(&stats[(size_t)b * 4 + (tid & 3)], tid & 3, sacc[s][tid & 3]);
    }
  }
}

// The same forward on 32x32 label tiles (grid: tiles_x, tiles_y, B), for K <= CES_KREG and an up-sampling factor >= 2: the tile's
// footprint in the small map goes to LDS once (fp32, like the backward below), a pixel's K interpolated logits stay in registers for
// the maximum, the sum and the label's own logit (the kernel above gathers every class twice from global memory with scalar
// loads - instruction-bound, 375 us at c2 against ~0.1 ms here).  Same arithmetic per pixel, same statistics.
constexpr int CES_KREG = 24;
struct CesCorner { int c00, c01, c10, c11; float wy0, wy1, wx0, wx1; };
__device__ __forceinline__ CesCorner ces_corner(int y, int x, float sh, float sw, int h, int w, int fy0, int fx0, int fw, int K) {
  float fy = sh * (float)y, fx = sw * (float)x;
  int y0 = (int)fy, x0 = (int)fx;
  if (y0 > h - 1) y0 = h - 1;
  if (x0 > w - 1) x0 = w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  CesCorner r;
  r.wy1 = fy - (float)y0; r.wy0 = 1.f - r.wy1; r.wx1 = fx - (float)x0; r.wx0 = 1.f - r.wx1;
  r.c00 = ((y0 - fy0) * fw + (x0 - fx0)) * K; r.c01 = ((y0 - fy0) * fw + (x1 - fx0)) * K;
  r.c10 = ((y1 - fy0) * fw + (x0 - fx0)) * K; r.c11 = ((y1 - fy0) * fw + (x1 - fx0)) * K;
  return r;
}
// LDS footprint of the tiled kernels: CES_KREG floats per small pixel (classes >= K hold CES_NEG: they never win the maximum and their
// exponential is 0), so the class loops have compile-time bounds and read / write 16-byte vectors.
constexpr float CES_NEG = -1e30f;
struct CesVec { float4 q[CES_KREG / 4]; };
__device__ __forceinline__ void ces_interp_all(const float* __restrict__ fp, const CesCorner& q, float (&v)[CES_KREG]) {
  const float4* p00 = reinterpret_cast<const float4*>(fp + q.c00);
  const float4* p01 = reinterpret_cast<const float4*>(fp + q.c01);
  const float4* p10 = reinterpret_cast<const float4*>(fp + q.c10);
  const float4* p11 = reinterpret_cast<const float4*>(fp + q.c11);
#pragma unroll
  for (int j = 0; j < CES_KREG / 4; ++j) {
    const float4 a = p00[j], b = p01[j], c = p10[j], d = p11[j];
    v[4 * j + 0] = q.wy0 * (q.wx0 * a.x + q.wx1 * b.x) + q.wy1 * (q.wx0 * c.x + q.wx1 * d.x);
    v[4 * j + 1] = q.wy0 * (q.wx0 * a.y + q.wx1 * b.y) + q.wy1 * (q.wx0 * c.y + q.wx1 * d.y);
    v[4 * j + 2] = q.wy0 * (q.wx0 * a.z + q.wx1 * b.z) + q.wy1 * (q.wx0 * c.z + q.wx1 * d.z);
    v[4 * j + 3] = q.wy0 * (q.wx0 * a.w + q.wx1 * b.w) + q.wy1 * (q.wx0 * c.w + q.wx1 * d.w);
  }
}
// stage the footprint (fh x fw small pixels from (fy0, fx0) of image b) as [cell][CES_KREG] fp32
template <typename T>
__device__ __forceinline__ void ces_stage(float* __restrict__ fp, const T* __restrict__ small, int ld, int h, int w, int b, int fy0, int fx0, int fh,
                                          int fw, int K, int tid) {
  for (int i = tid; i < fh * fw * CES_KREG; i += 256) {
    const int k = i % CES_KREG, c = i / CES_KREG, yy = c / fw, xx = c - yy * fw;
    fp[i] = k < K ? ElemT<T>::to_f(small[((size_t)(b * h + fy0 + yy) * w + fx0 + xx) * ld + k]) : CES_NEG;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_small_fwd_tile_kernel(const T* __restrict__ small, int ld, int h, int w, float sh, float sw,
                                                                const int64_t* __restrict__ label, const float* __restrict__ conf, float conf_thr,
                                                                const float* __restrict__ keep_thr, int K, int H, int W,
                                                                ce_acc_t* __restrict__ stats, float* __restrict__ gtprob_out) {
  extern __shared__ __attribute__((aligned(16))) float ces_lds[];
  __shared__ ce_acc_t sacc[4];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int X0 = blockIdx.x * CES_TW, Y0 = blockIdx.y * CES_TH;
  const int X1 = min(X0 + CES_TW, W) - 1, Y1 = min(Y0 + CES_TH, H) - 1;
  int fy0 = (int)(sh * (float)Y0), fx0 = (int)(sw * (float)X0);
  fy0 = min(fy0, h - 1); fx0 = min(fx0, w - 1);
  const int fy1 = min((int)(sh * (float)Y1) + 1, h - 1), fx1 = min((int)(sw * (float)X1) + 1, w - 1);
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
  ces_stage(ces_lds, small, ld, h, w, b, fy0, fx0, fh, fw, K, tid);
  if (tid < 4) sacc[tid] = 0;
  __syncthreads();
  const float kthr = keep_thr ? *keep_thr : 0.f;
  float a_s = 0.f, a_nv = 0.f, a_np = 0.f, a_nc = 0.f, a_bad = 0.f;
  const int x = X0 + (tid & 31);
#pragma unroll
  for (int trip = 0; trip < CES_TH / 8; ++trip) {
    const int y = Y0 + (tid >> 5) + 8 * trip;
    if (x >= W || y >= H) continue;
    const size_t p = ((size_t)b * H + y) * W + x;
    const int64_t lab = label[p];
    const CesCorner q = ces_corner(y, x, sh, sw, h, w, fy0, fx0, fw, CES_KREG);
    float v[CES_KREG];
    ces_interp_all(ces_lds, q, v);
    float mx = v[0];
#pragma unroll
    for (int k = 1; k < CES_KREG; ++k) mx = fmaxf(mx, v[k]);
    float se = 0.f, xg = 0.f;
#pragma unroll
    for (int k = 0; k < CES_KREG; ++k) {
      se += __expf(v[k] - mx);
      if (k == lab) xg = v[k];
    }
    bool valid = lab >= 0 && lab < K;
    float loss = 0.f, gtp = 1.f;
    if (valid) {
      loss = logf(se) + mx - xg;
      gtp = __expf(xg - mx) / se;
      if (!isfinite(loss)) a_bad += 1.f;
      if (keep_thr && !(gtp <= kthr)) { valid = false; loss = 0.f; }
    }
    if (gtprob_out) gtprob_out[p] = (lab >= 0 && lab < K) ? gtp : 1.f;
    if (valid) {
      a_s += loss;
      a_nv += 1.f;
      if (loss > 0.f) a_np += 1.f;
    }
    if (conf && conf[p] >= conf_thr) a_nc += 1.f;
  }
  if (stats) {
    a_s = wave_sum(a_s); a_nv = wave_sum(a_nv); a_np = wave_sum(a_np); a_nc = wave_sum(a_nc); a_bad = wave_sum(a_bad);
    // (a lane's four pixels add in program order, wave_sum is a fixed butterfly: the wave's sums are reproducible; from here on integers)
    if ((tid & 63) == 0) {
      atomicAdd(&sacc[ST_S], ce_fix(a_s)); atomicAdd(&sacc[ST_NVALID], (ce_acc_t)a_nv + (a_bad > 0.f ? CE_POISON : (ce_acc_t)0));
      atomicAdd(&sacc[ST_NPOS], (ce_acc_t)a_np);
      atomicAdd(&sacc[ST_NCONF], (ce_acc_t)a_nc);
    }
    __syncthreads();
    if (tid < 4) ce_flush(&stats[(size_t)b * 4 + tid], tid, sacc[tid]);
  }
}

// The footprints of neighbouring tiles share one row / column of small pixels, so the tiles' flushes into dsmall would have to be atomic -
// and fp32 atomics add in arrival order: the gradient of the step would differ in its last bits from run to run (r03: two fp32 runs of the
// same seeds drifted 4.8 steps apart in 30).  Instead the backward is launched once per COLOUR: a launch holds the tiles
// (cx + i nx, cy + j ny) only, nx / ny chosen so that two tiles of a launch are far enough apart for their footprints to be disjoint
// (ces_colours below: 2 x 2 launches for every factor up to 16), and every launch adds with plain read-modify-writes.  Launches run in
// stream order, so each cell of dsmall receives its (up to four) contributions in one fixed order: bit-reproducible, no workspace.
struct CesColour { int cx, cy, nx, ny; };
// smallest n such that tiles n apart cannot touch the same small pixel: floor(s (32 (t + n))) > floor(s (32 t + 31)) + 1 for every t
// follows from s (32 n - 31) >= 2 (with a margin for the rounding of the fp32 products)
static inline int ces_colours(float s, int tiles) {
  int n = 1;
  while (n < tiles && s * (float)(CES_TW * n - (CES_TW - 1)) < 2.5f) ++n;
  return n;
}

// Backward on the same tiles for K <= CES_KREG and an up-sampling factor in [2, 4]: per pixel ONE pass over the footprint for the K
// interpolated logits (registers), then the adjoint of the interpolation as four read-modify-write phases (one per corner, 16-byte
// vectors, no branch per class) into the wave's own gradient copy.  The phases stay whole instructions apart - a cell that is this lane's
// right-hand neighbour is another lane's own cell - and within a phase the 64 lanes touch 64 different cells: the 8x8 lattice below
// sends them to different small pixels, and a corner of weight zero (integral source coordinate: the only case in which a clamped
// neighbour coincides with another lane's cell of the same phase) is redirected to a cell of the lane's own behind the copies.
// (The general kernel below branches per class and corner: 21 x 4 serialized LDS round trips per pixel, 650 us at c2.)
template <typename T>
__global__ __launch_bounds__(256) void ce_small_bwd_tile_kernel(const T* __restrict__ small, int ld, int h, int w, float sh, float sw,
                                                                const int64_t* __restrict__ label, const float* __restrict__ keep_thr, int K, int H,
                                                                int W, const float* __restrict__ coef, const float* __restrict__ gscale, int pos_only,
                                                                float* __restrict__ dsmall, CesColour col) {
  extern __shared__ __attribute__((aligned(16))) float ces_lds[];
  constexpr int FPK = CES_FP * CES_FP * CES_KREG;
  const int tid = threadIdx.x, b = blockIdx.z, wave = tid >> 6;
  float* sout = ces_lds + FPK + wave * FPK;
  const int X0 = (blockIdx.x * col.nx + col.cx) * CES_TW, Y0 = (blockIdx.y * col.ny + col.cy) * CES_TH;
  const int X1 = min(X0 + CES_TW, W) - 1, Y1 = min(Y0 + CES_TH, H) - 1;
  int fy0 = (int)(sh * (float)Y0), fx0 = (int)(sw * (float)X0);
  fy0 = min(fy0, h - 1); fx0 = min(fx0, w - 1);
  const int fy1 = min((int)(sh * (float)Y1) + 1, h - 1), fx1 = min((int)(sw * (float)X1) + 1, w - 1);
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
  const float cb = coef[b] * (*gscale);
  if (cb == 0.f) return;                                  // (block-uniform) this image contributes nothing: e.g. no confident pseudo label in it
  {
    // nothing to do either when the tile holds no labeled pixel (ignore regions, unconfident pseudo labels)
    bool any = false;
    for (int i = tid; i < CES_TW * CES_TH; i += 256) {
      const int x = X0 + (i & (CES_TW - 1)), y = Y0 + i / CES_TW;
      if (x < W && y < H) {
        const int64_t lab = label[((size_t)b * H + y) * W + x];
        any |= lab >= 0 && lab < K;
      }
    }
    if (!__syncthreads_or(any)) return;
  }
  ces_stage(ces_lds, small, ld, h, w, b, fy0, fx0, fh, fw, K, tid);
  for (int i = tid; i < 4 * FPK / 4; i += 256) reinterpret_cast<float4*>(ces_lds + FPK)[i] = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const float kthr = keep_thr ? *keep_thr : 0.f;
  const int dmy = 4 * FPK - wave * FPK + (tid & 63) * CES_KREG;       // this lane's private cell (shared by the four waves: its content is never used), as an index relative to sout
  for (int trip = 0; trip < 4; ++trip) {
    const int phase = wave * 4 + trip, lane = tid & 63;
    const int x = X0 + (lane & 7) * 4 + (phase & 3), y = Y0 + (lane >> 3) * 4 + (phase >> 2);
    if (x >= W || y >= H) continue;
    const int64_t lab = label[((size_t)b * H + y) * W + x];
    if (!(lab >= 0 && lab < K) || cb == 0.f) continue;
    const CesCorner q = ces_corner(y, x, sh, sw, h, w, fy0, fx0, fw, CES_KREG);
    float v[CES_KREG];
    ces_interp_all(ces_lds, q, v);
    float mx = v[0], xg = 0.f;
#pragma unroll
    for (int k = 1; k < CES_KREG; ++k) mx = fmaxf(mx, v[k]);
#pragma unroll
    for (int k = 0; k < CES_KREG; ++k)
      if (k == lab) xg = v[k];
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < CES_KREG; ++k) {
      v[k] = __expf(v[k] - mx);
      se += v[k];
    }
    const float loss = logf(se) + mx - xg;                         // (the forward's expressions, term by term)
    if (keep_thr && !(__expf(xg - mx) / se <= kthr)) continue;
    if (pos_only && !(loss > 0.f)) continue;
    const float inv = 1.f / se;
#pragma unroll
    for (int k = 0; k < CES_KREG; ++k) v[k] = cb * (v[k] * inv - (k == lab ? 1.f : 0.f));      // (classes >= K: exp = 0, gradient 0)
    const float w01 = q.wy0 * q.wx1, w10 = q.wy1 * q.wx0, w11 = q.wy1 * q.wx1;
    const int base[4] = {q.c00, w01 != 0.f ? q.c01 : dmy, w10 != 0.f ? q.c10 : dmy, w11 != 0.f ? q.c11 : dmy};
    const float wgt[4] = {q.wy0 * q.wx0, w01, w10, w11};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4* cell = reinterpret_cast<float4*>(sout + base[c]);
      float4 r[CES_KREG / 4];
#pragma unroll
      for (int j = 0; j < CES_KREG / 4; ++j) r[j] = cell[j];
#pragma unroll
      for (int j = 0; j < CES_KREG / 4; ++j) {
        r[j].x += wgt[c] * v[4 * j]; r[j].y += wgt[c] * v[4 * j + 1]; r[j].z += wgt[c] * v[4 * j + 2]; r[j].w += wgt[c] * v[4 * j + 3];
        cell[j] = r[j];
      }
      __builtin_amdgcn_sched_barrier(0);        // (phases in program order)
    }
  }
  __syncthreads();
  for (int i = tid; i < fh * fw * K; i += 256) {
    const int k = i % K, c = i / K, yy = c / fw, xx = c - yy * fw;
    const float* so = ces_lds + FPK + c * CES_KREG + k;
    const float v = (so[0] + so[FPK]) + (so[2 * FPK] + so[3 * FPK]);
    if (v == 0.f) continue;
    dsmall[((size_t)(b * h + fy0 + yy) * w + fx0 + xx) * K + k] += v;      // (no atomic: the tiles of one launch have disjoint footprints, see CesColour)
  }
}

// grid: (tiles_x, tiles_y, B); dsmall fp32 [B][h][w][K], zeroed by the caller
template <typename T>
__global__ __launch_bounds__(256) void ce_small_bwd_kernel(const T* __restrict__ small, int ld, int h, int w, float sh, float sw,
                                                           const int64_t* __restrict__ label, const float* __restrict__ keep_thr, int K, int H, int W,
                                                           const float* __restrict__ coef, const float* __restrict__ gscale, int pos_only,
                                                           float* __restrict__ dsmall, int use_rmw, CesColour col) {
  // LDS (launcher): the footprint's logits, then FOUR gradient copies (one per wave).  LDS float atomics are slow on this
  // part, so a wave scatters with plain read-modify-write into its own copy; that is race-free because the lattice mapping
  // below sends the 64 lanes of an instruction to 64 different small pixels whenever the up-sampling factor is <= 4
  // (`rmw`); larger factors fall back to atomics on copy 0.
  extern __shared__ float ces_lds[];
  float* sin_ = ces_lds;
  const int FPK = CES_FP * CES_FP * K;
  const bool rmw = use_rmw != 0;                        // launcher: factor <= 4 and five copies fit in LDS
  const int ncopy = rmw ? 4 : 1;
  float* sout = ces_lds + FPK + (rmw ? (threadIdx.x >> 6) * FPK : 0);
  const int tid = threadIdx.x, b = blockIdx.z;
  const int X0 = (blockIdx.x * col.nx + col.cx) * CES_TW, Y0 = (blockIdx.y * col.ny + col.cy) * CES_TH;
  const int X1 = min(X0 + CES_TW, W) - 1, Y1 = min(Y0 + CES_TH, H) - 1;
  // footprint in the small map: floor(s*first) .. floor(s*last) + 1, clamped
  int fy0 = (int)(sh * (float)Y0), fx0 = (int)(sw * (float)X0);
  fy0 = min(fy0, h - 1); fx0 = min(fx0, w - 1);
  int fy1 = min((int)(sh * (float)Y1) + 1, h - 1), fx1 = min((int)(sw * (float)X1) + 1, w - 1);
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;      // <= CES_FP by the launcher's factor check
  for (int i = tid; i < fh * fw * K; i += 256) {
    const int k = i % K, c = i / K, yy = c / fw, xx = c - yy * fw;
    sin_[c * K + k] = ElemT<T>::to_f(small[((size_t)(b * h + fy0 + yy) * w + fx0 + xx) * ld + k]);
  }
  for (int i = tid; i < ncopy * FPK; i += 256) ces_lds[FPK + i] = 0.f;
  __syncthreads();
  const float cb = coef[b] * (*gscale);
  // lane -> pixel mapping: the 64 lanes of a wave take pixels 4 apart in x and y (an 8x8 lattice), so that at up-sampling
  // factors around 4 they scatter into 64 DIFFERENT small pixels per LDS-atomic instruction (consecutive pixels would hit
  // the same address 4-way and serialise); the 16 lattice phases are split over the 4 waves x 4 trips
  for (int trip = 0; trip < 4; ++trip) {
    const int phase = (tid >> 6) * 4 + trip, lane = tid & 63;
    const int x = X0 + (lane & 7) * 4 + (phase & 3), y = Y0 + (lane >> 3) * 4 + (phase >> 2);
    if (x >= W || y >= H) continue;
    const int64_t lab = label[((size_t)b * H + y) * W + x];
    if (!(lab >= 0 && lab < K) || cb == 0.f) continue;
    float fy = sh * (float)y, fx = sw * (float)x;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const int c00 = ((y0 - fy0) * fw + (x0 - fx0)) * K, c01 = ((y0 - fy0) * fw + (x1 - fx0)) * K;
    const int c10 = ((y1 - fy0) * fw + (x0 - fx0)) * K, c11 = ((y1 - fy0) * fw + (x1 - fx0)) * K;
    const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k)
      mx = fmaxf(mx, wy0 * (wx0 * sin_[c00 + k] + wx1 * sin_[c01 + k]) + wy1 * (wx0 * sin_[c10 + k] + wx1 * sin_[c11 + k]));
    float se = 0.f, xg = 0.f;
    for (int k = 0; k < K; ++k) {
      const float v = wy0 * (wx0 * sin_[c00 + k] + wx1 * sin_[c01 + k]) + wy1 * (wx0 * sin_[c10 + k] + wx1 * sin_[c11 + k]);
      se += __expf(v - mx);
      if (k == lab) xg = v;
    }
    const float loss = logf(se) + mx - xg;
    if (keep_thr && !(__expf(xg - mx) / se <= *keep_thr)) continue;
    if (pos_only && !(loss > 0.f)) continue;
    const float inv = 1.f / se;
    for (int k = 0; k < K; ++k) {
      const float v = wy0 * (wx0 * sin_[c00 + k] + wx1 * sin_[c01 + k]) + wy1 * (wx0 * sin_[c10 + k] + wx1 * sin_[c11 + k]);
      const float g = cb * (__expf(v - mx) * inv - (k == lab ? 1.f : 0.f));
      if (rmw) {
        sout[c00 + k] += w00 * g;
        if (w01 != 0.f) sout[c01 + k] += w01 * g;
        if (w10 != 0.f) sout[c10 + k] += w10 * g;
        if (w11 != 0.f) sout[c11 + k] += w11 * g;
      } else {
        atomicAdd(&sout[c00 + k], w00 * g);
        if (w01 != 0.f) atomicAdd(&sout[c01 + k], w01 * g);
        if (w10 != 0.f) atomicAdd(&sout[c10 + k], w10 * g);
        if (w11 != 0.f) atomicAdd(&sout[c11 + k], w11 * g);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < fh * fw * K; i += 256) {
    const float* so = ces_lds + FPK;
    const float v = rmw ? (so[i] + so[FPK + i]) + (so[2 * FPK + i] + so[3 * FPK + i]) : so[i];
    if (v == 0.f) continue;
    const int k = i % K, c = i / K, yy = c / fw, xx = c - yy * fw;
    dsmall[((size_t)(b * h + fy0 + yy) * w + fx0 + xx) * K + k] += v;      // (no atomic: see CesColour)
  }
}

// ---- launchers -----------------------------------------------------------
static inline int ce_grid(size_t P) {
  size_t g = (P + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
int css_launch_ce_fwd(const float* logits, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr, int K, long P,
                      int HW, int64_t* stats, float* gtprob_out, hipStream_t st) {
  if (K > CE_MAXK || K < 1) return CSS_ERR_ARG;
  hipLaunchKernelGGL(ce_kernel<false>, dim3(ce_grid((size_t)P)), dim3(256), 0, st, logits, label, conf, conf_thr, keep_thr, K, (size_t)P, HW,
                     (ce_acc_t*)stats, gtprob_out, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ce_finalize(const int64_t* stats, int B, int mode, float* loss, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, (const ce_acc_t*)stats, B, mode, loss, coef);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ce_bwd(const float* logits, const int64_t* label, const float* keep_thr, int K, long P, int HW, const float* coef,
                      const float* gscale, int pos_only, float* dlogits, hipStream_t st) {
  if (K > CE_MAXK || K < 1) return CSS_ERR_ARG;
  hipLaunchKernelGGL(ce_kernel<true>, dim3(ce_grid((size_t)P)), dim3(256), 0, st, logits, label, (const float*)nullptr, 0.f, keep_thr, K,
                     (size_t)P, HW, (ce_acc_t*)nullptr, (float*)nullptr, coef, gscale, pos_only, dlogits);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
// state: device buffer of css_ohem_state_bytes(); on return state->thr (offset css_ohem_thr_offset()) holds the keep threshold
size_t css_ohem_state_bytes_() { return sizeof(OhemState); }
size_t css_ohem_thr_offset_() { return offsetof(OhemState, thr); }
int css_launch_ohem_threshold(const float* gtprob, long P, const int64_t* stats, int B, int min_kept, float thresh, void* state,
                              hipStream_t st) {
  OhemState* s = reinterpret_cast<OhemState*>(state);
  hipLaunchKernelGGL(ohem_init_kernel, dim3(1), dim3(256), 0, st, s, (const ce_acc_t*)stats, B, P, min_kept);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(ce_grid((size_t)P)), dim3(256), 0, st, gtprob, (size_t)P, s, shift);
    hipLaunchKernelGGL(ohem_pick_kernel, dim3(1), dim3(64), 0, st, s, shift, thresh);
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_ce_small_fwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* conf, float conf_thr,
                            const float* keep_thr, int K, int H, int W, int64_t* stats_, float* gtprob_out, int dtype, hipStream_t st) {
  if (K > CE_MAXK || K < 1 || B <= 0) return CSS_ERR_ARG;
  ce_acc_t* stats = (ce_acc_t*)stats_;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const size_t P = (size_t)B * H * W;
  static const bool no_tile = getenv("CSS_CE_NO_TILE") != nullptr;      // (A/B and parity tests: the gather kernel for every shape)
  if (!no_tile && K <= CES_KREG && 2 * (h - 1) <= (H - 1) && 2 * (w - 1) <= (W - 1)) {
    // up-sampling factor >= 2 (a 32x32 tile's footprint fits CES_FP^2 small pixels) and the classes fit in registers: tiled kernel
    const dim3 g(cdiv(W, CES_TW), cdiv(H, CES_TH), B);
    const size_t lds = (size_t)CES_FP * CES_FP * CES_KREG * sizeof(float);
    if (dtype == CSS_BF16)
      hipLaunchKernelGGL(ce_small_fwd_tile_kernel<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)small, ld, h, w, sh, sw, label, conf, conf_thr, keep_thr, K, H,
                         W, stats, gtprob_out);
    else if (dtype == CSS_F32)
      hipLaunchKernelGGL(ce_small_fwd_tile_kernel<float>, g, dim3(256), lds, st, (const float*)small, ld, h, w, sh, sw, label, conf, conf_thr, keep_thr, K, H, W,
                         stats, gtprob_out);
    else return CSS_ERR_DTYPE;
    CSS_CHECK_LAUNCH();
    return CSS_OK;
  }
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(ce_small_fwd_kernel<bf16_t>, dim3(ce_grid(P)), dim3(256), 0, st, (const bf16_t*)small, ld, h, w, sh, sw, label, conf, conf_thr,
                       keep_thr, K, P, H, W, stats, gtprob_out);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL(ce_small_fwd_kernel<float>, dim3(ce_grid(P)), dim3(256), 0, st, (const float*)small, ld, h, w, sh, sw, label, conf, conf_thr,
                       keep_thr, K, P, H, W, stats, gtprob_out);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ce_small_bwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* keep_thr, int K, int H, int W,
                            const float* coef, const float* gscale, int pos_only, float* dsmall, int dtype, hipStream_t st) {
  if (K > CE_MAXK || K < 1 || B <= 0) return CSS_ERR_ARG;
  if (2 * (h - 1) > (H - 1) || 2 * (w - 1) > (W - 1)) return CSS_ERR_ARG;      // needs an up-sampling factor >= 2 (tile footprint)
  if (dtype != CSS_BF16 && dtype != CSS_F32) return CSS_ERR_DTYPE;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int tx = cdiv(W, CES_TW), ty = cdiv(H, CES_TH);
  const size_t fpk = (size_t)CES_FP * CES_FP * K * sizeof(float);
  static const bool no_tile = getenv("CSS_CE_NO_TILE") != nullptr;
  // factor in [2, 4], classes in registers: the tiled kernel (footprint + four per-wave gradient copies of CES_KREG floats per small
  // pixel + one private cell per lane: 157.9 KiB of LDS)
  const bool tile = !no_tile && K <= CES_KREG && sh >= 0.2499f && sw >= 0.2499f;
  const int use_rmw = sh >= 0.2499f && sw >= 0.2499f && 5 * fpk <= 152 * 1024;   // logits + four per-wave gradient copies (133 KiB at K = 21)
  const size_t lds = tile ? ((size_t)5 * CES_FP * CES_FP + 64) * CES_KREG * sizeof(float) : (use_rmw ? 5 : 2) * fpk;
  CesColour col;
  col.nx = ces_colours(sw, tx);
  col.ny = ces_colours(sh, ty);
  for (col.cy = 0; col.cy < col.ny; ++col.cy)
    for (col.cx = 0; col.cx < col.nx; ++col.cx) {
      const dim3 g(cdiv(tx - col.cx, col.nx), cdiv(ty - col.cy, col.ny), B);
      if (tile) {
        if (dtype == CSS_BF16)
          hipLaunchKernelGGL(ce_small_bwd_tile_kernel<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)small, ld, h, w, sh, sw, label, keep_thr, K, H, W,
                             coef, gscale, pos_only, dsmall, col);
        else
          hipLaunchKernelGGL(ce_small_bwd_tile_kernel<float>, g, dim3(256), lds, st, (const float*)small, ld, h, w, sh, sw, label, keep_thr, K, H, W,
                             coef, gscale, pos_only, dsmall, col);
      } else {
        // (factors above 4 scatter with LDS float atomics inside the workgroup - use_rmw = 0: the one path of this file whose sums are
        // not ordered; css_amd.loss.fused_upsample_ok keeps the trainer off it)
        if (dtype == CSS_BF16)
          hipLaunchKernelGGL(ce_small_bwd_kernel<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)small, ld, h, w, sh, sw, label, keep_thr, K, H, W, coef,
                             gscale, pos_only, dsmall, use_rmw, col);
        else
          hipLaunchKernelGGL(ce_small_bwd_kernel<float>, g, dim3(256), lds, st, (const float*)small, ld, h, w, sh, sw, label, keep_thr, K, H, W, coef,
                             gscale, pos_only, dsmall, use_rmw, col);
      }
    }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
