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
// per-image accumulators: S = sum of losses, NPOS = #(loss > 0), NVALID = #(counted pixels), NCONF = #(conf >= thr)
enum { ST_S = 0, ST_NPOS = 1, ST_NVALID = 2, ST_NCONF = 3 };

template <bool BWD>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ label,
                                                 const float* __restrict__ conf, float conf_thr, const float* __restrict__ keep_thr,
                                                 int K, size_t P, int HW, double* __restrict__ stats, float* __restrict__ gtprob_out,
                                                 const float* __restrict__ coef, const float* __restrict__ gscale, int pos_only,
                                                 float* __restrict__ dlogits) {
  __shared__ float tile[256 * CE_MAXK];
  __shared__ float sacc[2][4];
  const int tid = threadIdx.x;
  const int KS = K | 1;  // odd LDS stride -> conflict-free per-pixel walks
  for (size_t p0 = (size_t)blockIdx.x * 256; p0 < P; p0 += (size_t)gridDim.x * 256) {
    const int np = (int)min((size_t)256, P - p0);
    __syncthreads();
    if (tid < 8) sacc[tid >> 2][tid & 3] = 0.f;
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
      bool valid = lab >= 0 && lab < K;
      float loss = 0.f, gtp = 1.f;
      if (valid) {
        const float xg = row[lab];
        loss = logf(se) + mx - xg;
        gtp = __expf(xg - mx) / se;
        if (keep_thr && !(gtp <= *keep_thr)) { valid = false; loss = 0.f; }
      }
      if (!BWD) {
        if (gtprob_out) gtprob_out[p] = (lab >= 0 && lab < K) ? gtp : 1.f;
        if (stats) {
          const int s = b - b_first;
          if (valid) {
            atomicAdd(&sacc[s][ST_S], loss);
            atomicAdd(&sacc[s][ST_NVALID], 1.f);
            if (loss > 0.f) atomicAdd(&sacc[s][ST_NPOS], 1.f);
          }
          if (conf && conf[p] >= conf_thr) atomicAdd(&sacc[s][ST_NCONF], 1.f);
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
        if (sacc[s][tid & 3] != 0.f) atomicAdd(&stats[(size_t)b * 4 + (tid & 3)], (double)sacc[s][tid & 3]);
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
__global__ void ce_finalize_kernel(const double* __restrict__ stats, int B, int mode, float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double S = 0, NV = 0, NP = 0, WS = 0;
  for (int b = 0; b < B; ++b) {
    S += stats[b * 4 + ST_S];
    NV += stats[b * 4 + ST_NVALID];
    NP += stats[b * 4 + ST_NPOS];
  }
  if (mode == 0) {
    *loss = (float)(S / NV);
    for (int b = 0; b < B; ++b) coef[b] = (float)(1.0 / NV);
  } else {
    for (int b = 0; b < B; ++b) {
      // weighting = #(logits >= thr) / #(label >= 0): fp32 division like the reference (int64 / float32)
      const float wgt = (float)stats[b * 4 + ST_NCONF] / (float)stats[b * 4 + ST_NVALID];
      if (stats[b * 4 + ST_NPOS] > 0) WS += (double)wgt * stats[b * 4 + ST_S];
      coef[b] = stats[b * 4 + ST_NPOS] > 0 ? (float)((double)wgt / NP) : 0.f;
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
__global__ void ohem_init_kernel(OhemState* s, const double* __restrict__ stats, int B, long P, int min_kept) {
  if (threadIdx.x >= 256) return;
  s->hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    double nv = 0;
    for (int b = 0; b < B; ++b) nv += stats[b * 4 + ST_NVALID];
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

// ---- launchers -----------------------------------------------------------
static inline int ce_grid(size_t P) {
  size_t g = (P + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
int css_launch_ce_fwd(const float* logits, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr, int K, long P,
                      int HW, double* stats, float* gtprob_out, hipStream_t st) {
  if (K > CE_MAXK || K < 1) return CSS_ERR_ARG;
  hipLaunchKernelGGL(ce_kernel<false>, dim3(ce_grid((size_t)P)), dim3(256), 0, st, logits, label, conf, conf_thr, keep_thr, K, (size_t)P, HW,
                     stats, gtprob_out, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ce_finalize(const double* stats, int B, int mode, float* loss, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, st, stats, B, mode, loss, coef);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_ce_bwd(const float* logits, const int64_t* label, const float* keep_thr, int K, long P, int HW, const float* coef,
                      const float* gscale, int pos_only, float* dlogits, hipStream_t st) {
  if (K > CE_MAXK || K < 1) return CSS_ERR_ARG;
  hipLaunchKernelGGL(ce_kernel<true>, dim3(ce_grid((size_t)P)), dim3(256), 0, st, logits, label, (const float*)nullptr, 0.f, keep_thr, K,
                     (size_t)P, HW, (double*)nullptr, (float*)nullptr, coef, gscale, pos_only, dlogits);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
// state: device buffer of css_ohem_state_bytes(); on return state->thr (offset css_ohem_thr_offset()) holds the keep threshold
size_t css_ohem_state_bytes_() { return sizeof(OhemState); }
size_t css_ohem_thr_offset_() { return offsetof(OhemState, thr); }
int css_launch_ohem_threshold(const float* gtprob, long P, const double* stats, int B, int min_kept, float thresh, void* state,
                              hipStream_t st) {
  OhemState* s = reinterpret_cast<OhemState*>(state);
  hipLaunchKernelGGL(ohem_init_kernel, dim3(1), dim3(256), 0, st, s, stats, B, P, min_kept);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(ce_grid((size_t)P)), dim3(256), 0, st, gtprob, (size_t)P, s, shift);
    hipLaunchKernelGGL(ohem_pick_kernel, dim3(1), dim3(64), 0, st, s, shift, thresh);
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
