// Embedding x class-prototype cosine similarity and the pseudo-labelling that consumes it.
// Replaces (reference file:line)
//   generalframeworks/networks/ddp_model.py:104-110  teacher  normalize(rep_u) @ normalize(prototypes).T
//   generalframeworks/networks/ddp_model.py:147-154  student  prob_all = softmax(sim / temp)
//   generalframeworks/networks/ddp_model.py:111-118  bilinear x4 + softmax + max (rep and cls space) + agreement mask
//   mix_label.py:175-183 + generalframeworks/utils.py:116-136  label_all / mask_all assembly (as a class-id map)
//
// similarity_kernel: HBM-bound (reads each embedding row once); the (pixels x C)·(C x K) contraction
// runs on MFMA (32x32x16 bf16 / 32x32x2 f32) with the prototypes as the "A" operand, so a lane owns one
// pixel and 16 of the 32 padded classes; its partner lane (+32) owns the other 16.
#include "common.h"
#include <cstdlib>

template <typename T> struct SimMma;
template <> struct SimMma<bf16_t> {
  static constexpr int KS = 16;
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag load(const bf16_t* p, int h) { return *reinterpret_cast<const frag*>(p + h * 8); }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct SimMma<float> {
  static constexpr int KS = 2;
  typedef float frag;
  static __device__ __forceinline__ frag load(const float* p, int h) { return p[h]; }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

// prototypes fp32 [K][C] -> L2-normalised (F.normalize, eps 1e-12) T [32][C], rows >= K zero
template <typename T>
__global__ __launch_bounds__(64) void proto_normalize_kernel(const float* __restrict__ proto, T* __restrict__ out, int K, int C) {
  const int k = blockIdx.x, lane = threadIdx.x;
  float ss = 0.f;
  if (k < K)
    for (int c = lane; c < C; c += 64) { float v = proto[(size_t)k * C + c]; ss += v * v; }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int c = lane; c < C; c += 64) out[(size_t)k * C + c] = (T)(k < K ? proto[(size_t)k * C + c] * inv : 0.f);
}

// rep [P][ld] (T), pn [32][C] (T, normalised).  Outputs (any may be null):
//   sim  [P][K] fp32 cosine;  prob [P][K] fp32 softmax(sim / temp);
//   hard [P] uint8: (cls[p] >= 0 && prob[p][cls[p]] < strong_thr)   (fused student path, cls given)
template <typename T, int C>
__global__ __launch_bounds__(256) void similarity_kernel(const T* __restrict__ rep, int ld, const T* __restrict__ pn, float* __restrict__ sim,
                                                         float* __restrict__ prob, const int* __restrict__ cls, uint8_t* __restrict__ hard,
                                                         int P, int K, float inv_temp, float strong_thr) {
  using MM = SimMma<T>;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CH = 128 / sizeof(T);       // channels per staged chunk (128 B per row)
  constexpr int LPR = CH / VEC;             // lanes per row = 8
  constexpr int STR = CH + VEC;
  constexpr int PSTR = C + VEC;
  __shared__ __attribute__((aligned(16))) T ps[32 * PSTR];
  __shared__ __attribute__((aligned(16))) T xs[4][32 * STR];
  __shared__ float nrm[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * (C / VEC); i += 256) {
    const int r = i / (C / VEC), cv = i - r * (C / VEC);
    *reinterpret_cast<uint4*>(ps + r * PSTR + cv * VEC) = *reinterpret_cast<const uint4*>(pn + (size_t)r * C + cv * VEC);
  }
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntiles = (P + 127) / 128;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * 128 + wave * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ss[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's partial sum of squares for rows (lane/LPR + 8*i)
    T* xw = xs[wave];
    for (int c0 = 0; c0 < C; c0 += CH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = lane / LPR + 8 * i, cv = lane % LPR;
        Vec16<T> v;
        if (p0 + row < P) v.load(rep + (size_t)(p0 + row) * ld + c0 + cv * VEC);
        else v.zero();
#pragma unroll
        for (int e = 0; e < VEC; ++e) { float f = v.f(e); ss[i] += f * f; }
        v.store(xw + row * STR + cv * VEC);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
#pragma unroll
      for (int ks = 0; ks < CH / MM::KS; ++ks) {
        typename MM::frag fp = MM::load(ps + l31 * PSTR + c0 + ks * MM::KS, lh);
        typename MM::frag fx = MM::load(xw + l31 * STR + ks * MM::KS, lh);
        acc = MM::mma(fp, fx, acc);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    // row norms: reduce over the 8 lanes of a row, publish through LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = ss[i];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((lane % LPR) == 0) nrm[wave][lane / LPR + 8 * i] = s;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const float inv = 1.f / fmaxf(sqrtf(nrm[wave][l31]), 1e-12f);   // F.normalize eps
    const int p = p0 + l31;
    // this lane: classes (r&3) + 8*(r>>2) + 4*lh ; partner lane^32 has the rest
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
      v[r] = acc[r] * inv;
      if (k < K) mx = fmaxf(mx, v[r] * inv_temp);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (k < K) se += __expf(v[r] * inv_temp - mx);
    }
    se += __shfl_xor(se, 32, 64);
    const float inv_se = 1.f / se;
    const int mycls = (cls && p < P) ? cls[p] : -1;
    int is_hard = 0;
    if (p < P) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < K) {
          const float pr = __expf(v[r] * inv_temp - mx) * inv_se;
          if (sim) sim[(size_t)p * K + k] = v[r];
          if (prob) prob[(size_t)p * K + k] = pr;
          if (k == mycls && pr < strong_thr) is_hard = 1;
        }
      }
    }
    if (hard) {
      is_hard |= __shfl_xor(is_hard, 32, 64);
      if (lh == 0 && p < P) hard[p] = (uint8_t)is_hard;
    }
  }
}

// ---- teacher pseudo labels (ddp_model.py:111-118) --------------------------------------
// sim [B][h][w][K] fp32 (cosine), pred [B][h][w][ldp] (T); outputs at [B][H][W].
template <typename T>
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ sim, const T* __restrict__ pred, int ldp, int B, int h, int w,
                                                           int K, int H, int W, float inv_temp, float sh, float sw,
                                                           float* __restrict__ logits_rep, int64_t* __restrict__ labels_rep,
                                                           float* __restrict__ logits_cls, int64_t* __restrict__ labels_cls,
                                                           float* __restrict__ pseudo) {
  const size_t total = (size_t)B * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    size_t t = idx / W;
    const int y = (int)(t % H), b = (int)(t / H);
    float fy = sh * (float)y, fx = sw * (float)x;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const size_t o00 = ((size_t)(b * h + y0) * w + x0), o01 = ((size_t)(b * h + y0) * w + x1);
    const size_t o10 = ((size_t)(b * h + y1) * w + x0), o11 = ((size_t)(b * h + y1) * w + x1);
    // rep space: softmax(sim_large / temp)
    float mr = -INFINITY, mc = -INFINITY;
    int ar = 0, ac = 0;
    float sr = 0.f, sc = 0.f;
    // online softmax denominators (single pass, rescaled on a new max)
    for (int k = 0; k < K; ++k) {
      const float vr = (wy0 * (wx0 * sim[o00 * K + k] + wx1 * sim[o01 * K + k]) + wy1 * (wx0 * sim[o10 * K + k] + wx1 * sim[o11 * K + k])) * inv_temp;
      const float vc = wy0 * (wx0 * (float)pred[o00 * ldp + k] + wx1 * (float)pred[o01 * ldp + k]) +
                       wy1 * (wx0 * (float)pred[o10 * ldp + k] + wx1 * (float)pred[o11 * ldp + k]);
      if (vr > mr) { sr = sr * __expf(mr - vr) + 1.f; mr = vr; ar = k; } else sr += __expf(vr - mr);
      if (vc > mc) { sc = sc * __expf(mc - vc) + 1.f; mc = vc; ac = k; } else sc += __expf(vc - mc);
    }
    logits_rep[idx] = 1.f / sr;
    labels_rep[idx] = ar;
    logits_cls[idx] = 1.f / sc;
    labels_cls[idx] = ac;
    if (pseudo) pseudo[idx] = (ar == ac) ? (float)ac : 255.f;
  }
}

// The same on 32x32 output tiles (grid: tiles_x, tiles_y, B) for an up-sampling factor >= 2: the tile's footprint in the two small maps
// (<= 18 x 18 cells x K) goes to LDS once, raw; a pixel's 4 corners x 2 maps x K values then come from LDS instead of 168 scalar loads from
// L2 per pixel (the kernel above: latency-bound, 415 us at c2 for 118 MB of output).  Same expressions in the same order per pixel: bit-identical.
constexpr int PL_T = 32, PL_FP = 18;
template <typename T>
__global__ __launch_bounds__(256) void pseudo_label_tile_kernel(const float* __restrict__ sim, const T* __restrict__ pred, int ldp, int h, int w, int K,
                                                                int H, int W, float inv_temp, float sh, float sw, float* __restrict__ logits_rep,
                                                                int64_t* __restrict__ labels_rep, float* __restrict__ logits_cls,
                                                                int64_t* __restrict__ labels_cls, float* __restrict__ pseudo) {
  extern __shared__ float pl_lds[];
  const int KS = K | 1;                                   // odd cell stride: the corners of neighbouring pixels start on different banks
  float* s_sim = pl_lds;
  float* s_pred = pl_lds + PL_FP * PL_FP * KS;
  const int tid = threadIdx.x, b = blockIdx.z;
  const int X0 = blockIdx.x * PL_T, Y0 = blockIdx.y * PL_T;
  const int X1 = min(X0 + PL_T, W) - 1, Y1 = min(Y0 + PL_T, H) - 1;
  int fy0 = (int)(sh * (float)Y0), fx0 = (int)(sw * (float)X0);
  fy0 = min(fy0, h - 1); fx0 = min(fx0, w - 1);
  const int fy1 = min((int)(sh * (float)Y1) + 1, h - 1), fx1 = min((int)(sw * (float)X1) + 1, w - 1);
  const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;       // <= PL_FP by the launcher's factor check
  for (int i = tid; i < fh * fw * K; i += 256) {
    const int k = i % K, c = i / K, yy = c / fw, xx = c - yy * fw;
    const size_t g = (size_t)(b * h + fy0 + yy) * w + fx0 + xx;
    s_sim[c * KS + k] = sim[g * K + k];
    s_pred[c * KS + k] = (float)pred[g * ldp + k];
  }
  __syncthreads();
  const int x = X0 + (tid & 31);
#pragma unroll
  for (int trip = 0; trip < PL_T / 8; ++trip) {
    const int y = Y0 + (tid >> 5) + 8 * trip;
    if (x >= W || y >= H) continue;
    const size_t idx = ((size_t)b * H + y) * W + x;
    float fy = sh * (float)y, fx = sw * (float)x;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const int o00 = ((y0 - fy0) * fw + (x0 - fx0)) * KS, o01 = ((y0 - fy0) * fw + (x1 - fx0)) * KS;
    const int o10 = ((y1 - fy0) * fw + (x0 - fx0)) * KS, o11 = ((y1 - fy0) * fw + (x1 - fx0)) * KS;
    float mr = -INFINITY, mc = -INFINITY;
    int ar = 0, ac = 0;
    float sr = 0.f, sc = 0.f;
    for (int k = 0; k < K; ++k) {
      const float vr = (wy0 * (wx0 * s_sim[o00 + k] + wx1 * s_sim[o01 + k]) + wy1 * (wx0 * s_sim[o10 + k] + wx1 * s_sim[o11 + k])) * inv_temp;
      const float vc = wy0 * (wx0 * s_pred[o00 + k] + wx1 * s_pred[o01 + k]) + wy1 * (wx0 * s_pred[o10 + k] + wx1 * s_pred[o11 + k]);
      if (vr > mr) { sr = sr * __expf(mr - vr) + 1.f; mr = vr; ar = k; } else sr += __expf(vr - mr);
      if (vc > mc) { sc = sc * __expf(mc - vc) + 1.f; mc = vc; ac = k; } else sc += __expf(vc - mc);
    }
    logits_rep[idx] = 1.f / sr;
    labels_rep[idx] = ar;
    logits_cls[idx] = 1.f / sc;
    labels_cls[idx] = ac;
    if (pseudo) pseudo[idx] = (ar == ac) ? (float)ac : 255.f;
  }
}

// ---- class-id / validity map at embedding resolution (mix_label.py:175-183) ----------------
// cls[p] for the labeled half:   l_lab >= 0 ? l_lab : -1                    (mask = l_lab>=0, label = onehot(relu))
//        for the unlabeled half: (u_lab >= 0 && u_logits >= weak) ? u_lab : -1  (onehot_2 drops channel 0 = label -1)
// nearest down-sampling F.interpolate(mode='nearest'): src = floor(dst * H / h)
__global__ __launch_bounds__(256) void class_map_kernel(const int64_t* __restrict__ l_lab, const int64_t* __restrict__ u_lab,
                                                        const float* __restrict__ u_logits, float weak_thr, int B, int H, int W, int h, int w,
                                                        float rh, float rw, int* __restrict__ cls) {
  const size_t total = (size_t)2 * B * h * w;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    size_t t = idx / w;
    const int y = (int)(t % h), b = (int)(t / h);
    int ys = (int)floorf((float)y * rh), xs = (int)floorf((float)x * rw);
    if (ys > H - 1) ys = H - 1;
    if (xs > W - 1) xs = W - 1;
    int c;
    if (b < B) {
      const int64_t l = l_lab[((size_t)b * H + ys) * W + xs];
      c = l >= 0 ? (int)l : -1;
    } else {
      const size_t o = ((size_t)(b - B) * H + ys) * W + xs;
      const int64_t l = u_lab[o];
      c = (l >= 0 && u_logits[o] >= weak_thr) ? (int)l : -1;
    }
    cls[idx] = c;
  }
}

// ---- hard flags from the student's own class logits (ori_pseudo.py:178-180: prob_all = softmax(pred_all), loss.py:90-91) -----
// hard[p] = cls[p] >= 0 && softmax(pred[p])[cls[p]] < strong_thr; pred: low-resolution logits [P][ld], one pixel per lane
template <typename T>
__global__ __launch_bounds__(256) void softmax_hard_kernel(const T* __restrict__ pred, int ld, const int* __restrict__ cls, int P, int K,
                                                           float strong_thr, uint8_t* __restrict__ hard) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int c = cls[p];
    const T* row = pred + (size_t)p * ld;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, (float)row[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf((float)row[k] - m);
    const float own = c >= 0 ? expf((float)row[c] - m) / s : 1.f;
    hard[p] = (c >= 0 && own < strong_thr) ? 1 : 0;
  }
}

// ---- launchers -----------------------------------------------------------
int css_launch_softmax_hard(const void* pred, int ld, const int* cls, int P, int K, float strong_thr, uint8_t* hard, int dtype, hipStream_t st) {
  if (P <= 0) return CSS_OK;
  if (K <= 0 || ld < K) return CSS_ERR_ARG;
  int grid = (P + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (dtype == CSS_BF16) hipLaunchKernelGGL(softmax_hard_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)pred, ld, cls, P, K, strong_thr, hard);
  else if (dtype == CSS_F32) hipLaunchKernelGGL(softmax_hard_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)pred, ld, cls, P, K, strong_thr, hard);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_proto_normalize(const float* proto, void* out, int K, int C, int dtype, hipStream_t st) {
  if (K > 32) return CSS_ERR_ARG;
  if (dtype == CSS_BF16) hipLaunchKernelGGL(proto_normalize_kernel<bf16_t>, dim3(32), dim3(64), 0, st, proto, (bf16_t*)out, K, C);
  else if (dtype == CSS_F32) hipLaunchKernelGGL(proto_normalize_kernel<float>, dim3(32), dim3(64), 0, st, proto, (float*)out, K, C);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_similarity(const void* rep, int ld, const void* pn, float* sim, float* prob, const int* cls, uint8_t* hard, int P, int K,
                          int C, float temp, float strong_thr, int dtype, int n_cu, hipStream_t st) {
  if (P <= 0) return CSS_OK;
  if (K > 32 || C != 256) return CSS_ERR_ARG;   // output_dim 256 (ddp_model.py:78); K <= 32 padded classes
  const int vec = dtype == CSS_BF16 ? 8 : 4;
  if (ld % vec || (reinterpret_cast<uintptr_t>(rep) & 15)) return CSS_ERR_ARG;
  int grid = (P + 127) / 128;
  if (grid > n_cu * 4) grid = n_cu * 4;
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL((similarity_kernel<bf16_t, 256>), dim3(grid), dim3(256), 0, st, (const bf16_t*)rep, ld, (const bf16_t*)pn, sim, prob,
                       cls, hard, P, K, 1.f / temp, strong_thr);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL((similarity_kernel<float, 256>), dim3(grid), dim3(256), 0, st, (const float*)rep, ld, (const float*)pn, sim, prob, cls,
                       hard, P, K, 1.f / temp, strong_thr);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_pseudo_label(const float* sim, const void* pred, int ldp, int B, int h, int w, int K, int H, int W, float temp,
                            float* logits_rep, int64_t* labels_rep, float* logits_cls, int64_t* labels_cls, float* pseudo, int dtype,
                            hipStream_t st) {
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const size_t total = (size_t)B * H * W;
  static const bool no_tile = getenv("CSS_PSEUDO_NO_TILE") != nullptr;       // (A/B and parity tests: the gather kernel for every shape)
  if (!no_tile && B > 0 && K <= 32 && 2 * (h - 1) <= (H - 1) && 2 * (w - 1) <= (W - 1) && (dtype == CSS_BF16 || dtype == CSS_F32)) {
    // up-sampling factor >= 2: a 32 x 32 tile's footprint fits PL_FP^2 cells
    const dim3 g(cdiv(W, PL_T), cdiv(H, PL_T), B);
    const size_t lds = (size_t)2 * PL_FP * PL_FP * (K | 1) * sizeof(float);
    if (dtype == CSS_BF16)
      hipLaunchKernelGGL(pseudo_label_tile_kernel<bf16_t>, g, dim3(256), lds, st, sim, (const bf16_t*)pred, ldp, h, w, K, H, W, 1.f / temp, sh, sw,
                         logits_rep, labels_rep, logits_cls, labels_cls, pseudo);
    else
      hipLaunchKernelGGL(pseudo_label_tile_kernel<float>, g, dim3(256), lds, st, sim, (const float*)pred, ldp, h, w, K, H, W, 1.f / temp, sh, sw,
                         logits_rep, labels_rep, logits_cls, labels_cls, pseudo);
    CSS_CHECK_LAUNCH();
    return CSS_OK;
  }
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(pseudo_label_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, sim, (const bf16_t*)pred, ldp, B, h, w, K, H, W, 1.f / temp,
                       sh, sw, logits_rep, labels_rep, logits_cls, labels_cls, pseudo);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL(pseudo_label_kernel<float>, dim3(grid), dim3(256), 0, st, sim, (const float*)pred, ldp, B, h, w, K, H, W, 1.f / temp, sh,
                       sw, logits_rep, labels_rep, logits_cls, labels_cls, pseudo);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_class_map(const int64_t* l_lab, const int64_t* u_lab, const float* u_logits, float weak_thr, int B, int H, int W, int h, int w,
                         int* cls, hipStream_t st) {
  const size_t total = (size_t)2 * B * h * w;
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  // torch 'nearest': scale = in/out (float), src = min(floor(dst*scale), in-1)
  hipLaunchKernelGGL(class_map_kernel, dim3(grid), dim3(256), 0, st, l_lab, u_lab, u_logits, weak_thr, B, H, W, h, w, (float)H / (float)h,
                     (float)W / (float)w, cls);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
