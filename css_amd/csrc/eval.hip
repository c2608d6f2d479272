// Evaluation path of the reference (mix_label.py:199-225, "next" row 8f-3 of SURVEY.md):
//   pred = F.interpolate(pred, size=label.shape[1:], mode='bilinear', align_corners=True)
//   ConfMatrix.update(pred.argmax(1).flatten(), label.flatten())        (util/meter.py:39-48)
// fused into one pass: the up-sampled [B,K,H,W] logits are never materialised (c2: 16 x 21 x 513^2 x 4 B = 354 MB per batch);
// each thread interpolates the K logits of one label pixel, takes the arg-max (first maximum, like torch.argmax) and votes
// into a per-workgroup K x K histogram in LDS, flushed with 64-bit global atomics.  Integer work: bit-exact.
#include "common.h"
#include "launchers.h"

constexpr int EVAL_MAX_K = 32;

template <typename T>
__global__ __launch_bounds__(256) void eval_confusion_kernel(const T* __restrict__ pred, int ldp, const int64_t* __restrict__ label, int B, int h,
                                                             int w, int K, int H, int W, float sh, float sw,
                                                             unsigned long long* __restrict__ mat, unsigned char* __restrict__ argmax_out) {
  __shared__ unsigned int hist[EVAL_MAX_K * EVAL_MAX_K];
  for (int i = threadIdx.x; i < K * K; i += 256) hist[i] = 0;
  __syncthreads();
  const size_t total = (size_t)B * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    size_t t = idx / W;
    const int y = (int)(t % H), b = (int)(t / H);
    // torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1), lambda1 = src - floor(src)
    float fy = sh * (float)y, fx = sw * (float)x;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wy0 = 1.f - wy1, wx1 = fx - (float)x0, wx0 = 1.f - wx1;
    const T* p00 = pred + ((size_t)(b * h + y0) * w + x0) * ldp;
    const T* p01 = pred + ((size_t)(b * h + y0) * w + x1) * ldp;
    const T* p10 = pred + ((size_t)(b * h + y1) * w + x0) * ldp;
    const T* p11 = pred + ((size_t)(b * h + y1) * w + x1) * ldp;
    float best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      const float v = wy0 * (wx0 * ElemT<T>::to_f(p00[k]) + wx1 * ElemT<T>::to_f(p01[k])) +
                      wy1 * (wx0 * ElemT<T>::to_f(p10[k]) + wx1 * ElemT<T>::to_f(p11[k]));
      if (v > best) { best = v; arg = k; }
    }
    if (argmax_out) argmax_out[idx] = (unsigned char)arg;
    const int64_t tgt = label[idx];
    if (tgt >= 0 && tgt < K) atomicAdd(&hist[(int)tgt * K + arg], 1u);     // k = (target >= 0) & (target < n)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256)
    if (hist[i]) atomicAdd(&mat[i], (unsigned long long)hist[i]);
}

// ConfMatrix.update(pred, target) with class indices already computed: mat[n*target + pred] += 1 over valid targets
__global__ __launch_bounds__(256) void confusion_bincount_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ label, size_t n, int K,
                                                                 unsigned long long* __restrict__ mat) {
  __shared__ unsigned int hist[EVAL_MAX_K * EVAL_MAX_K];
  for (int i = threadIdx.x; i < K * K; i += 256) hist[i] = 0;
  __syncthreads();
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int64_t tgt = label[idx], p = pred[idx];
    if (tgt >= 0 && tgt < K && p >= 0 && p < K) atomicAdd(&hist[(int)tgt * K + (int)p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256)
    if (hist[i]) atomicAdd(&mat[i], (unsigned long long)hist[i]);
}

int css_launch_eval_confusion(const void* pred, int ldp, const int64_t* label, int B, int h, int w, int K, int H, int W, int64_t* mat,
                              uint8_t* argmax_out, int dtype, hipStream_t st) {
  if (K <= 0 || K > EVAL_MAX_K || B <= 0) return CSS_ERR_ARG;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const size_t total = (size_t)B * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  unsigned long long* m = reinterpret_cast<unsigned long long*>(mat);
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(eval_confusion_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)pred, ldp, label, B, h, w, K, H, W, sh, sw, m, argmax_out);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL(eval_confusion_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)pred, ldp, label, B, h, w, K, H, W, sh, sw, m, argmax_out);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_confusion_bincount(const int64_t* pred, const int64_t* label, long n, int K, int64_t* mat, hipStream_t st) {
  if (K <= 0 || K > EVAL_MAX_K) return CSS_ERR_ARG;
  if (n <= 0) return CSS_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(confusion_bincount_kernel, dim3(grid), dim3(256), 0, st, pred, label, (size_t)n, K, reinterpret_cast<unsigned long long*>(mat));
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
