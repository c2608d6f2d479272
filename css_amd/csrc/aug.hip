// In-step augmentation of the unlabeled batch on the device: what the reference does by a
// GPU -> CPU -> PIL -> GPU round trip per image inside Model_*.forward (dataset_helpers/VOC.py: tensor_to_pil_2 :284-291,
// transform_2 :126-196, batch_transform_2 :339-352).  Integer / byte work on 8-bit planes, restated bit-exactly:
//   aug_geom    : denormalise + 8-bit quantisation of the image and of the two confidence maps (to_pil_image),
//                 PIL BILINEAR resize (two-pass, 22-bit fixed-point coefficients, antialiased when shrinking) of the image,
//                 PIL NEAREST resize of label / confidence maps, pad (reflect / 255 / 0) at the right and bottom, crop
//   aug_finish  : horizontal flip, to_tensor (q/255), ImageNet normalisation, label 255 -> -1 (int64)
// Every random draw (scale, crop offset, flip ...) is made on the host and passed in: the kernels are deterministic.
#include "common.h"
#include "launchers.h"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;   // PIL Resample.c

__device__ __forceinline__ unsigned char quant_image(float x, int c) {
  // denormalise (VOC.py:309-314): normalize(x, 0, 1/std) then normalize(., -mean, 1); to_pil_image: mul(255).byte()
  const float inv_std[3] = {(float)(1 / 0.229), (float)(1 / 0.224), (float)(1 / 0.225)};
  const float neg_mean[3] = {-0.485f, -0.456f, -0.406f};
  float d = __fdiv_rn(__fsub_rn(x, 0.f), inv_std[c]);
  d = __fdiv_rn(__fsub_rn(d, neg_mean[c]), 1.f);
  d = __fmul_rn(d, 255.f);
  d = fminf(fmaxf(d, 0.f), 255.f);            // (.byte() of an out-of-range float is undefined in the reference)
  return (unsigned char)d;
}
__device__ __forceinline__ unsigned char quant_unit(float x) {   // to_pil_image of a [0,1] map: mul(255).byte()
  float d = __fmul_rn(x, 255.f);
  d = fminf(fmaxf(d, 0.f), 255.f);
  return (unsigned char)d;
}
__device__ __forceinline__ unsigned char quant_label(float l) {  // label.float()/255 -> mul(255).byte(); -1 wraps to 255
  if (l < 0.f) return 255;
  float d = __fmul_rn(__fdiv_rn(l, 255.f), 255.f);
  d = fminf(d, 255.f);
  return (unsigned char)d;
}
__device__ __forceinline__ int reflect_index(int p, int n) {     // np.pad(mode='reflect') continued periodically
  if (n <= 1) return 0;
  const int period = 2 * (n - 1);
  int m = p % period;
  return m < n ? m : period - m;
}
// PIL precompute_coeffs for the bilinear (triangle, support 1) filter: taps [xmin, xmin+cnt) and their 22-bit weights
struct Taps { int xmin, cnt; int k[8]; };
__device__ __forceinline__ Taps pil_bilinear_taps(int xx, int in_size, int out_size) {
  Taps t;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > 8) xmax = 8;                      // support <= 2 + rounding: at most 6 taps for scales >= 0.5 (launcher checks)
  double w[8], ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (x + xmin - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    w[x] = a < 1.0 ? 1.0 - a : 0.0;
    ww += w[x];
  }
  for (int x = 0; x < xmax; ++x) {
    double v = ww != 0.0 ? w[x] / ww : w[x];
    t.k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
  }
  t.xmin = xmin;
  t.cnt = xmax;
  return t;
}
__device__ __forceinline__ int clip8_fixed(int ss) {
  int v = ss >> PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
}  // namespace

// PIL NEAREST resize (ImagingScaleAffine): xin = (int)(xo), xo starting at 0.5*a and ADVANCED BY REPEATED ADDITION of
// a = in/out in double - reproduced as such (one thread per (image, axis)) so that exact .5 boundaries fall the same way.
__global__ void aug_nearest_table_kernel(const int* __restrict__ params, int B, int H, int W, int maxlen, int* __restrict__ tab) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2 * B) return;
  const int b = id >> 1, axis = id & 1;
  const int out = params[b * 4 + axis], in = axis == 0 ? H : W;
  int* t = tab + (size_t)id * maxlen;
  if (out == in) {                             // PIL returns a copy without resampling
    for (int x = 0; x < out; ++x) t[x] = x;
    return;
  }
  const double a = (double)in / (double)out;
  double xo = 0.0 + a * 0.5;
  for (int x = 0; x < out; ++x) {
    int xin = xo < 0.0 ? -1 : (int)xo;
    t[x] = xin < in ? xin : in - 1;
    xo += a;
  }
}

// one thread per output (crop) pixel: 3 image channels + label + 2 confidence maps
__global__ __launch_bounds__(256) void aug_geom_kernel(const float* __restrict__ img, const float* __restrict__ label, const float* __restrict__ l1,
                                                       const float* __restrict__ l2, const int* __restrict__ params,
                                                       const int* __restrict__ tab, int maxlen, int B, int H, int W, int Hc, int Wc,
                                                       unsigned char* __restrict__ img_q, unsigned char* __restrict__ lab_q,
                                                       unsigned char* __restrict__ l1_q, unsigned char* __restrict__ l2_q) {
  const size_t total = (size_t)B * Hc * Wc;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wc);
    size_t t = idx / Wc;
    const int y = (int)(t % Hc), b = (int)(t / Hc);
    const int rh = params[b * 4 + 0], rw = params[b * 4 + 1], ci = params[b * 4 + 2], cj = params[b * 4 + 3];
    const int py = y + ci, px = x + cj;                      // position in the (padded) rescaled image
    const bool inside = py < rh && px < rw;
    const size_t plane = (size_t)H * W;
    // label / confidence maps: constant padding, nearest sampling
    unsigned char lq = 255, q1 = 0, q2 = 0;
    if (inside) {
      const int sy = tab[(size_t)(2 * b) * maxlen + py], sx = tab[(size_t)(2 * b + 1) * maxlen + px];
      const size_t o = (size_t)b * plane + (size_t)sy * W + sx;
      lq = quant_label(label[o]);
      q1 = quant_unit(l1[o]);
      q2 = quant_unit(l2[o]);
    }
    lab_q[idx] = lq;
    l1_q[idx] = q1;
    l2_q[idx] = q2;
    // image: reflect padding of the RESIZED image, two-pass PIL bilinear
    const int ry = py < rh ? py : reflect_index(py, rh), rx = px < rw ? px : reflect_index(px, rw);
    for (int c = 0; c < 3; ++c) {
      const float* src = img + ((size_t)b * 3 + c) * plane;
      int out;
      if (rh == H && rw == W) {
        out = quant_image(src[(size_t)ry * W + rx], c);
      } else {
        const Taps tx = pil_bilinear_taps(rx, W, rw), ty = pil_bilinear_taps(ry, H, rh);
        int acc = 1 << (PRECISION_BITS - 1);
        for (int r = 0; r < ty.cnt; ++r) {
          const float* row = src + (size_t)(ty.xmin + r) * W;
          int h;
          if (rw == W) {
            h = quant_image(row[rx], c);                     // no horizontal pass when the width is unchanged
          } else {
            int ss = 1 << (PRECISION_BITS - 1);
            for (int k = 0; k < tx.cnt; ++k) ss += (int)quant_image(row[tx.xmin + k], c) * tx.k[k];
            h = clip8_fixed(ss);
          }
          acc += h * ty.k[r];
        }
        if (rh == H) {                                        // no vertical pass when the height is unchanged
          int ss = 1 << (PRECISION_BITS - 1);
          const float* row = src + (size_t)ry * W;
          for (int k = 0; k < tx.cnt; ++k) ss += (int)quant_image(row[tx.xmin + k], c) * tx.k[k];
          out = clip8_fixed(ss);
        } else {
          out = clip8_fixed(acc);
        }
      }
      img_q[((size_t)b * 3 + c) * Hc * Wc + (size_t)y * Wc + x] = (unsigned char)out;
    }
  }
}

// flip + to_tensor + normalise; flags[b] bit 0 = horizontal flip
__global__ __launch_bounds__(256) void aug_finish_kernel(const unsigned char* __restrict__ img_q, const unsigned char* __restrict__ lab_q,
                                                         const unsigned char* __restrict__ l1_q, const unsigned char* __restrict__ l2_q,
                                                         const int* __restrict__ flags, int B, int Hc, int Wc, float* __restrict__ img,
                                                         int64_t* __restrict__ label, float* __restrict__ l1, float* __restrict__ l2) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const size_t total = (size_t)B * Hc * Wc;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wc);
    size_t t = idx / Wc;
    const int y = (int)(t % Hc), b = (int)(t / Hc);
    const int sx = (flags[b] & 1) ? Wc - 1 - x : x;
    const size_t so = ((size_t)b * Hc + y) * Wc + sx;
    const int lq = lab_q[so];
    label[idx] = lq == 255 ? -1 : lq;
    l1[idx] = __fdiv_rn((float)l1_q[so], 255.f);
    l2[idx] = __fdiv_rn((float)l2_q[so], 255.f);
    for (int c = 0; c < 3; ++c) {
      const float v = __fdiv_rn((float)img_q[((size_t)b * 3 + c) * Hc * Wc + (size_t)y * Wc + sx], 255.f);
      img[((size_t)b * 3 + c) * Hc * Wc + (size_t)y * Wc + x] = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);
    }
  }
}

// ---- colour jitter and Gaussian blur on the uint8 RGB planes ----------------------------------------------------------
// torchvision ColorJitter on a PIL image = PIL ImageEnhance.{Brightness, Contrast, Color} (all three are Image.blend of a
// "degenerate" image with the input, Blend.c) and a hue shift in PIL's 8-bit HSV space (Convert.c rgb2hsv / hsv2rgb),
// applied in a random order; ImageFilter.GaussianBlur = three box-blur passes per axis with a fractional radius (BoxBlur.c),
// for sigma <= 1.15 a 3-tap integer filter per pass.  jp int32 [B][16]:
//   0 jitter on, 1..4 op order (0 brightness, 1 contrast, 2 saturation, 3 hue), 5..7 factors (float bits), 8 hue shift (uint8),
//   9 blur on, 10 ww, 11 fw (box-blur centre / neighbour weights, 24-bit fixed point)
namespace {
__device__ __forceinline__ int pil_blend(int in1, int in2, float alpha) {            // Blend.c
  const float temp = __fadd_rn((float)in1, __fmul_rn(alpha, (float)(in2 - in1)));
  if (alpha >= 0.f && alpha <= 1.f) return (int)(unsigned char)temp;
  if (temp <= 0.f) return 0;
  if (temp >= 255.f) return 255;
  return (int)(unsigned char)temp;
}
__device__ __forceinline__ int pil_luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c L24
__device__ __forceinline__ void pil_rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
  us = is < 0 ? 0 : (is > 255 ? 255 : is);
}
__device__ __forceinline__ int clip8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ void pil_hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) { r = g = b = v; return; }
  const double hh = (double)(float)h * 6.0 / 255.0;
  const int i = (int)floor(hh);
  const float f = (float)(hh - (double)(float)i);
  const float fs = (float)((double)(float)s / 255.0);
  const int p = clip8i((int)round((double)(float)v * (1.0 - (double)fs)));
  const int q = clip8i((int)round((double)(float)v * (1.0 - (double)fs * (double)f)));
  const int t = clip8i((int)round((double)(float)v * (1.0 - (double)fs * (1.0 - (double)f))));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}
// the ops of one image's jitter, op index range [first, last); `mean_l` = rounded mean luma for the contrast op
__device__ __forceinline__ void jitter_ops(int& r, int& g, int& b, const int* __restrict__ jp, int first, int last, int mean_l) {
  for (int k = first; k < last; ++k) {
    const int op = jp[1 + k];
    if (op == 0) {
      const float f = __int_as_float(jp[5]);
      r = pil_blend(0, r, f); g = pil_blend(0, g, f); b = pil_blend(0, b, f);
    } else if (op == 1) {
      const float f = __int_as_float(jp[6]);
      r = pil_blend(mean_l, r, f); g = pil_blend(mean_l, g, f); b = pil_blend(mean_l, b, f);
    } else if (op == 2) {
      const float f = __int_as_float(jp[7]);
      const int l = pil_luma(r, g, b);
      r = pil_blend(l, r, f); g = pil_blend(l, g, f); b = pil_blend(l, b, f);
    } else {
      int h, s, v;
      pil_rgb2hsv(r, g, b, h, s, v);
      h = (h + jp[8]) & 0xff;
      pil_hsv2rgb(h, s, v, r, g, b);
    }
  }
}
}  // namespace

// sum of the luma of every image as it stands right before its contrast op (ImageStat.Stat(image.convert('L')).mean)
__global__ __launch_bounds__(256) void aug_luma_sum_kernel(const unsigned char* __restrict__ img_q, const int* __restrict__ jp, int B, int HW,
                                                           unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const int* p = jp + b * 16;
  if (!p[0]) return;
  int cpos = 0;
  while (cpos < 4 && p[1 + cpos] != 1) ++cpos;
  unsigned long long acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    int r = img_q[((size_t)b * 3 + 0) * HW + i], g = img_q[((size_t)b * 3 + 1) * HW + i], bl = img_q[((size_t)b * 3 + 2) * HW + i];
    jitter_ops(r, g, bl, p, 0, cpos, 0);
    acc += (unsigned long long)pil_luma(r, g, bl);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sums[b], acc);
}
__global__ __launch_bounds__(256) void aug_jitter_kernel(unsigned char* __restrict__ img_q, const int* __restrict__ jp, int B, int HW,
                                                         const unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const int* p = jp + b * 16;
  if (!p[0]) return;
  const int mean_l = (int)((double)sums[b] / (double)HW + 0.5);      // int(stat.mean[0] + 0.5)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    unsigned char* pr = img_q + ((size_t)b * 3 + 0) * HW + i;
    unsigned char* pg = img_q + ((size_t)b * 3 + 1) * HW + i;
    unsigned char* pb = img_q + ((size_t)b * 3 + 2) * HW + i;
    int r = *pr, g = *pg, bl = *pb;
    jitter_ops(r, g, bl, p, 0, 4, mean_l);
    *pr = (unsigned char)r; *pg = (unsigned char)g; *pb = (unsigned char)bl;
  }
}
// three box-blur passes along one axis (BoxBlur.c ImagingLineBoxBlur8 with integer radius 0): each pass is
// out = (c*ww + (l + r)*fw + 2^23) >> 24 with the line's edge pixels replicated; recursion on a 7-pixel neighbourhood
__global__ __launch_bounds__(256) void aug_blur_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, const int* __restrict__ jp,
                                                       int B, int H, int W, int vertical) {
  const size_t total = (size_t)B * 3 * H * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    size_t t = idx / W;
    const int y = (int)(t % H);
    const int bc = (int)(t / H), b = bc / 3;
    const int* p = jp + b * 16;
    if (!p[9]) { out[idx] = in[idx]; continue; }
    const unsigned ww = (unsigned)p[10], fw = (unsigned)p[11];
    const int n = vertical ? H : W, pos = vertical ? y : x;
    const size_t stride = vertical ? (size_t)W : 1;
    const unsigned char* line = in + (idx - (size_t)pos * stride);
    unsigned v0[7], v1[7], v2[7];
    for (int k = 0; k < 7; ++k) {
      int q = pos - 3 + k;
      q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
      v0[k] = line[(size_t)q * stride];
    }
    // a value "at position q" of pass k for q outside [0, n) never exists: clamp the INDEX at every level
    auto at = [&](const unsigned* v, int q) {            // q relative index into the 7-window for absolute position pos-3+q
      int a = pos - 3 + q;
      a = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
      return v[a - (pos - 3)];
    };
    for (int k = 1; k < 6; ++k) v1[k] = (at(v0, k) * ww + (at(v0, k - 1) + at(v0, k + 1)) * fw + (1u << 23)) >> 24;
    for (int k = 2; k < 5; ++k) v2[k] = (at(v1, k) * ww + (at(v1, k - 1) + at(v1, k + 1)) * fw + (1u << 23)) >> 24;
    out[idx] = (unsigned char)((at(v2, 3) * ww + (at(v2, 2) + at(v2, 4)) * fw + (1u << 23)) >> 24);
  }
}

int css_launch_aug_color(unsigned char* img_q, unsigned char* tmp, const int* jp, unsigned long long* sums, int B, int H, int W, int any_jitter,
                         int any_blur, hipStream_t st) {
  if (B <= 0) return CSS_ERR_ARG;
  const int HW = H * W;
  if (any_jitter) {
    (void)hipMemsetAsync(sums, 0, sizeof(unsigned long long) * B, st);
    dim3 g(cdiv(HW, 256 * 4) < 1 ? 1 : cdiv(HW, 256 * 4), B);
    hipLaunchKernelGGL(aug_luma_sum_kernel, g, dim3(256), 0, st, img_q, jp, B, HW, sums);
    hipLaunchKernelGGL(aug_jitter_kernel, g, dim3(256), 0, st, img_q, jp, B, HW, sums);
  }
  if (any_blur) {
    const size_t total = (size_t)B * 3 * HW;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(aug_blur_kernel, dim3(grid), dim3(256), 0, st, img_q, tmp, jp, B, H, W, 0);
    hipLaunchKernelGGL(aug_blur_kernel, dim3(grid), dim3(256), 0, st, tmp, img_q, jp, B, H, W, 1);
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_aug_geom(const float* img, const float* label, const float* l1, const float* l2, const int* params, int* table, int maxlen, int B,
                        int H, int W, int Hc, int Wc, unsigned char* img_q, unsigned char* lab_q, unsigned char* l1_q, unsigned char* l2_q,
                        hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || Hc <= 0 || Wc <= 0 || maxlen < (H > W ? H : W)) return CSS_ERR_ARG;
  hipLaunchKernelGGL(aug_nearest_table_kernel, dim3(cdiv(2 * B, 64)), dim3(64), 0, st, params, B, H, W, maxlen, table);
  const size_t total = (size_t)B * Hc * Wc;
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(aug_geom_kernel, dim3(grid), dim3(256), 0, st, img, label, l1, l2, params, table, maxlen, B, H, W, Hc, Wc, img_q, lab_q, l1_q,
                     l2_q);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

int css_launch_aug_finish(const unsigned char* img_q, const unsigned char* lab_q, const unsigned char* l1_q, const unsigned char* l2_q,
                          const int* flags, int B, int Hc, int Wc, float* img, int64_t* label, float* l1, float* l2, hipStream_t st) {
  if (B <= 0) return CSS_ERR_ARG;
  const size_t total = (size_t)B * Hc * Wc;
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(aug_finish_kernel, dim3(grid), dim3(256), 0, st, img_q, lab_q, l1_q, l2_q, flags, B, Hc, Wc, img, label, l1, l2);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

// ---- cutmix / cutout boxes of a whole batch in ONE launch per tensor (generate_cut_gather*, dataset_helpers/VOC.py:354-477) -----------------
// out[b][p][y][x] = inside box b ? (mode 0: partner[pj[b]][p][y][x], mode 1: fill) : self[b][p][y][x]; boxes int32 [B][4] = {y0, y1, x0, x1}
// (half-open), tensors contiguous [B][P][H][W] of 4- or 8-byte elements.  Replaces a clone + one strided copy per image and tensor (64 copy
// launches of ~5 us per c2 step); a pure copy, bit-exact.
template <typename E>
__global__ __launch_bounds__(256) void mix_boxes_kernel(const E* __restrict__ self, const E* __restrict__ partner, E* __restrict__ out,
                                                        const int* __restrict__ boxes, const int* __restrict__ pj, int B, int P, int H, int W,
                                                        int mode, E fill) {
  const size_t plane = (size_t)H * W, per_img = plane * P, total = per_img * B;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / per_img);
    const size_t r = idx - (size_t)b * per_img;
    const int yx = (int)(r % plane), y = yx / W, x = yx - y * W;
    const int* bx = boxes + 4 * b;
    const bool in = y >= bx[0] && y < bx[1] && x >= bx[2] && x < bx[3];
    out[idx] = in ? (mode == 0 ? partner[(size_t)pj[b] * per_img + r] : fill) : self[idx];
  }
}
int css_launch_mix_boxes(const void* self, const void* partner, void* out, const int* boxes, const int* pj, int B, int P, int H, int W, int elem_bytes,
                         int mode, long long fill_bits, hipStream_t st) {
  if (B <= 0 || P <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1) || (mode == 0 && (!partner || !pj))) return CSS_ERR_ARG;
  const size_t total = (size_t)B * P * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  if (elem_bytes == 4) {
    const unsigned f = (unsigned)fill_bits;
    hipLaunchKernelGGL(mix_boxes_kernel<unsigned>, dim3(grid), dim3(256), 0, st, (const unsigned*)self, (const unsigned*)partner, (unsigned*)out, boxes, pj, B,
                       P, H, W, mode, f);
  } else if (elem_bytes == 8) {
    hipLaunchKernelGGL(mix_boxes_kernel<unsigned long long>, dim3(grid), dim3(256), 0, st, (const unsigned long long*)self,
                       (const unsigned long long*)partner, (unsigned long long*)out, boxes, pj, B, P, H, W, mode, (unsigned long long)fill_bits);
  } else {
    return CSS_ERR_DTYPE;
  }
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}

