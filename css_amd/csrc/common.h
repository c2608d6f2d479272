// Shared device/host helpers for the CSS hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { CSS_F32 = 0, CSS_BF16 = 1 };

// error codes returned through the C ABI (0 = ok)
enum {
  CSS_OK = 0,
  CSS_ERR_ARG = -1,      // bad shape / alignment / unsupported configuration
  CSS_ERR_DTYPE = -2,
  CSS_ERR_LAUNCH = -3,   // hipGetLastError() != success after launch
  CSS_ERR_WORKSPACE = -4
};

#define CSS_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return CSS_ERR_LAUNCH;            \
  } while (0)

// ---- division by a runtime constant (valid for numerators < 2^31) -------
struct FastDiv {
  uint32_t mul, shr, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) {
    f.mul = 0;
    f.shr = 0;
    return f;
  }
  uint32_t lg = 31 - __builtin_clz(d);
  if (d & (d - 1)) lg += 1;  // ceil(log2 d)
  uint32_t p = 31 + lg;
  f.mul = (uint32_t)((((uint64_t)1 << p) + d - 1) / d);
  f.shr = p - 32;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

// ---- element helpers -----------------------------------------------------
template <typename T> struct ElemT;
template <> struct ElemT<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct ElemT<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

// 16-byte vector of T with float views
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  union {
    uint4 raw;
    T e[N];
  };
  __device__ __forceinline__ Vec16() {}
  __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ float f(int i) const { return ElemT<T>::to_f(e[i]); }
  __device__ __forceinline__ void set(int i, float v) { e[i] = ElemT<T>::from_f(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
