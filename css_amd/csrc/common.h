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
    uint32_t w[4];
  };
  __device__ __forceinline__ Vec16() {}
  __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  // non-temporal forms (round 6): a streaming pass that reads a tensor ONCE should not allocate it in the caches on its way through -
  // bn_apply ran 15 % faster with them (profiles/r06_bn_nontemporal_ab.txt); `nt` false = the plain access (a uniform branch)
  typedef __attribute__((ext_vector_type(4))) unsigned int nt_u32x4;
  __device__ __forceinline__ void load(const T* p, bool nt) {
    if (nt) {
      const nt_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
      raw = make_uint4(t[0], t[1], t[2], t[3]);
    } else {
      load(p);
    }
  }
  __device__ __forceinline__ void store(T* p, bool nt) const {
    if (nt) {
      const nt_u32x4 t = {raw.x, raw.y, raw.z, raw.w};
      __builtin_nontemporal_store(t, reinterpret_cast<nt_u32x4*>(p));
    } else {
      store(p);
    }
  }
  // write-through store (sc1: the line is not kept in the XCD's L2 - MI355X_MICROARCH.md, stores of each flavour): the outputs of the streaming
  // batch-norm passes (70-280 MB, read next by another kernel from beyond L2 anyway): -0.5 ms in bn_apply, -0.35 in bn_bwd_apply per c2 step
  __device__ __forceinline__ void store_sc1(T* p) const {
    const nt_u32x4 t = {raw.x, raw.y, raw.z, raw.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
  }
  __device__ __forceinline__ float f(int i) const { return ElemT<T>::to_f(e[i]); }
  __device__ __forceinline__ void set(int i, float v) { e[i] = ElemT<T>::from_f(v); }
};

// Pair view of a Vec16: the elementwise kernels do their arithmetic on float2 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two
// elements per instruction - same rounding as the scalar forms) and convert two elements per v_cvt_pk_bf16_f32.
typedef __attribute__((ext_vector_type(2))) float f32x2;
// two floats -> two bf16 in one 32-bit word (element 0 in the low half) with ONE v_cvt_pk_bf16_f32: written as two scalar conversions the
// compiler emits two of them plus a shift and an or (r03: 135 instead of 32 instructions in conv_ws_kernel's epilogue); same rounding
typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2_pk;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2_pk));
}
template <typename T> struct Pairs;
template <> struct Pairs<float> {
  static constexpr int NP = 2;
  static __device__ __forceinline__ f32x2 get(const Vec16<float>& v, int p) { return f32x2{v.e[2 * p], v.e[2 * p + 1]}; }
  static __device__ __forceinline__ void set(Vec16<float>& v, int p, f32x2 x) { v.e[2 * p] = x[0]; v.e[2 * p + 1] = x[1]; }
};
template <> struct Pairs<bf16_t> {
  static constexpr int NP = 4;
  static __device__ __forceinline__ f32x2 get(const Vec16<bf16_t>& v, int p) {
    return f32x2{__builtin_bit_cast(float, v.w[p] << 16), __builtin_bit_cast(float, v.w[p] & 0xffff0000u)};
  }
  static __device__ __forceinline__ void set(Vec16<bf16_t>& v, int p, f32x2 x) { v.w[p] = pack2_bf16(x[0], x[1]); }
};
// x where bit `b` of bits is set, else +0 (bit test without a compare: a signed 1-bit field extract gives 0 / ~0)
__device__ __forceinline__ float keep_if_bit(float x, unsigned bits, int b) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & (unsigned)__builtin_amdgcn_sbfe((int)bits, b, 1));
}

// the same for a 32-bit word of two bf16 (elements 2w, 2w+1 of a 16-byte vector): 0xFFFF in every half whose bit is set
__device__ __forceinline__ unsigned keep_mask_bf16x2(unsigned bits, int w) {
  return ((unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * w, 1) & 0xFFFFu) | ((unsigned)__builtin_amdgcn_sbfe((int)bits, 2 * w + 1, 1) & 0xFFFF0000u);
}

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

// 16-byte buffer load: out-of-range offsets (>= num_records) return zeros in hardware, so the im2col zero padding,
// the M / Cout / K tails and the "ghost" prefetches past the last K tile need no branches, and the compiler can keep
// exact vmcnt counts for a prefetch distance of two tiles.  (conv.hip, conv_wgrad.hip)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
typedef __attribute__((address_space(3))) void lds_void;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
