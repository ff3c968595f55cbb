// Shared device helpers for the gfx950 kernels of libmvhip.so.
// wave = 64 lanes everywhere; "row" = one 16-lane DPP row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvhip.h"

#define MV_WAVE 64
// explicit address spaces for accesses through pointers whose provenance the compiler cannot see (pointers read from
// memory, pointers carried in structs): without them such accesses become flat_ instructions
#define MV_AS1 __attribute__((address_space(1)))
#define MV_AS3 __attribute__((address_space(3)))

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// NOTE: never __builtin_bit_cast an ext-vector ELEMENT lvalue (v[i]) directly: clang
// (ROCm 7.2) reads the vector's first element for every i.  Go through these by-value helpers.
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ bf16x2_t u2bf2(uint32_t u) { return __builtin_bit_cast(bf16x2_t, u); }
__device__ __forceinline__ f16x2_t u2h2(uint32_t u) { return __builtin_bit_cast(f16x2_t, u); }

struct bf16_tag {};
struct f16_tag {};
struct f32_tag {};

// ---- scalar conversions -------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) {
  return __builtin_bit_cast(float, hi16 << 16);
}
// round-to-nearest-even, NaN preserved (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(uint16_t, h);
}

// Element traits: T = storage type, 16-byte vector holds VEC elements.
template <typename Tag> struct elem;
template <> struct elem<bf16_tag> {
  typedef uint16_t T;
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float ld(uint16_t v) { return bf16_bits_to_f32(v); }
  static __device__ __forceinline__ uint16_t st(float f) { return f32_to_bf16_bits(f); }
};
template <> struct elem<f16_tag> {
  typedef uint16_t T;
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float ld(uint16_t v) { return f16_bits_to_f32(v); }
  static __device__ __forceinline__ uint16_t st(float f) { return f32_to_f16_bits(f); }
};
template <> struct elem<f32_tag> {
  typedef float T;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float f) { return f; }
};

// Unpack one 16-byte vector (8 x 16-bit) into 8 floats.
template <typename Tag>
__device__ __forceinline__ void unpack8(const u32x4 &v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (__is_same(Tag, bf16_tag)) {
      f[2 * i] = u2f(v[i] << 16);
      f[2 * i + 1] = u2f(v[i] & 0xffff0000u);
    } else {
      f[2 * i] = f16_bits_to_f32((uint16_t)(v[i] & 0xffffu));
      f[2 * i + 1] = f16_bits_to_f32((uint16_t)(v[i] >> 16));
    }
  }
}
template <typename Tag>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t lo = elem<Tag>::st(f[2 * i]);
    uint32_t hi = elem<Tag>::st(f[2 * i + 1]);
    v[i] = lo | (hi << 16);
  }
  return v;
}

// ---- cross-lane ----------------------------------------------------------
// DPP row rotate within each 16-lane row; ctrl = 0x120 + n (row_ror:n).
template <int N>
__device__ __forceinline__ float row_ror(float x) {
  int xi = __builtin_bit_cast(int, x);
  int r = __builtin_amdgcn_update_dpp(0, xi, 0x120 + N, 0xf, 0xf, false);
  return __builtin_bit_cast(float, r);
}
// Sum over the 16 lanes of a DPP row; every lane of the row gets the total.
__device__ __forceinline__ float row16_sum(float x) {
  x += row_ror<8>(x);
  x += row_ror<4>(x);
  x += row_ror<2>(x);
  x += row_ror<1>(x);
  return x;
}
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, row_ror<8>(x));
  x = fmaxf(x, row_ror<4>(x));
  x = fmaxf(x, row_ror<2>(x));
  x = fmaxf(x, row_ror<1>(x));
  return x;
}
// Full 64-lane reductions (every lane gets the result).
__device__ __forceinline__ float wave_sum(float x) {
  x = row16_sum(x);
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
  x = row16_max(x);
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  x = fmaxf(x, __shfl_xor(x, 32, 64));
  return x;
}

// ---- host-side helpers ------------------------------------------------------
static inline int mv_check_hip(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
static inline bool mv_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

#define MV_LAUNCH_CHECK()                       \
  do {                                          \
    hipError_t _e = hipGetLastError();          \
    if (_e != hipSuccess) return (int)_e;       \
  } while (0)

#define MV_DISPATCH_DTYPE(dtype, ...)                     \
  switch (dtype) {                                        \
    case MV_BF16: { typedef bf16_tag Tag; __VA_ARGS__; } break; \
    case MV_F16:  { typedef f16_tag Tag;  __VA_ARGS__; } break; \
    case MV_F32:  { typedef f32_tag Tag;  __VA_ARGS__; } break; \
    default: return MV_E_DTYPE;                           \
  }
