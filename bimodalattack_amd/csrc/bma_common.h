// Shared device helpers for the bma kernels (gfx950 only: wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bma.h"

#define BMA_WAVE 64

#define BMA_LAUNCH_CHECK()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return BMA_ELAUNCH; \
  } while (0)

namespace bma {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b16) { return __uint_as_float(b16 << 16); }

__device__ __forceinline__ float f16_bits_to_f32(uint32_t h16) {
  return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(h16)));
}

// round-to-nearest-even f32 -> bf16 bits.  A plain cast: hipcc emits gfx950's v_cvt_pk_bf16_f32
// (one instruction per PAIR of values, NaN stays NaN) instead of ~7 integer ops per value.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<__bf16>(f)));
}

__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(f)));
}

// element `i` (0/1) of a packed pair of 16-bit values
template <int DT>
__device__ __forceinline__ float unpack16(uint32_t w, int i) {
  const uint32_t h = i ? (w >> 16) : (w & 0xffffu);
  return DT == BMA_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ uint32_t pack16(float lo, float hi) {
  if (DT == BMA_BF16) {
    float2_t v;
    v.x = lo;
    v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
  }
  return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16);
}

template <int DT>
struct elem_bytes { static constexpr int value = (DT == BMA_F32) ? 4 : 2; };

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, BMA_WAVE));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, BMA_WAVE);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, BMA_WAVE);
  return v;
}

}  // namespace bma
