// Shared helpers for the gfx950 kernels of libmiseg.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "miseg.h"

// The inference kernels are compiled twice: once for bf16 storage (namespace miseg, the format
// the training path uses too) and once, with -DMISEG_BUILD_F16=1, for IEEE fp16 storage (namespace
// miseg_f16).  Same kernels, same MFMA rate, same bytes; fp16 keeps 11 significand bits instead
// of 8, which is what strict-argmax mask parity with an fp32 run needs.  Everything format
// dependent goes through the h16_* helpers and the mfma_* wrappers below.
#ifndef MISEG_BUILD_F16
#define MISEG_BUILD_F16 0
#endif
#if MISEG_BUILD_F16
#define MISEG_KNS miseg_f16
#else
#define MISEG_KNS miseg
#endif

namespace miseg {
void set_error(const char* fmt, ...);
}

namespace MISEG_KNS {
using ::miseg::set_error;

#define MISEG_CHECK_ARG(cond, ...)                    \
  do {                                                \
    if (!(cond)) {                                    \
      ::miseg::set_error(__VA_ARGS__);                \
      return MISEG_E_INVALID_ARG;                     \
    }                                                 \
  } while (0)

#define MISEG_CHECK_HIP(expr)                                                     \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      ::miseg::set_error("%s failed: %s", #expr, hipGetErrorString(e__));         \
      return MISEG_E_HIP;                                                         \
    }                                                                             \
  } while (0)

typedef unsigned short bf16_t;  // storage type

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ float bf16lo_to_f32(unsigned packed) {
  return __uint_as_float(packed << 16);
}
__device__ __forceinline__ float bf16hi_to_f32(unsigned packed) {
  return __uint_as_float(packed & 0xffff0000u);
}
// round-to-nearest-even; the plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 v = {lo, hi};
  bf16x2 b = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, b);
}


// ---- the 16-bit storage format of this compilation ----
typedef unsigned short h16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#if MISEG_BUILD_F16
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
// fp16 has no room above 65504: saturate instead of producing infinities
__device__ __forceinline__ float h16_clamp(float f) { return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f); }
// ReLU and the saturation in one v_med3_f32 (what a plain max costs on the bf16 side); feed the
// results to pack_h16x2_raw
__device__ __forceinline__ float h16_relu(float f) { return __builtin_amdgcn_fmed3f(f, 0.f, 65504.f); }
__device__ __forceinline__ float h16_act(float f, bool relu) {
  return __builtin_amdgcn_fmed3f(f, relu ? 0.f : -65504.f, 65504.f);
}
__device__ __forceinline__ float h16_to_f32(h16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float h16lo_to_f32(unsigned packed) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(packed & 0xffffu));
}
__device__ __forceinline__ float h16hi_to_f32(unsigned packed) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(packed >> 16));
}
__device__ __forceinline__ h16_t f32_to_h16(float f) {
  return __builtin_bit_cast(h16_t, (_Float16)h16_clamp(f));        // round-to-nearest-even
}
__device__ __forceinline__ unsigned pack_h16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  f16x2 b = {(_Float16)h16_clamp(lo), (_Float16)h16_clamp(hi)};
  return __builtin_bit_cast(unsigned, b);
}
// for values that went through h16_relu / h16_act
__device__ __forceinline__ unsigned pack_h16x2_raw(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  f16x2 b = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ f32x16_t mfma_32x32x16(h16x8 a, h16x8 b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_16x16x32(h16x8 a, h16x8 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 h16x8;
__device__ __forceinline__ float h16_to_f32(h16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float h16lo_to_f32(unsigned packed) { return bf16lo_to_f32(packed); }
__device__ __forceinline__ float h16hi_to_f32(unsigned packed) { return bf16hi_to_f32(packed); }
__device__ __forceinline__ h16_t f32_to_h16(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ unsigned pack_h16x2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ unsigned pack_h16x2_raw(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ float h16_relu(float f) { return fmaxf(f, 0.f); }
__device__ __forceinline__ float h16_act(float f, bool relu) { return relu ? fmaxf(f, 0.f) : f; }
__device__ __forceinline__ f32x16_t mfma_32x32x16(h16x8 a, h16x8 b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_16x16x32(h16x8 a, h16x8 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif

}  // namespace MISEG_KNS
