// Shared helpers for the gfx950 kernels of libmiseg.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "miseg.h"

namespace miseg {

void set_error(const char* fmt, ...);

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

}  // namespace miseg
