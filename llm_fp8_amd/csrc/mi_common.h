// Shared device/host helpers for libmi_fp8 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mi_fp8.h"

namespace mi {

// thread-local error string behind mi_last_error()
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define MI_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mi::set_error(__VA_ARGS__);          \
      return MI_ERR_ARG;                   \
    }                                      \
  } while (0)

#define MI_CHECK_LAUNCH(what)                                   \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return mi::hip_fail(e__, what);      \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

__device__ __forceinline__ float bf16_bits_to_float(u32 b) { return __uint_as_float(b << 16); }

// fp32 -> bf16 bits, RNE, NaN kept (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u32 float_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return (u32)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ u32 pack_bf16x2(float lo, float hi) {
  return float_to_bf16_bits(lo) | (float_to_bf16_bits(hi) << 16);
}

template <int FMT>
__device__ __forceinline__ float fp8_max_of() { return FMT == MI_FMT_E4M3 ? 448.0f : 57344.0f; }

// Two fp32 -> two fp8 bytes in the low half of the result.  Inputs are clamped to +-max first,
// so the hardware converter only ever sees in-range finite values (RNE); NaN is patched to 0x7F
// by the caller.
template <int FMT>
__device__ __forceinline__ u32 cvt_pk_fp8(float a, float b) {
  const float mx = fp8_max_of<FMT>();
  a = __builtin_amdgcn_fmed3f(a, -mx, mx);
  b = __builtin_amdgcn_fmed3f(b, -mx, mx);
  if (FMT == MI_FMT_E4M3) return (u32)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu;
  return (u32)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xFFFFu;
}

// 4 fp32 -> one dword of 4 fp8 bytes (byte i = v[i]); NaN -> 0x7F.
template <int FMT>
__device__ __forceinline__ u32 cvt4_fp8(float v0, float v1, float v2, float v3) {
  const float mx = fp8_max_of<FMT>();
  float c0 = __builtin_amdgcn_fmed3f(v0, -mx, mx), c1 = __builtin_amdgcn_fmed3f(v1, -mx, mx);
  float c2 = __builtin_amdgcn_fmed3f(v2, -mx, mx), c3 = __builtin_amdgcn_fmed3f(v3, -mx, mx);
  int r = 0;
  if (FMT == MI_FMT_E4M3) {
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c2, c3, r, true);
  } else {
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c0, c1, r, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c2, c3, r, true);
  }
  u32 u = (u32)r;
  // rare path: canonical NaN byte
  if (__builtin_expect((v0 != v0) | (v1 != v1) | (v2 != v2) | (v3 != v3), 0)) {
    if (v0 != v0) u = (u & 0xFFFFFF00u) | 0x0000007Fu;
    if (v1 != v1) u = (u & 0xFFFF00FFu) | 0x00007F00u;
    if (v2 != v2) u = (u & 0xFF00FFFFu) | 0x007F0000u;
    if (v3 != v3) u = (u & 0x00FFFFFFu) | 0x7F000000u;
  }
  return u;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

}  // namespace mi
