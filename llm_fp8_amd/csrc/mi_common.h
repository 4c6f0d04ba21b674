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
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

__device__ __forceinline__ float bf16_bits_to_float(u32 b) { return __uint_as_float(b << 16); }

// fp32 -> bf16 bits, RNE, NaN kept (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u32 float_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return (u32)__builtin_bit_cast(unsigned short, h);
}
// two fp32 -> packed bf16x2 (RNE, NaN kept): one v_cvt_pk_bf16_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pack_bf16x2(float lo, float hi) {
  f32x2 v = {lo, hi};
  return __builtin_bit_cast(u32, __builtin_convertvector(v, bf16x2));
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

// NaN screen on raw bf16 pairs: running packed-u16 max of the magnitudes; any lane value > 0x7F80
// afterwards means "this thread saw a NaN".  (v_max_f32 in IEEE mode turns max(a, sNaN) into qNaN
// and so forgets `a`; threads that saw a NaN take a sanitising slow path instead.)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 nan_screen(u32 acc, u32 w) {
  u16x2 a = __builtin_bit_cast(u16x2, acc), b = __builtin_bit_cast(u16x2, w & 0x7FFF7FFFu);
  return __builtin_bit_cast(u32, __builtin_elementwise_max(a, b));
}
__device__ __forceinline__ bool nan_seen(u32 acc) { return max(acc & 0xFFFFu, acc >> 16) > 0x7F80u; }

// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {lo = bytes 0..3, hi = bytes 4..7}
__device__ __forceinline__ u32 bperm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 4x4 byte transpose: in r[i] = row i (byte j = col j) -> out c[j] = col j (byte i = row i)
__device__ __forceinline__ void transpose4x4(u32 r0, u32 r1, u32 r2, u32 r3, u32& c0, u32& c1, u32& c2,
                                             u32& c3) {
  u32 t01l = bperm(r1, r0, 0x05010400u);  // [r0b0, r1b0, r0b1, r1b1]
  u32 t01h = bperm(r1, r0, 0x07030602u);  // [r0b2, r1b2, r0b3, r1b3]
  u32 t23l = bperm(r3, r2, 0x05010400u);
  u32 t23h = bperm(r3, r2, 0x07030602u);
  c0 = bperm(t23l, t01l, 0x05040100u);  // [t01l.b0, t01l.b1, t23l.b0, t23l.b1]
  c1 = bperm(t23l, t01l, 0x07060302u);
  c2 = bperm(t23h, t01h, 0x05040100u);
  c3 = bperm(t23h, t01h, 0x07060302u);
}

// E8M0 biased exponent of (amax * 1/fp8_max), rounded up to the next power of two.
__device__ __forceinline__ u32 e8m0_roundup(float val) {
  u32 u = __float_as_uint(val);
  u32 e = (u >> 23) & 0xFFu, man = u & 0x7FFFFFu;
  if (man > 0 && e != 0xFEu && !(e == 0 && man <= 0x400000u)) ++e;
  if (val != val) e = 0xFFu;
  else if (isinf(val)) e = 0xFEu;
  else if (val == 0.0f) e = 0u;
  return e;
}
// 2^(127 - e) as fp32 (exact; subnormal for e = 254, NaN for e = 255)
__device__ __forceinline__ float e8m0_inv(u32 e) {
  if (e == 0xFFu) return __uint_as_float(0x7FC00000u);
  if (e == 0xFEu) return __uint_as_float(0x00400000u);
  return __uint_as_float((254u - e) << 23);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

}  // namespace mi
// 8-byte FP8 stores of the cast / quantise kernels (one 8 x 8 block per lane: 8 lanes = one 64-byte row segment).
// MI_NT_Y / MI_NT_YT (build-time, timing experiments): nontemporal stores for the row-major / the transposed copy.
#ifndef MI_NT_Y
#define MI_NT_Y 0
#endif
#ifndef MI_NT_YT
#define MI_NT_YT 0
#endif
#if defined(__HIPCC__)
namespace mi {
template <bool NT>
__device__ __forceinline__ void st8(uint8_t* p, unsigned int a, unsigned int b) {
  typedef unsigned int v2u_st8 __attribute__((ext_vector_type(2)));
  const v2u_st8 w = {a, b};
  if (NT) __builtin_nontemporal_store(w, reinterpret_cast<v2u_st8*>(p));
  else *reinterpret_cast<v2u_st8*>(p) = w;
}
}  // namespace mi
#endif

