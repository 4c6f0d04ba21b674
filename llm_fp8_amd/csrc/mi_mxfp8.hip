// K7: MXFP8 block quantise -- one E8M0 (power-of-two) scale per 32 consecutive elements along the
// GEMM contraction axis.  Both orientations in ONE pass over the bf16 input (2 B read, 2 B + 2/32 B
// written per element): the row-wise copy (blocks along the last dim) feeds fprop, the column-wise
// copy (blocks along the first dim, emitted transposed) feeds dgrad / wgrad.
// Scales are stored BLOCK-MAJOR ([K/32, rows]: all rows' scales of one 32-block are contiguous): the quantiser
// writes 8 bytes per lane instead of 8 single bytes and the GEMM stages one K-tile's scales of a 256-row tile as
// four 256-byte runs.
// Same tiling as the delayed-scaling cast: 128x128 tile per workgroup, 8x8 block per lane; a
// 32-element block spans 4 neighbouring lanes, reduced with two DPP shuffles.
// Replaces TE's MXFP8 quantise under MXFP8BlockScaling(fp8_format=E4M3)
// (te_llama_mxfp8.py:28-29,86,93; SURVEY.md 2.3 K7, Appendix A "MXFP8").
#include "mi_common.h"

namespace mi {

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

template <int FMT, bool ROWWISE, bool COLWISE>
__global__ __launch_bounds__(256) void mxfp8_quant_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y_row,
                                                          uint8_t* __restrict__ s_row, uint8_t* __restrict__ y_colT,
                                                          uint8_t* __restrict__ s_colT, int rows, int cols,
                                                          int tiles_c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
  const int r0 = tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int c0 = tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  const float rcp = 1.0f / fp8_max_of<FMT>();
  const bool active = (r0 < rows) && (c0 < cols);
  float f[8][8];  // values as cast
  float g[8][8];  // |values| with NaN -> 0, for the block amax (fmaxf semantics)
  u32 screen = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v4i raw = {0, 0, 0, 0};
    if (active) raw = *reinterpret_cast<const v4i*>(x + (int64_t)(r0 + i) * cols + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 w = (u32)raw[j];
      screen = nan_screen(screen, w);
      f[i][2 * j] = __uint_as_float(w << 16);
      f[i][2 * j + 1] = __uint_as_float(w & 0xFFFF0000u);
    }
  }
  if (__builtin_expect(nan_seen(screen), 0)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) g[i][j] = (f[i][j] != f[i][j]) ? 0.0f : fabsf(f[i][j]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) g[i][j] = fabsf(f[i][j]);
  }
  // NOTE: shuffles below are executed by every lane (inactive lanes carry zeros); rows/cols are
  // multiples of 32 so a 4-lane block group is all-active or all-inactive.
  if (ROWWISE) {
    u32 sbytes[8];
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a = fmaxf(a, g[i][j]);
      a = fmaxf(a, __shfl_xor(a, 1));
      a = fmaxf(a, __shfl_xor(a, 2));
      u32 e = e8m0_roundup(a * rcp);
      float inv = e8m0_inv(e);
      sbytes[i] = e;
      lo[i] = cvt4_fp8<FMT>(f[i][0] * inv, f[i][1] * inv, f[i][2] * inv, f[i][3] * inv);
      hi[i] = cvt4_fp8<FMT>(f[i][4] * inv, f[i][5] * inv, f[i][6] * inv, f[i][7] * inv);
    }
    if (active) {
      uint8_t* dst = y_row + (int64_t)r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2*>(dst + (int64_t)i * cols) = make_uint2(lo[i], hi[i]);
      if ((lane & 3) == 0) {  // block-major scales [cols/32, rows]: this lane's 8 rows are 8 contiguous bytes
        const u32 lo4 = sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24);
        const u32 hi4 = sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24);
        *reinterpret_cast<uint2*>(s_row + (int64_t)(c0 / 32) * rows + r0) = make_uint2(lo4, hi4);
      }
    }
  }
  if (COLWISE) {
    u32 sbytes[8];
    float inv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) a = fmaxf(a, g[i][j]);
      a = fmaxf(a, __shfl_xor(a, 8));
      a = fmaxf(a, __shfl_xor(a, 16));
      u32 e = e8m0_roundup(a * rcp);
      sbytes[j] = e;
      inv[j] = e8m0_inv(e);
    }
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      lo[i] = cvt4_fp8<FMT>(f[i][0] * inv[0], f[i][1] * inv[1], f[i][2] * inv[2], f[i][3] * inv[3]);
      hi[i] = cvt4_fp8<FMT>(f[i][4] * inv[4], f[i][5] * inv[5], f[i][6] * inv[6], f[i][7] * inv[7]);
    }
    if (active) {
      u32 a[4], b[4], c[4], d[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
      uint8_t* dst = y_colT + (int64_t)c0 * rows + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<uint2*>(dst + (int64_t)j * rows) = make_uint2(a[j], b[j]);
        *reinterpret_cast<uint2*>(dst + (int64_t)(j + 4) * rows) = make_uint2(c[j], d[j]);
      }
      if (((lane >> 3) & 3) == 0) {  // block-major scales [rows/32, cols]: 8 contiguous bytes for this lane's 8 columns
        const u32 lo4 = sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24);
        const u32 hi4 = sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24);
        *reinterpret_cast<uint2*>(s_colT + (int64_t)(r0 / 32) * cols + c0) = make_uint2(lo4, hi4);
      }
    }
  }
}

template <int FMT>
static int launch_mx(const void* x, void* y_row, void* s_row, void* y_colT, void* s_colT, int64_t rows, int64_t cols,
                     hipStream_t st) {
  const int tiles_r = (int)((rows + 127) / 128), tiles_c = (int)((cols + 127) / 128);
  dim3 grid((unsigned)(tiles_r * tiles_c)), block(256);
  const uint16_t* xp = (const uint16_t*)x;
  uint8_t *yr = (uint8_t*)y_row, *sr = (uint8_t*)s_row, *yc = (uint8_t*)y_colT, *sc = (uint8_t*)s_colT;
  if (y_row && y_colT)
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, true, true>), grid, block, 0, st, xp, yr, sr, yc, sc, (int)rows, (int)cols, tiles_c);
  else if (y_row)
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, true, false>), grid, block, 0, st, xp, yr, sr, yc, sc, (int)rows, (int)cols, tiles_c);
  else
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, false, true>), grid, block, 0, st, xp, yr, sr, yc, sc, (int)rows, (int)cols, tiles_c);
  MI_CHECK_LAUNCH("mi_mxfp8_quantize launch");
  return MI_OK;
}

}  // namespace mi

extern "C" int mi_mxfp8_quantize(const void* x_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT,
                                 int64_t rows, int64_t cols, int fmt, void* stream) {
  MI_CHECK_ARG(x_bf16, "mi_mxfp8_quantize: null input");
  MI_CHECK_ARG((y_row && s_row) || (y_colT && s_colT), "mi_mxfp8_quantize: need (y_row,s_row) and/or (y_colT,s_colT)");
  MI_CHECK_ARG((!y_row) == (!s_row) && (!y_colT) == (!s_colT), "mi_mxfp8_quantize: data and scale pointers must pair up");
  MI_CHECK_ARG(rows >= 0 && cols >= 0 && rows % 32 == 0 && cols % 32 == 0,
               "mi_mxfp8_quantize: rows (%lld) and cols (%lld) must be multiples of 32", (long long)rows, (long long)cols);
  MI_CHECK_ARG(rows < (1LL << 31) && cols < (1LL << 31), "mi_mxfp8_quantize: shape too large");
  MI_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)y_row % 8) == 0 && ((uintptr_t)y_colT % 8) == 0,
               "mi_mxfp8_quantize: misaligned pointer");
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "mi_mxfp8_quantize: bad fmt %d", fmt);
  if (rows == 0 || cols == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3) return mi::launch_mx<MI_FMT_E4M3>(x_bf16, y_row, s_row, y_colT, s_colT, rows, cols, st);
  return mi::launch_mx<MI_FMT_E5M2>(x_bf16, y_row, s_row, y_colT, s_colT, rows, cols, st);
}
