// K1/K2: bf16 -> FP8 cast (+ transposed copy) + per-tensor amax, delayed scaling.
//
// HBM-bound: 2 B read + 1 B (K1) or 2 B (K2) written per element.  One workgroup owns a
// 128x128 element tile (4 waves as 2x2, one 64x64 sub-tile per wave, one 8x8 block per lane).
//  - loads : 8 x 16 B per lane; the 8 lanes that share a row cover one full 128-B line.
//  - amax  : fmaxf in registers (NaN-screened: fmaxf semantics, NaN elements are ignored) -> wave shuffle (DPP) -> LDS across the 4 waves -> ONE
//            atomicMax per workgroup on the float's bit pattern (non-negative floats order
//            like unsigned ints).
//  - y     : 8 B per lane per row (64-B runs per wave, 128-B lines per workgroup).
//  - yT    : the lane's 8x8 byte block is transposed in registers with v_perm_b32 and stored
//            8 B per column, 64-B runs per wave and 128-B lines per workgroup.
// Replaces TE's cast / cast_transpose kernels on the reference path
// (te_llama.py:76-80 -> te.Linear forward/backward).
#include "mi_common.h"

namespace mi {

// COLSUM: also emits colsum[tile_r, c] = sum of the tile's (up to) 64 rows of x[:, c] in fp32 -- the bias gradient of a
// Linear comes out of the cast of its grad_output instead of a separate reduction pass over dy.
//
// Persistent form: every WAVE walks 64x64-element tiles (one 8x8 block per lane) with stride = waves in the grid, the 8
// row loads of its next tile already in flight while the current tile is converted and stored.  A one-shot grid (one tile
// per workgroup, all workgroups resident at once) runs in two global phases -- everybody loads, then everybody stores --
// and measured the same 25 us for 8192x3072 with or without the transposed copy; the loop overlaps the two directions.
// The 4 waves of a workgroup take 4 neighbouring column tiles, so a workgroup row-load covers 512 contiguous bytes.
template <int FMT, bool WRITE_Y, bool WRITE_T, bool COLSUM = false>
__global__ __launch_bounds__(256) void cast_amax_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y,
                                                        uint8_t* __restrict__ yT, const float* __restrict__ scale_p,
                                                        float* amax_out, int rows, int cols, int64_t ld_y,
                                                        int64_t ld_yT, int tiles_c, float* __restrict__ colsum = nullptr) {
  __shared__ float s_amax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = ((rows + 63) / 64) * tiles_c;
  const int stride = gridDim.x * 4;
  const int lr = (lane >> 3) * 8, lc = (lane & 7) * 8;
  const float scale = *scale_p;
  float amax = 0.0f;
  int t = blockIdx.x * 4 + wave;
  v4i nxt[8];
  auto load_tile = [&](int tt, v4i (&raw)[8]) {
    const int r0 = (tt / tiles_c) * 64 + lr, c0 = (tt % tiles_c) * 64 + lc;
    if (r0 < rows && c0 < cols) {  // dims are multiples of 8: blocks are all-in or all-out
      const uint16_t* src = x + (int64_t)r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src + (int64_t)i * cols));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = (v4i){0, 0, 0, 0};
    }
  };
  if (t < ntiles) load_tile(t, nxt);
  while (t < ntiles) {
    v4i raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = nxt[i];
    const int tn = t + stride;
    if (tn < ntiles) load_tile(tn, nxt);
    const int tile_r = t / tiles_c;
    const int r0 = tile_r * 64 + lr, c0 = (t % tiles_c) * 64 + lc;
    const bool active = (r0 < rows) && (c0 < cols);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32 screen = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) screen = nan_screen(screen, (u32)raw[i][j]);
    const bool has_nan = nan_seen(screen);
    u32 lo[8], hi[8];  // fp8 bytes of row i: lo = cols 0..3, hi = cols 4..7
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32 w = (u32)raw[i][j];
        f[2 * j] = __uint_as_float(w << 16);
        f[2 * j + 1] = __uint_as_float(w & 0xFFFF0000u);
      }
      if (__builtin_expect(has_nan, 0)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, (f[j] != f[j]) ? 0.0f : fabsf(f[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
      }
      if (COLSUM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += f[j];
      }
      lo[i] = cvt4_fp8<FMT>(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
      hi[i] = cvt4_fp8<FMT>(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
    }
    if (active) {
      if (WRITE_Y) {
        uint8_t* dst = y + (int64_t)r0 * ld_y + c0;
#pragma unroll
        for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * ld_y, lo[i], hi[i]);
      }
      if (WRITE_T) {
        u32 a[4], b[4], c[4], d[4];
        transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);  // cols 0..3, rows 0..3
        transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);  // cols 0..3, rows 4..7
        transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);  // cols 4..7, rows 0..3
        transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);  // cols 4..7, rows 4..7
        uint8_t* dst = yT + (int64_t)c0 * ld_yT + r0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mi::st8<MI_NT_YT>(dst + (int64_t)j * ld_yT, a[j], b[j]);
          mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * ld_yT, c[j], d[j]);
        }
      }
    }
    if (COLSUM) {  // inactive lanes carry zeros; lanes 0..7 end up with the 64-row sums of their 8 columns
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs[j] += __shfl_xor(cs[j], 8);
        cs[j] += __shfl_xor(cs[j], 16);
        cs[j] += __shfl_xor(cs[j], 32);
      }
      if (lane < 8 && c0 < cols) {
        float* dst = colsum + (int64_t)tile_r * cols + c0;
        *reinterpret_cast<v4f*>(dst) = (v4f){cs[0], cs[1], cs[2], cs[3]};
        *reinterpret_cast<v4f*>(dst + 4) = (v4f){cs[4], cs[5], cs[6], cs[7]};
      }
    }
    t = tn;
  }
  if (amax_out != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      float m = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
      // only a workgroup that would raise the value goes to the atomic unit: same-address atomics are served one at a time and a
      // wave cannot retire before its atomic has returned
      if (m > 0.0f && m > __builtin_nontemporal_load(amax_out)) atomicMax(reinterpret_cast<unsigned int*>(amax_out), __float_as_uint(m));
    }
  }
}

static int cast_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

// second stage of every [P, C] fp32 partial column sum (cast, dSwiGLU, RMSNorm backward): out[c] = sum_p part[p, c]
__device__ __forceinline__ void colsum_finish_block(const float* __restrict__ part, int P, int C, void* __restrict__ out,
                                                    int out_bf16, int block) {
  // 32 columns x 8 row-groups per workgroup: 128-byte row segments, 8 independent accumulation chains per column
  __shared__ float s_acc[8][32];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int c = block * 32 + cx;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (c < C) {
    int p = py;
    for (; p + 24 < P; p += 32) {
      a0 += part[(int64_t)p * C + c];
      a1 += part[(int64_t)(p + 8) * C + c];
      a2 += part[(int64_t)(p + 16) * C + c];
      a3 += part[(int64_t)(p + 24) * C + c];
    }
    for (; p < P; p += 8) a0 += part[(int64_t)p * C + c];
  }
  s_acc[py][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (py == 0 && c < C) {
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += s_acc[i][cx];
    if (out_bf16) reinterpret_cast<uint16_t*>(out)[c] = (uint16_t)float_to_bf16_bits(a);
    else reinterpret_cast<float*>(out)[c] = a;
  }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int P, int C, void* __restrict__ out,
                                                            int out_bf16) {
  colsum_finish_block(part, P, C, out, out_bf16, blockIdx.x);
}

// up to 4 finishes in one launch (a layer's bias gradients and its RMSNorm weight gradient: each is a 5-6 us kernel that is
// all launch latency): the descriptors travel as kernel arguments, a workgroup finds its own by block range
struct FinishDesc {
  const float* part;
  void* out;
  int P, C, out_bf16, block_end;  // block_end: first block index AFTER this descriptor's range
};
struct FinishArgs {
  FinishDesc d[4];
};
__global__ __launch_bounds__(256) void colsum_finish_multi_kernel(FinishArgs a, int n) {
  int i = 0;
  while (i + 1 < n && (int)blockIdx.x >= a.d[i].block_end) ++i;
  const int first = i == 0 ? 0 : a.d[i - 1].block_end;
  colsum_finish_block(a.d[i].part, a.d[i].P, a.d[i].C, a.d[i].out, a.d[i].out_bf16, (int)blockIdx.x - first);
}

template <int FMT>
static int launch_cast(const void* x, void* y, void* yT, const float* scale, float* amax, int64_t rows,
                       int64_t cols, int64_t ld_y, int64_t ld_yT, hipStream_t st, float* colsum = nullptr) {
  const int tiles_r = (int)((rows + 63) / 64), tiles_c = (int)((cols + 63) / 64);
  const int64_t wgs = ((int64_t)tiles_r * tiles_c + 3) / 4;
  // two workgroups per CU, a whole multiple of the CU count: 8192 x 3072 takes 24.1 us with 512 workgroups, 26.2 with 768, 27-28 with
  // 320-448 or 576-640 (uneven CU loads), 23.2 us one-shot but 26.3 when the amax slot starts at zero as it does in the step
  // (profiles/r03_cast_grid_sweep.txt)
  const int64_t cap2 = (int64_t)cast_num_cus() * 2;
  dim3 grid((unsigned)(wgs < cap2 ? wgs : cap2)), block(256);
  const uint16_t* xp = (const uint16_t*)x;
  uint8_t *yp = (uint8_t*)y, *tp = (uint8_t*)yT;
  if (colsum) {
    if (y && yT)
      hipLaunchKernelGGL((cast_amax_kernel<FMT, true, true, true>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                         (int)cols, ld_y, ld_yT, tiles_c, colsum);
    else if (y)
      hipLaunchKernelGGL((cast_amax_kernel<FMT, true, false, true>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                         (int)cols, ld_y, ld_yT, tiles_c, colsum);
    else
      hipLaunchKernelGGL((cast_amax_kernel<FMT, false, true, true>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                         (int)cols, ld_y, ld_yT, tiles_c, colsum);
    MI_CHECK_LAUNCH("mi_cast_amax_colsum launch");
    return MI_OK;
  }
  if (y && yT)
    hipLaunchKernelGGL((cast_amax_kernel<FMT, true, true>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                       (int)cols, ld_y, ld_yT, tiles_c, (float*)nullptr);
  else if (y)
    hipLaunchKernelGGL((cast_amax_kernel<FMT, true, false>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                       (int)cols, ld_y, ld_yT, tiles_c, (float*)nullptr);
  else
    hipLaunchKernelGGL((cast_amax_kernel<FMT, false, true>), grid, block, 0, st, xp, yp, tp, scale, amax, (int)rows,
                       (int)cols, ld_y, ld_yT, tiles_c, (float*)nullptr);
  MI_CHECK_LAUNCH("mi_cast_amax launch");
  return MI_OK;
}

}  // namespace mi

static int cast_amax_impl(const void* x_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                          int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, int fmt, void* stream, float* colsum) {
  MI_CHECK_ARG(x_bf16 && scale, "mi_cast_amax: x and scale must be non-null");
  MI_CHECK_ARG(y_fp8 || yT_fp8, "mi_cast_amax: at least one of y, yT must be non-null");
  MI_CHECK_ARG(rows >= 0 && cols >= 0, "mi_cast_amax: negative shape");
  MI_CHECK_ARG(rows % 8 == 0 && cols % 8 == 0, "mi_cast_amax: rows (%lld) and cols (%lld) must be multiples of 8",
               (long long)rows, (long long)cols);
  MI_CHECK_ARG(rows < (1LL << 31) && cols < (1LL << 31) && ((rows + 63) / 64) * ((cols + 63) / 64) < (1LL << 31),
               "mi_cast_amax: shape too large");
  MI_CHECK_ARG(!y_fp8 || (ld_y >= cols && ld_y % 8 == 0), "mi_cast_amax: ld_y must be >= cols and a multiple of 8");
  MI_CHECK_ARG(!yT_fp8 || (ld_yT >= rows && ld_yT % 8 == 0), "mi_cast_amax: ld_yT must be >= rows and a multiple of 8");
  MI_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)y_fp8 % 8) == 0 && ((uintptr_t)yT_fp8 % 8) == 0,
               "mi_cast_amax: x must be 16-byte aligned, y/yT 8-byte aligned");
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "mi_cast_amax: bad fmt %d", fmt);
  if (rows == 0 || cols == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3) return mi::launch_cast<MI_FMT_E4M3>(x_bf16, y_fp8, yT_fp8, scale, amax, rows, cols, ld_y, ld_yT, st, colsum);
  return mi::launch_cast<MI_FMT_E5M2>(x_bf16, y_fp8, yT_fp8, scale, amax, rows, cols, ld_y, ld_yT, st, colsum);
}

namespace mi {
// yT[c * ld_yT + r] = y[r * ld_y + c] for FP8 bytes: the transposed copy of an already quantised operand (after an FP8
// all-gather of row shards the wgrad / dgrad GEMMs still want the [K, N] copy: distributed.ShardedFP8DP).  Tiling of the cast
// kernels: 128 x 128 bytes per workgroup, an 8 x 8 block per lane, byte transpose in registers (v_perm_b32).
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t* __restrict__ y, uint8_t* __restrict__ yT, int64_t rows,
                                                           int64_t cols, int64_t ld_y, int64_t ld_yT, int tiles_c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
  const int64_t r0 = (int64_t)tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int64_t c0 = (int64_t)tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  if (r0 >= rows || c0 >= cols) return;
  u32 lo[8], hi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint2 v = *reinterpret_cast<const uint2*>(y + (r0 + i) * ld_y + c0);
    lo[i] = v.x;
    hi[i] = v.y;
  }
  u32 a[4], b[4], c[4], d[4];
  transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
  transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
  transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
  transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
  uint8_t* dst = yT + c0 * ld_yT + r0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mi::st8<MI_NT_YT>(dst + (int64_t)j * ld_yT, a[j], b[j]);
    mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * ld_yT, c[j], d[j]);
  }
}
}  // namespace mi

extern "C" int mi_transpose_u8(const void* y, void* yT, int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, void* stream) {
  MI_CHECK_ARG(y && yT, "mi_transpose_u8: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols >= 0 && rows % 8 == 0 && cols % 8 == 0 && ld_y >= cols && ld_yT >= rows && ld_y % 8 == 0 && ld_yT % 8 == 0,
               "mi_transpose_u8: rows, cols and leading dimensions must be multiples of 8 (got %lld x %lld)", (long long)rows, (long long)cols);
  MI_CHECK_ARG(((uintptr_t)y % 8) == 0 && ((uintptr_t)yT % 8) == 0, "mi_transpose_u8: pointers must be 8-byte aligned");
  if (rows == 0 || cols == 0) return MI_OK;
  const int64_t tiles_r = (rows + 127) / 128, tiles_c = (cols + 127) / 128;
  MI_CHECK_ARG(tiles_r * tiles_c < (1LL << 31), "mi_transpose_u8: too many tiles");
  hipLaunchKernelGGL(mi::transpose_u8_kernel, dim3((unsigned)(tiles_r * tiles_c)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)y,
                     (uint8_t*)yT, rows, cols, ld_y, ld_yT, (int)tiles_c);
  MI_CHECK_LAUNCH("mi_transpose_u8 launch");
  return MI_OK;
}

extern "C" int mi_cast_amax(const void* x_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                            int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, int fmt, void* stream) {
  return cast_amax_impl(x_bf16, y_fp8, yT_fp8, scale, amax, rows, cols, ld_y, ld_yT, fmt, stream, nullptr);
}

extern "C" int mi_cast_amax_colsum(const void* x_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                                   float* colsum_partial, int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, int fmt,
                                   void* stream) {
  MI_CHECK_ARG(colsum_partial, "mi_cast_amax_colsum: colsum_partial must be non-null");
  return cast_amax_impl(x_bf16, y_fp8, yT_fp8, scale, amax, rows, cols, ld_y, ld_yT, fmt, stream, colsum_partial);
}

extern "C" int mi_colsum_finish(const float* partial, int64_t P, int64_t C, void* out, int out_dtype, void* stream) {
  MI_CHECK_ARG(partial && out, "mi_colsum_finish: null pointer");
  MI_CHECK_ARG(P >= 1 && P < (1 << 24) && C >= 1 && C < (1LL << 31), "mi_colsum_finish: bad shape");
  MI_CHECK_ARG(out_dtype == MI_OUT_BF16 || out_dtype == MI_OUT_F32, "mi_colsum_finish: bad out_dtype %d", out_dtype);
  hipLaunchKernelGGL(mi::colsum_finish_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, (hipStream_t)stream, partial,
                     (int)P, (int)C, out, out_dtype == MI_OUT_BF16 ? 1 : 0);
  MI_CHECK_LAUNCH("mi_colsum_finish launch");
  return MI_OK;
}

extern "C" int mi_colsum_finish_multi(const void* const* partials, const int64_t* P, const int64_t* C, void* const* outs,
                                      const int* out_dtypes, int n, void* stream) {
  MI_CHECK_ARG(partials && P && C && outs && out_dtypes && n >= 1 && n <= 4, "mi_colsum_finish_multi: 1 to 4 finishes per launch");
  mi::FinishArgs a;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    MI_CHECK_ARG(partials[i] && outs[i], "mi_colsum_finish_multi: null pointer");
    MI_CHECK_ARG(P[i] >= 1 && P[i] < (1 << 24) && C[i] >= 1 && C[i] < (1LL << 31), "mi_colsum_finish_multi: bad shape");
    MI_CHECK_ARG(out_dtypes[i] == MI_OUT_BF16 || out_dtypes[i] == MI_OUT_F32, "mi_colsum_finish_multi: bad out_dtype %d", out_dtypes[i]);
    blocks += (int)((C[i] + 31) / 32);
    a.d[i] = {(const float*)partials[i], outs[i], (int)P[i], (int)C[i], out_dtypes[i] == MI_OUT_BF16 ? 1 : 0, blocks};
  }
  for (int i = n; i < 4; ++i) a.d[i] = {nullptr, nullptr, 0, 0, 0, blocks};
  hipLaunchKernelGGL(mi::colsum_finish_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, n);
  MI_CHECK_LAUNCH("mi_colsum_finish_multi launch");
  return MI_OK;
}
