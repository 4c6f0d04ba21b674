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

// PRE selects what is quantised (the value block f[8][8] of this lane), so the neighbours of the GEMMs fuse into
// the quantiser exactly as they do on the delayed-scaling path (K9 / K10):
//   0  x[r,c]                                               x: [rows, cols]
//   1  (x[r,c] * rstd[r]) * gamma[c]          (RMSNorm)     x: [rows, cols], aux32 = rstd, aux16 = gamma
//   2  silu(h[r,c]) * h[r,F+c]                (SwiGLU)      x = h: [rows, 2*cols]
//   3  dSwiGLU: [d*dsilu(g)*u | d*silu(g)]                  x = h: [rows, cols] (cols = 2F), aux16 = d: [rows, F];
//      colsum[tile_r, c] = per-128-row column sums (fc1 bias gradient, fixed order)
__device__ __forceinline__ float mx_sigmoid(float g) { return 1.0f / (1.0f + __expf(-g)); }

template <int FMT, bool ROWWISE, bool COLWISE, int PRE>
__global__ __launch_bounds__(256) void mxfp8_quant_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ aux16,
                                                          const float* __restrict__ aux32, uint8_t* __restrict__ y_row,
                                                          uint8_t* __restrict__ s_row, uint8_t* __restrict__ y_colT,
                                                          uint8_t* __restrict__ s_colT, float* __restrict__ colsum, int rows,
                                                          int cols, int tiles_c, int ldr, const uint16_t* __restrict__ e0 = nullptr,
                                                          const float* __restrict__ e1 = nullptr, int i0 = 0, int i1 = 0,
                                                          int i2 = 0) {
  // ldr: leading dimension of s_row / y_colT (= rows, or the row count of a larger operand this tensor is a row-block of)
  __shared__ float s_col[2][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
  const int r0 = tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int c0 = tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  const float rcp = 1.0f / fp8_max_of<FMT>();
  const bool active = (r0 < rows) && (c0 < cols);
  float f[8][8];  // values as cast; a NaN is replaced by 0 here and recorded in `nanmask` (bit 8 i + j): the block amax then
                  // needs no per-element test (fmaxf semantics: NaN ignored) and the byte is patched to 0x7F after the cast
  unsigned long long nanmask = 0;
  bool any_nan = false;
  if (PRE == 0) {
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
    any_nan = nan_seen(screen);
  } else if (PRE == 1) {
    float gm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
      const v4i gv = *reinterpret_cast<const v4i*>(aux16 + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gm[2 * j] = __uint_as_float((u32)gv[j] << 16);
        gm[2 * j + 1] = __uint_as_float((u32)gv[j] & 0xFFFF0000u);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v4i raw = {0, 0, 0, 0};
      float rs = 0.0f;
      if (active) {
        raw = *reinterpret_cast<const v4i*>(x + (int64_t)(r0 + i) * cols + c0);
        rs = aux32[r0 + i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32 w = (u32)raw[j];
        f[i][2 * j] = (__uint_as_float(w << 16) * rs) * gm[2 * j];
        f[i][2 * j + 1] = (__uint_as_float(w & 0xFFFF0000u) * rs) * gm[2 * j + 1];
      }
    }
    any_nan = true;  // computed values: take the explicit NaN test below
  } else if (PRE == 4) {
    // RoPE backward of the attention gradients, quantised as the fused d(qkv) [rows, W] they merge into (grad_output of the
    // q|k|v projection): x = dq [rows, nq*128], aux16 = dk, e0 = dv [rows, nk*128], aux32 / e1 = cos / sin [seq, 64] fp32,
    // i0 = nq, i1 = nk, i2 = seq; head_dim 128.  Same expressions and bf16 rounding as rope_qkv_kernel<1> (mi_fused.hip), so
    // the result equals mi_rope_qkv(backward) + mi_mxfp8_quantize bit for bit; the bf16 d(qkv) is never written.
    constexpr int HD = 128, HALF = 64;
    const int nq = i0, nk = i1, seq = i2;
    const int hd = c0 / HD, within = c0 % HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.0f;
      if (!active) continue;
      const int64_t r = r0 + i;
      if (hd < nq + nk) {
        const uint16_t* src = hd < nq ? x + (r * nq + hd) * HD : aux16 + (r * nk + (hd - nq)) * HD;
        const bool upper = within >= HALF;
        const int cc = upper ? within - HALF : within;
        const v4i a = *reinterpret_cast<const v4i*>(src + cc), b = *reinterpret_cast<const v4i*>(src + HALF + cc);
        const int pos = (int)(r % seq);
        float c[8], sn[8];
        *reinterpret_cast<v4f*>(c) = *reinterpret_cast<const v4f*>(aux32 + (int64_t)pos * HALF + cc);
        *reinterpret_cast<v4f*>(c + 4) = *reinterpret_cast<const v4f*>(aux32 + (int64_t)pos * HALF + cc + 4);
        *reinterpret_cast<v4f*>(sn) = *reinterpret_cast<const v4f*>(e1 + (int64_t)pos * HALF + cc);
        *reinterpret_cast<v4f*>(sn + 4) = *reinterpret_cast<const v4f*>(e1 + (int64_t)pos * HALF + cc + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 wa = (u32)a[j], wb = (u32)b[j];
          const float x1l = __uint_as_float(wa << 16), x1h = __uint_as_float(wa & 0xFFFF0000u);
          const float x2l = __uint_as_float(wb << 16), x2h = __uint_as_float(wb & 0xFFFF0000u);
          const float cl = c[2 * j], ch = c[2 * j + 1], sl = -1.0f * sn[2 * j], sh = -1.0f * sn[2 * j + 1];
          const float ol = upper ? x2l * cl + x1l * sl : x1l * cl - x2l * sl;
          const float oh = upper ? x2h * ch + x1h * sh : x1h * ch - x2h * sh;
          f[i][2 * j] = __uint_as_float(float_to_bf16_bits(ol) << 16);
          f[i][2 * j + 1] = __uint_as_float(float_to_bf16_bits(oh) << 16);
        }
      } else {
        const v4i a = *reinterpret_cast<const v4i*>(e0 + r * (int64_t)(nk * HD) + (c0 - (nq + nk) * HD));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[i][2 * j] = __uint_as_float((u32)a[j] << 16);
          f[i][2 * j + 1] = __uint_as_float((u32)a[j] & 0xFFFF0000u);
        }
      }
    }
    any_nan = true;
  } else {
    const int F = PRE == 2 ? cols : cols / 2;
    const bool up_half = (PRE == 3) && (c0 >= F);
    const int cg = up_half ? c0 - F : c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v4i gv = {0, 0, 0, 0}, uv = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
      if (active) {
        const int64_t r = r0 + i;
        gv = *reinterpret_cast<const v4i*>(x + r * 2 * F + cg);
        uv = *reinterpret_cast<const v4i*>(x + r * 2 * F + F + cg);
        if (PRE == 3) dv = *reinterpret_cast<const v4i*>(aux16 + r * F + cg);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 wg = (u32)gv[j], wu = (u32)uv[j], wd = (u32)dv[j];
        const float g2[2] = {__uint_as_float(wg << 16), __uint_as_float(wg & 0xFFFF0000u)};
        const float u2[2] = {__uint_as_float(wu << 16), __uint_as_float(wu & 0xFFFF0000u)};
        const float d2[2] = {__uint_as_float(wd << 16), __uint_as_float(wd & 0xFFFF0000u)};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float sg = mx_sigmoid(g2[e]);
          float v;
          if (PRE == 2) v = g2[e] * sg * u2[e];
          else if (!up_half) v = d2[e] * u2[e] * (sg * (1.0f + g2[e] * (1.0f - sg)));
          else v = d2[e] * (g2[e] * sg);
          f[i][2 * j + e] = v;
        }
      }
    }
    any_nan = true;
  }
  if ((PRE == 3 || PRE == 0) && colsum != nullptr) {  // column sums of the values being quantised (bias gradients)
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) v += f[i][j];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      csum[j] = v;
    }
    if ((lane >> 3) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_col[wave >> 1][(wave & 1) * 64 + (lane & 7) * 8 + j] = csum[j];
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tile_c * 128 + tid;
      if (c < cols) colsum[(int64_t)tile_r * cols + c] = s_col[0][tid] + s_col[1][tid];
    }
  }
  if (__builtin_expect(any_nan, PRE != 0)) {  // PRE == 0: only threads whose packed screen saw a NaN come here
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (f[i][j] != f[i][j]) {
          nanmask |= 1ull << (8 * i + j);
          f[i][j] = 0.0f;
        }
  }
  // bytes of a packed fp8 word (elements (i, j0..j0+3)) -> 0x7F where the mask says the source was a NaN (rare path)
  auto patch4 = [&](u32 w, int bit0) -> u32 {
    if (__builtin_expect(nanmask != 0, 0)) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((nanmask >> (bit0 + k)) & 1) w = (w & ~(0xFFu << (8 * k))) | (0x7Fu << (8 * k));
    }
    return w;
  };
  // NOTE: shuffles below are executed by every lane (inactive lanes carry zeros); rows/cols are
  // multiples of 32 so a 4-lane block group is all-active or all-inactive.
  if (ROWWISE) {
    u32 sbytes[8];
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(f[i][j]));
      a = fmaxf(a, __shfl_xor(a, 1));
      a = fmaxf(a, __shfl_xor(a, 2));
      u32 e = e8m0_roundup(a * rcp);
      float inv = e8m0_inv(e);
      sbytes[i] = e;
      lo[i] = patch4(cvt4_fp8<FMT>(f[i][0] * inv, f[i][1] * inv, f[i][2] * inv, f[i][3] * inv), 8 * i);
      hi[i] = patch4(cvt4_fp8<FMT>(f[i][4] * inv, f[i][5] * inv, f[i][6] * inv, f[i][7] * inv), 8 * i + 4);
    }
    if (active) {
      uint8_t* dst = y_row + (int64_t)r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * cols, lo[i], hi[i]);
      if ((lane & 3) == 0) {  // block-major scales [cols/32, rows]: this lane's 8 rows are 8 contiguous bytes
        const u32 lo4 = sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24);
        const u32 hi4 = sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24);
        *reinterpret_cast<uint2*>(s_row + (int64_t)(c0 / 32) * ldr + r0) = make_uint2(lo4, hi4);
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
      for (int i = 0; i < 8; ++i) a = fmaxf(a, fabsf(f[i][j]));
      a = fmaxf(a, __shfl_xor(a, 8));
      a = fmaxf(a, __shfl_xor(a, 16));
      u32 e = e8m0_roundup(a * rcp);
      sbytes[j] = e;
      inv[j] = e8m0_inv(e);
    }
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      lo[i] = patch4(cvt4_fp8<FMT>(f[i][0] * inv[0], f[i][1] * inv[1], f[i][2] * inv[2], f[i][3] * inv[3]), 8 * i);
      hi[i] = patch4(cvt4_fp8<FMT>(f[i][4] * inv[4], f[i][5] * inv[5], f[i][6] * inv[6], f[i][7] * inv[7]), 8 * i + 4);
    }
    if (active) {
      u32 a[4], b[4], c[4], d[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
      uint8_t* dst = y_colT + (int64_t)c0 * ldr + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mi::st8<MI_NT_YT>(dst + (int64_t)j * ldr, a[j], b[j]);
        mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * ldr, c[j], d[j]);
      }
      if (((lane >> 3) & 3) == 0) {  // block-major scales [rows/32, cols]: 8 contiguous bytes for this lane's 8 columns
        const u32 lo4 = sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24);
        const u32 hi4 = sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24);
        *reinterpret_cast<uint2*>(s_colT + (int64_t)(r0 / 32) * cols + c0) = make_uint2(lo4, hi4);
      }
    }
  }
}

template <int FMT, int PRE>
static int launch_mx(const void* x, const void* aux16, const float* aux32, void* y_row, void* s_row, void* y_colT, void* s_colT,
                     float* colsum, int64_t rows, int64_t cols, hipStream_t st, int64_t ld_rows = 0, const void* e0 = nullptr,
                     const float* e1 = nullptr, int i0 = 0, int i1 = 0, int i2 = 0) {
  const int ldr = (int)(ld_rows > 0 ? ld_rows : rows);
  const int tiles_r = (int)((rows + 127) / 128), tiles_c = (int)((cols + 127) / 128);
  dim3 grid((unsigned)(tiles_r * tiles_c)), block(256);
  const uint16_t *xp = (const uint16_t*)x, *ap = (const uint16_t*)aux16;
  uint8_t *yr = (uint8_t*)y_row, *sr = (uint8_t*)s_row, *yc = (uint8_t*)y_colT, *sc = (uint8_t*)s_colT;
  if (y_row && y_colT)
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, true, true, PRE>), grid, block, 0, st, xp, ap, aux32, yr, sr, yc, sc, colsum, (int)rows, (int)cols, tiles_c, ldr, (const uint16_t*)e0, e1, i0, i1, i2);
  else if (y_row)
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, true, false, PRE>), grid, block, 0, st, xp, ap, aux32, yr, sr, yc, sc, colsum, (int)rows, (int)cols, tiles_c, ldr, (const uint16_t*)e0, e1, i0, i1, i2);
  else
    hipLaunchKernelGGL((mxfp8_quant_kernel<FMT, false, true, PRE>), grid, block, 0, st, xp, ap, aux32, yr, sr, yc, sc, colsum, (int)rows, (int)cols, tiles_c, ldr, (const uint16_t*)e0, e1, i0, i1, i2);
  MI_CHECK_LAUNCH("mi_mxfp8_quantize launch");
  return MI_OK;
}

}  // namespace mi

static int mx_common_check(const char* who, const void* x, const void* y_row, const void* s_row, const void* y_colT,
                           const void* s_colT, int64_t rows, int64_t cols, int fmt) {
  MI_CHECK_ARG(x, "%s: null input", who);
  MI_CHECK_ARG((y_row && s_row) || (y_colT && s_colT), "%s: need (y_row,s_row) and/or (y_colT,s_colT)", who);
  MI_CHECK_ARG((!y_row) == (!s_row) && (!y_colT) == (!s_colT), "%s: data and scale pointers must pair up", who);
  MI_CHECK_ARG(rows >= 0 && cols >= 0 && rows % 32 == 0 && cols % 32 == 0,
               "%s: rows (%lld) and cols (%lld) must be multiples of 32", who, (long long)rows, (long long)cols);
  MI_CHECK_ARG(rows < (1LL << 31) && cols < (1LL << 31), "%s: shape too large", who);
  MI_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y_row % 8) == 0 && ((uintptr_t)y_colT % 8) == 0 &&
                   ((uintptr_t)s_row % 8) == 0 && ((uintptr_t)s_colT % 8) == 0, "%s: misaligned pointer", who);
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "%s: bad fmt %d", who, fmt);
  return MI_OK;
}

#define MI_MX_DISPATCH(PRE, x, a16, a32, cs, rows, cols)                                                                       \
  (fmt == MI_FMT_E4M3 ? mi::launch_mx<MI_FMT_E4M3, PRE>(x, a16, a32, y_row, s_row, y_colT, s_colT, cs, rows, cols, (hipStream_t)stream) \
                      : mi::launch_mx<MI_FMT_E5M2, PRE>(x, a16, a32, y_row, s_row, y_colT, s_colT, cs, rows, cols, (hipStream_t)stream))

extern "C" int mi_mxfp8_quantize(const void* x_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT,
                                 int64_t rows, int64_t cols, int fmt, void* stream) {
  int rc = mx_common_check("mi_mxfp8_quantize", x_bf16, y_row, s_row, y_colT, s_colT, rows, cols, fmt);
  if (rc != MI_OK) return rc;
  if (rows == 0 || cols == 0) return MI_OK;
  return MI_MX_DISPATCH(0, x_bf16, nullptr, nullptr, nullptr, rows, cols);
}

extern "C" int mi_mxfp8_quantize_ex(const void* x_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT, float* colsum,
                                    int64_t rows, int64_t cols, int64_t ld_rows, int fmt, void* stream) {
  int rc = mx_common_check("mi_mxfp8_quantize_ex", x_bf16, y_row, s_row, y_colT, s_colT, rows, cols, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(ld_rows >= rows && ld_rows % 8 == 0 && ld_rows < (1LL << 31), "mi_mxfp8_quantize_ex: ld_rows (%lld) must be >= rows and a multiple of 8",
               (long long)ld_rows);
  if (rows == 0 || cols == 0) return MI_OK;
  return (fmt == MI_FMT_E4M3
              ? mi::launch_mx<MI_FMT_E4M3, 0>(x_bf16, nullptr, nullptr, y_row, s_row, y_colT, s_colT, colsum, rows, cols, (hipStream_t)stream, ld_rows)
              : mi::launch_mx<MI_FMT_E5M2, 0>(x_bf16, nullptr, nullptr, y_row, s_row, y_colT, s_colT, colsum, rows, cols, (hipStream_t)stream, ld_rows));
}

extern "C" int mi_mxfp8_norm_quantize(const void* x_bf16, const float* rstd, const void* gamma_bf16, void* y_row, void* s_row,
                                      void* y_colT, void* s_colT, int64_t rows, int64_t cols, int fmt, void* stream) {
  int rc = mx_common_check("mi_mxfp8_norm_quantize", x_bf16, y_row, s_row, y_colT, s_colT, rows, cols, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(rstd && gamma_bf16 && ((uintptr_t)gamma_bf16 % 16) == 0, "mi_mxfp8_norm_quantize: rstd / gamma missing or misaligned");
  if (rows == 0 || cols == 0) return MI_OK;
  return MI_MX_DISPATCH(1, x_bf16, gamma_bf16, rstd, nullptr, rows, cols);
}

extern "C" int mi_mxfp8_swiglu_quantize(const void* h_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT, int64_t rows,
                                        int64_t F, int fmt, void* stream) {
  int rc = mx_common_check("mi_mxfp8_swiglu_quantize", h_bf16, y_row, s_row, y_colT, s_colT, rows, F, fmt);
  if (rc != MI_OK) return rc;
  if (rows == 0 || F == 0) return MI_OK;
  return MI_MX_DISPATCH(2, h_bf16, nullptr, nullptr, nullptr, rows, F);
}

extern "C" int mi_mxfp8_dswiglu_quantize(const void* h_bf16, const void* dact_bf16, void* y_row, void* s_row, void* y_colT,
                                         void* s_colT, float* colsum, int64_t rows, int64_t F, int fmt, void* stream) {
  int rc = mx_common_check("mi_mxfp8_dswiglu_quantize", h_bf16, y_row, s_row, y_colT, s_colT, rows, 2 * F, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(dact_bf16 && ((uintptr_t)dact_bf16 % 16) == 0, "mi_mxfp8_dswiglu_quantize: dact missing or misaligned");
  if (rows == 0 || F == 0) return MI_OK;
  return MI_MX_DISPATCH(3, h_bf16, dact_bf16, nullptr, colsum, rows, 2 * F);
}

extern "C" int mi_mxfp8_rope_bwd_quantize(const void* dq_bf16, const void* dk_bf16, const void* dv_bf16, const float* cos_tab,
                                          const float* sin_tab, void* y_row, void* s_row, void* y_colT, void* s_colT, int64_t rows,
                                          int64_t seq, int n_q_heads, int n_kv_heads, int head_dim, int fmt, void* stream) {
  MI_CHECK_ARG(head_dim == 128, "mi_mxfp8_rope_bwd_quantize: head_dim %d not supported (128)", head_dim);
  MI_CHECK_ARG(n_q_heads >= 1 && n_kv_heads >= 1 && seq >= 1, "mi_mxfp8_rope_bwd_quantize: bad head counts / seq");
  const int64_t W = (int64_t)(n_q_heads + 2 * n_kv_heads) * head_dim;
  int rc = mx_common_check("mi_mxfp8_rope_bwd_quantize", dq_bf16, y_row, s_row, y_colT, s_colT, rows, W, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(dk_bf16 && dv_bf16 && cos_tab && sin_tab && ((uintptr_t)dk_bf16 % 16) == 0 && ((uintptr_t)dv_bf16 % 16) == 0 &&
                   ((uintptr_t)cos_tab % 16) == 0 && ((uintptr_t)sin_tab % 16) == 0, "mi_mxfp8_rope_bwd_quantize: null or misaligned input");
  if (rows == 0) return MI_OK;
  return (fmt == MI_FMT_E4M3
              ? mi::launch_mx<MI_FMT_E4M3, 4>(dq_bf16, dk_bf16, cos_tab, y_row, s_row, y_colT, s_colT, nullptr, rows, W, (hipStream_t)stream,
                                              0, dv_bf16, sin_tab, n_q_heads, n_kv_heads, (int)seq)
              : mi::launch_mx<MI_FMT_E5M2, 4>(dq_bf16, dk_bf16, cos_tab, y_row, s_row, y_colT, s_colT, nullptr, rows, W, (hipStream_t)stream,
                                              0, dv_bf16, sin_tab, n_q_heads, n_kv_heads, (int)seq));
}
