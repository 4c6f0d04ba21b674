// HBM-bound fusions on either side of the FP8 GEMMs (SURVEY.md 8f "next" rows):
//   mi_rope_qkv         fwd: split the fused QKV GEMM output into q, k, v and rotate q, k (RoPE) in the same pass;
//                       bwd: merge dq, dk, dv back into the fused gradient with the conjugate rotation --
//                       one launch each instead of the split/mul/neg/cat/add chain (TE uses a fused RoPE
//                       kernel at this point of the reference path, te_llama.py:77).
//   mi_swiglu_cast      K10 fwd: act = silu(h[:, :F]) * h[:, F:] in fp32 -> FP8 (+ transposed copy) + amax,
//                       without materialising the bf16 activation (te_llama.py:62 activation="swiglu").
//   mi_rmsnorm_stats / mi_norm_cast / mi_rmsnorm_bwd   K9: RMSNorm fused with the FP8 cast of the GEMM input
//                       (the normalised bf16 activation is never materialised) and its backward.
//   mi_dswiglu_cast     K10 bwd: dh = [dact * dsilu(g) * u | dact * silu(g)] in fp32 -> FP8 (+T) + amax,
//                       plus deterministic per-row-block column sums for the fc1 bias gradient.
#include "mi_common.h"

namespace mi {

// ------------------------------------------------------------------------------------------------ RoPE
// fused: [rows, W] bf16 with columns q (nq heads) | k (nk heads) | v (nk heads), head size D.
// DIR 0 (forward):  q, k, v (separate contiguous [rows, heads*D]) <- fused, q/k rotated by +theta(pos), v copied
// DIR 1 (backward): fused gradient <- dq, dk, dv, q/k parts rotated by -theta(pos)
// Row r has position (r % seq) (bshd).  cos/sin: [>= seq, D/2] fp32.
// out1 = x1*c - x2*s, out2 = x2*c + x1*s with x1/x2 the two halves of a head (TE's non-interleaved RoPE).
template <int DIR>
__global__ __launch_bounds__(256) void rope_qkv_kernel(uint16_t* __restrict__ fused, uint16_t* __restrict__ q,
                                                       uint16_t* __restrict__ k, uint16_t* __restrict__ v,
                                                       const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                       int rows, int seq, int nq, int nk, int D) {
  const int half = D >> 1;
  const int vph = half >> 3;  // 8-element vector pairs per head
  const int W = (nq + 2 * nk) * D;
  const int rot_items = (nq + nk) * vph;    // per row
  const int cpy_items = (nk * D) >> 3;      // per row
  const int per_row = rot_items + cpy_items;
  const int64_t items = (int64_t)rows * per_row;
  const float sgn = DIR == 0 ? 1.0f : -1.0f;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = it / per_row;
    const int w = (int)(it - r * per_row);
    uint16_t* frow = fused + r * W;
    if (w < rot_items) {
      const int hd = w / vph, vv = w - hd * vph;
      uint16_t* sep = hd < nq ? q + (r * nq + hd) * D : k + (r * nk + (hd - nq)) * D;
      uint16_t* f1 = frow + hd * D + vv * 8;
      uint16_t* s1 = sep + vv * 8;
      const uint16_t* in1 = DIR == 0 ? f1 : s1;
      uint16_t* out1 = DIR == 0 ? s1 : f1;
      const v4i a = *reinterpret_cast<const v4i*>(in1), b = *reinterpret_cast<const v4i*>(in1 + half);
      const int pos = (int)(r % seq);
      const float* cp = cosT + (int64_t)pos * half + vv * 8;
      const float* sp = sinT + (int64_t)pos * half + vv * 8;
      float c[8], sn[8];
      *reinterpret_cast<v4f*>(c) = *reinterpret_cast<const v4f*>(cp);
      *reinterpret_cast<v4f*>(c + 4) = *reinterpret_cast<const v4f*>(cp + 4);
      *reinterpret_cast<v4f*>(sn) = *reinterpret_cast<const v4f*>(sp);
      *reinterpret_cast<v4f*>(sn + 4) = *reinterpret_cast<const v4f*>(sp + 4);
      v4i o1, o2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 wa = (u32)a[j], wb = (u32)b[j];
        const float x1l = __uint_as_float(wa << 16), x1h = __uint_as_float(wa & 0xFFFF0000u);
        const float x2l = __uint_as_float(wb << 16), x2h = __uint_as_float(wb & 0xFFFF0000u);
        const float cl = c[2 * j], ch = c[2 * j + 1], sl = sgn * sn[2 * j], sh = sgn * sn[2 * j + 1];
        o1[j] = (int)pack_bf16x2(x1l * cl - x2l * sl, x1h * ch - x2h * sh);
        o2[j] = (int)pack_bf16x2(x2l * cl + x1l * sl, x2h * ch + x1h * sh);
      }
      *reinterpret_cast<v4i*>(out1) = o1;
      *reinterpret_cast<v4i*>(out1 + half) = o2;
    } else {
      const int cv = w - rot_items;
      uint16_t* fp = frow + (nq + nk) * D + cv * 8;
      uint16_t* sp2 = v + r * (int64_t)(nk * D) + cv * 8;
      if (DIR == 0) *reinterpret_cast<v4i*>(sp2) = *reinterpret_cast<const v4i*>(fp);
      else *reinterpret_cast<v4i*>(fp) = *reinterpret_cast<const v4i*>(sp2);
    }
  }
}

// RoPE backward fused with the FP8 cast of the result: the fused gradient d(qkv) [rows, W] is grad_output of the q|k|v
// projection, whose backward would read it once more just to quantise it.  Here it leaves as FP8 (y [rows, W] and
// yT [W, rows]) + amax and is never written in bf16.  Values are rounded to bf16 before the cast, so the bytes equal
// mi_rope_qkv(backward) followed by mi_cast_amax bit for bit.  head_dim 128: a wave owns 64 rows x one whole q/k head
// (the two 64-column halves x1 | x2 that rotate into each other) or 64 rows x 64 columns of the v part.
template <int FMT, bool WRITE_Y, bool WRITE_T>
__global__ __launch_bounds__(256) void rope_bwd_cast_kernel(const uint16_t* __restrict__ dq, const uint16_t* __restrict__ dk,
                                                            const uint16_t* __restrict__ dv, const float* __restrict__ cosT,
                                                            const float* __restrict__ sinT, uint8_t* __restrict__ y,
                                                            uint8_t* __restrict__ yT, const float* __restrict__ scale_p,
                                                            float* amax_out, int rows, int seq, int nq, int nk) {
  constexpr int D = 128, half = 64;
  __shared__ float s_amax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = (nq + 2 * nk) * D;
  const int units = (nq + nk) + 2 * nk;  // per 64-row block: rotated heads, then 64-column tiles of v
  const int64_t u_all = (int64_t)blockIdx.x * 4 + wave;
  const int rb = (int)(u_all / units), u = (int)(u_all % units);
  const int r0 = rb * 64 + (lane >> 3) * 8, lc = (lane & 7) * 8;
  const float scale = *scale_p;
  float amax = 0.0f;
  auto emit = [&](const float (&f)[8][8], int c0) {  // 8 x 8 block (rows r0.., fused columns c0..) -> FP8 both ways
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, (f[i][j] != f[i][j]) ? 0.0f : fabsf(f[i][j]));
      lo[i] = cvt4_fp8<FMT>(f[i][0] * scale, f[i][1] * scale, f[i][2] * scale, f[i][3] * scale);
      hi[i] = cvt4_fp8<FMT>(f[i][4] * scale, f[i][5] * scale, f[i][6] * scale, f[i][7] * scale);
    }
    if (WRITE_Y) {
      uint8_t* dst = y + (int64_t)r0 * W + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * W, lo[i], hi[i]);
    }
    if (WRITE_T) {
      u32 a[4], b[4], c[4], d[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
      uint8_t* dst = yT + (int64_t)c0 * rows + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mi::st8<MI_NT_YT>(dst + (int64_t)j * rows, a[j], b[j]);
        mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * rows, c[j], d[j]);
      }
    }
  };
  auto bf16_round = [](float v) { return __uint_as_float(float_to_bf16_bits(v) << 16); };
  if (r0 < rows) {  // rows is a multiple of 8: a lane's 8 rows are all in or all out
    if (u < nq + nk) {
      const int hd = u;
      const uint16_t* src = hd < nq ? dq + ((int64_t)r0 * nq + hd) * D : dk + ((int64_t)r0 * nk + (hd - nq)) * D;
      const int64_t rstride = (int64_t)(hd < nq ? nq : nk) * D;
      float f1[8][8], f2[8][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v4i a = *reinterpret_cast<const v4i*>(src + i * rstride + lc);
        const v4i b = *reinterpret_cast<const v4i*>(src + i * rstride + half + lc);
        const int pos = (r0 + i) % seq;
        float c[8], sn[8];
        *reinterpret_cast<v4f*>(c) = *reinterpret_cast<const v4f*>(cosT + (int64_t)pos * half + lc);
        *reinterpret_cast<v4f*>(c + 4) = *reinterpret_cast<const v4f*>(cosT + (int64_t)pos * half + lc + 4);
        *reinterpret_cast<v4f*>(sn) = *reinterpret_cast<const v4f*>(sinT + (int64_t)pos * half + lc);
        *reinterpret_cast<v4f*>(sn + 4) = *reinterpret_cast<const v4f*>(sinT + (int64_t)pos * half + lc + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 wa = (u32)a[j], wb = (u32)b[j];
          const float x1l = __uint_as_float(wa << 16), x1h = __uint_as_float(wa & 0xFFFF0000u);
          const float x2l = __uint_as_float(wb << 16), x2h = __uint_as_float(wb & 0xFFFF0000u);
          // same expressions as rope_qkv_kernel<1> (sgn = -1), then its bf16 rounding
          const float cl = c[2 * j], ch = c[2 * j + 1], sl = -1.0f * sn[2 * j], sh = -1.0f * sn[2 * j + 1];
          f1[i][2 * j] = bf16_round(x1l * cl - x2l * sl);
          f1[i][2 * j + 1] = bf16_round(x1h * ch - x2h * sh);
          f2[i][2 * j] = bf16_round(x2l * cl + x1l * sl);
          f2[i][2 * j + 1] = bf16_round(x2h * ch + x1h * sh);
        }
      }
      emit(f1, hd * D + lc);
      emit(f2, hd * D + half + lc);
    } else {
      const int tc = u - (nq + nk);  // 64-column tile of the v part
      float f[8][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v4i a = *reinterpret_cast<const v4i*>(dv + (int64_t)(r0 + i) * (nk * D) + tc * 64 + lc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[i][2 * j] = __uint_as_float((u32)a[j] << 16);
          f[i][2 * j + 1] = __uint_as_float((u32)a[j] & 0xFFFF0000u);
        }
      }
      emit(f, (nq + nk) * D + tc * 64 + lc);
    }
  }
  if (amax_out != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      const float m = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
      // only a workgroup that would raise the value goes to the atomic unit: same-address atomics are served one at a time and a
      // wave cannot retire before its atomic has returned (thousands of workgroups per launch)
      if (m > 0.0f && m > __builtin_nontemporal_load(amax_out)) atomicMax(reinterpret_cast<unsigned int*>(amax_out), __float_as_uint(m));
    }
  }
}

// ------------------------------------------------------------------------------------------------ SwiGLU (+cast)
__device__ __forceinline__ float sigmoidf_(float g) { return 1.0f / (1.0f + __expf(-g)); }

// Same tiling as the one-shot cast: a 128x128 tile of the [rows, F] GATE space per workgroup, 8x8 block per lane.
// MODE 0 (fwd):  in = h [rows, 2F];            val(r,c) = silu(h[r,c]) * h[r,F+c],   c in [0,F)  -> out [rows, F]
// MODE 1 (bwd):  in = h [rows, 2F], d [rows,F]; val(r,c) = d[r,c]*dsilu(g)*u (c < F) | d[r,c-F]*silu(g) (c >= F) -> out [rows, 2F]
//               BOTH output blocks (columns c and F + c) come from ONE load of g, u, d (as two separate tiles each half
//               re-read all three inputs: 16 instead of 10 bytes per gate element).  The 8 rows of a lane's block are
//               handled in two passes of 4 (pinned with sched_barrier): 151 us instead of 170 us per 8192x16384 (forcing
//               128 VGPRs for a 4th wave per SIMD spills and doubles the time).
//               colsum[(tile_r), c] = sum over the tile's 128 rows of val (fp32): fc1 bias gradient.
//               `bias` (optional, bf16 [2F]): h holds the fc1 GEMM's output WITHOUT its bias and the bias is added here, in fp32,
//               to the unpacked gate / up values (TE's bias + activation fusion: the add rides in an HBM-bound kernel whose VALU
//               is idle instead of in the GEMM epilogue, where it costs 12 % of the K = 3072 GEMM).
template <int FMT, int MODE, bool WRITE_Y, bool WRITE_T>
__global__ __launch_bounds__(256) void swiglu_cast_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ d,
                                                          uint8_t* __restrict__ y, uint8_t* __restrict__ yT,
                                                          const float* __restrict__ scale_p, float* amax_out,
                                                          float* __restrict__ colsum, int rows, int F, int tiles_c,
                                                          const uint16_t* __restrict__ bias) {
  __shared__ float s_amax[4];
  __shared__ float s_col[MODE == 1 ? 2 : 1][2][128];
  constexpr int NOUT = MODE == 0 ? 1 : 2;
  const int ocols = MODE == 0 ? F : 2 * F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
  const int r0 = tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int cg = tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;  // column inside the gate / d tensors
  const float scale = *scale_p;
  float amax = 0.0f;
  float csum[NOUT][8];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[o][j] = 0.0f;
  const bool active = (r0 < rows) && (cg < F);
  if (active) {
    u32 lo[NOUT][8], hi[NOUT][8];
    float bg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bu[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
      const v4i vg = *reinterpret_cast<const v4i*>(bias + cg), vu = *reinterpret_cast<const v4i*>(bias + F + cg);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bg[2 * j] = __uint_as_float((u32)vg[j] << 16);
        bg[2 * j + 1] = __uint_as_float((u32)vg[j] & 0xFFFF0000u);
        bu[2 * j] = __uint_as_float((u32)vu[j] << 16);
        bu[2 * j + 1] = __uint_as_float((u32)vu[j] & 0xFFFF0000u);
      }
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      v4i gv[4], uv[4], dv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + 4 * pass + i;
        gv[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(h + r * 2 * F + cg));
        uv[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(h + r * 2 * F + F + cg));
        dv[i] = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const v4i*>(d + r * F + cg)) : (v4i){0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[NOUT][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 wg = (u32)gv[i][j], wu = (u32)uv[i][j], wd = (u32)dv[i][j];
          float g2[2] = {__uint_as_float(wg << 16), __uint_as_float(wg & 0xFFFF0000u)};
          float u2[2] = {__uint_as_float(wu << 16), __uint_as_float(wu & 0xFFFF0000u)};
          const float d2[2] = {__uint_as_float(wd << 16), __uint_as_float(wd & 0xFFFF0000u)};
          if (bias != nullptr) {
            g2[0] += bg[2 * j];
            g2[1] += bg[2 * j + 1];
            u2[0] += bu[2 * j];
            u2[1] += bu[2 * j + 1];
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float sg = sigmoidf_(g2[e]);
            if (MODE == 0) {
              f[0][2 * j + e] = g2[e] * sg * u2[e];
            } else {
              f[0][2 * j + e] = d2[e] * u2[e] * (sg * (1.0f + g2[e] * (1.0f - sg)));
              f[NOUT - 1][2 * j + e] = d2[e] * (g2[e] * sg);
            }
          }
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            amax = fmaxf(amax, (f[o][j] != f[o][j]) ? 0.0f : fabsf(f[o][j]));
            if (MODE == 1) csum[o][j] += f[o][j];
          }
          lo[o][4 * pass + i] = cvt4_fp8<FMT>(f[o][0] * scale, f[o][1] * scale, f[o][2] * scale, f[o][3] * scale);
          hi[o][4 * pass + i] = cvt4_fp8<FMT>(f[o][4] * scale, f[o][5] * scale, f[o][6] * scale, f[o][7] * scale);
        }
      }
      if (MODE == 1) __builtin_amdgcn_sched_barrier(0);  // keep the second pass's loads behind the first pass's arithmetic
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int c0 = cg + o * F;
      if (WRITE_Y) {
        uint8_t* dst = y + (int64_t)r0 * ocols + c0;
#pragma unroll
        for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * ocols, lo[o][i], hi[o][i]);
      }
      if (WRITE_T) {
        u32 a[4], b[4], c[4], dd[4];
        transpose4x4(lo[o][0], lo[o][1], lo[o][2], lo[o][3], a[0], a[1], a[2], a[3]);
        transpose4x4(lo[o][4], lo[o][5], lo[o][6], lo[o][7], b[0], b[1], b[2], b[3]);
        transpose4x4(hi[o][0], hi[o][1], hi[o][2], hi[o][3], c[0], c[1], c[2], c[3]);
        transpose4x4(hi[o][4], hi[o][5], hi[o][6], hi[o][7], dd[0], dd[1], dd[2], dd[3]);
        uint8_t* dst = yT + (int64_t)c0 * rows + r0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mi::st8<MI_NT_YT>(dst + (int64_t)j * rows, a[j], b[j]);
          mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * rows, c[j], dd[j]);
        }
      }
    }
  }
  if (MODE == 1 && colsum != nullptr) {
    // reduce the 8 row-blocks of a wave (lanes differing in lane>>3), then the 2 wave-rows through LDS:
    // a fixed order, so the bias gradient is bitwise reproducible
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = csum[o][j];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if ((lane >> 3) == 0) s_col[o][wave >> 1][(wave & 1) * 64 + (lane & 7) * 8 + j] = v;
      }
    __syncthreads();
    const int c = tile_c * 128 + (tid & 127), o = tid >> 7;
    if (c < F) colsum[(int64_t)tile_r * ocols + o * F + c] = s_col[MODE == 1 ? o : 0][0][tid & 127] + s_col[MODE == 1 ? o : 0][1][tid & 127];
  }
  if (amax_out != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      float m = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
      // only a workgroup that would raise the value goes to the atomic unit: same-address atomics are served one at a time and a
      // wave cannot retire before its atomic has returned (thousands of workgroups per launch)
      if (m > 0.0f && m > __builtin_nontemporal_load(amax_out)) atomicMax(reinterpret_cast<unsigned int*>(amax_out), __float_as_uint(m));
    }
  }
}

template <int FMT, int MODE>
static int launch_swiglu(const void* h, const void* d, void* y, void* yT, const float* scale, float* amax, float* colsum,
                         int64_t rows, int64_t F, hipStream_t st, const void* bias = nullptr) {
  const int64_t ocols = MODE == 0 ? F : 2 * F;
  (void)ocols;
  const int tiles_r = (int)((rows + 127) / 128), tiles_c = (int)((F + 127) / 128);  // tiles of the [rows, F] gate space
  dim3 grid((unsigned)(tiles_r * tiles_c)), block(256);
  const uint16_t *hp = (const uint16_t*)h, *dp = (const uint16_t*)d, *bp = (const uint16_t*)bias;
  uint8_t *yp = (uint8_t*)y, *tp = (uint8_t*)yT;
  if (y && yT)
    hipLaunchKernelGGL((swiglu_cast_kernel<FMT, MODE, true, true>), grid, block, 0, st, hp, dp, yp, tp, scale, amax, colsum, (int)rows, (int)F, tiles_c, bp);
  else if (y)
    hipLaunchKernelGGL((swiglu_cast_kernel<FMT, MODE, true, false>), grid, block, 0, st, hp, dp, yp, tp, scale, amax, colsum, (int)rows, (int)F, tiles_c, bp);
  else
    hipLaunchKernelGGL((swiglu_cast_kernel<FMT, MODE, false, true>), grid, block, 0, st, hp, dp, yp, tp, scale, amax, colsum, (int)rows, (int)F, tiles_c, bp);
  MI_CHECK_LAUNCH("mi_swiglu_cast launch");
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ K9: RMSNorm -> FP8
// rstd[r] = rsqrt(mean_c x[r,c]^2 + eps).  One wave per row, 16-B loads, fp32 accumulation in a fixed order.
__global__ __launch_bounds__(256) void rmsnorm_stats_kernel(const uint16_t* __restrict__ x, float* __restrict__ rstd, int rows,
                                                            int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint16_t* xr = x + (int64_t)row * cols;
  float acc = 0.0f;
  for (int c = lane * 8; c < cols; c += 512) {
    const v4i v = *reinterpret_cast<const v4i*>(xr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 w = (u32)v[j];
      const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xFFFF0000u);
      acc += a * a + b * b;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) rstd[row] = rsqrtf(acc / (float)cols + eps);
}

// y = (x * rstd[row]) * gamma[col] in fp32 -> FP8 (+ transposed copy) + amax: the normalised activation is never
// written in bf16 (TE LayerNormLinear / LayerNormMLP do the same on the reference path, te_llama.py:45-63).
// Persistent form (round 3), the walk of cast_amax_kernel (mi_cast.hip): every WAVE walks 64 x 64-element tiles (an 8 x 8 block per
// lane) with stride = waves in the grid, the 8 row loads, the 8 rstd values and the gamma slice of its NEXT tile already in flight
// while the current tile is normalised, converted and stored; two workgroups per CU (26.5 -> 24 us per 8192 x 3072 in the step's
// conditions: profiles/r03_cast_grid_sweep.txt has the grid sweep of the sibling kernel).
template <int FMT, bool WRITE_Y, bool WRITE_T>
__global__ __launch_bounds__(256) void norm_cast_kernel(const uint16_t* __restrict__ x, const float* __restrict__ rstd,
                                                        const uint16_t* __restrict__ gamma, uint8_t* __restrict__ y,
                                                        uint8_t* __restrict__ yT, const float* __restrict__ scale_p,
                                                        float* amax_out, int rows, int cols, int tiles_c) {
  __shared__ float s_amax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = ((rows + 63) / 64) * tiles_c;  // tiles_c: 64-column tiles
  const int stride = gridDim.x * 4;
  const int lr = (lane >> 3) * 8, lc = (lane & 7) * 8;
  const float scale = *scale_p;
  float amax = 0.0f;
  int t = blockIdx.x * 4 + wave;
  v4i nxt[8], nxt_g;
  v4f nxt_r[2];
  auto load_tile = [&](int tt, v4i (&raw)[8], v4i& gv, v4f (&rs)[2]) {
    const int r0 = (tt / tiles_c) * 64 + lr, c0 = (tt % tiles_c) * 64 + lc;
    if (r0 < rows && c0 < cols) {  // dims are multiples of 8: blocks are all-in or all-out
      const uint16_t* src = x + (int64_t)r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src + (int64_t)i * cols));
      gv = *reinterpret_cast<const v4i*>(gamma + c0);
      rs[0] = *reinterpret_cast<const v4f*>(rstd + r0);
      rs[1] = *reinterpret_cast<const v4f*>(rstd + r0 + 4);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = (v4i){0, 0, 0, 0};
      gv = (v4i){0, 0, 0, 0};
      rs[0] = rs[1] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
  };
  if (t < ntiles) load_tile(t, nxt, nxt_g, nxt_r);
  while (t < ntiles) {
    v4i raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = nxt[i];
    const v4i gv = nxt_g;
    const v4f rs0 = nxt_r[0], rs1 = nxt_r[1];
    const int tn = t + stride;
    if (tn < ntiles) load_tile(tn, nxt, nxt_g, nxt_r);
    const int r0 = (t / tiles_c) * 64 + lr, c0 = (t % tiles_c) * 64 + lc;
    if ((r0 < rows) && (c0 < cols)) {
      float g[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[2 * j] = __uint_as_float((u32)gv[j] << 16);
        g[2 * j + 1] = __uint_as_float((u32)gv[j] & 0xFFFF0000u);
      }
      u32 lo[8], hi[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float rs = i < 4 ? rs0[i] : rs1[i - 4];
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 w = (u32)raw[i][j];
          f[2 * j] = (__uint_as_float(w << 16) * rs) * g[2 * j];
          f[2 * j + 1] = (__uint_as_float(w & 0xFFFF0000u) * rs) * g[2 * j + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, (f[j] != f[j]) ? 0.0f : fabsf(f[j]));
        lo[i] = cvt4_fp8<FMT>(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
        hi[i] = cvt4_fp8<FMT>(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
      }
      if (WRITE_Y) {
        uint8_t* dst = y + (int64_t)r0 * cols + c0;
#pragma unroll
        for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * cols, lo[i], hi[i]);
      }
      if (WRITE_T) {
        u32 a[4], b[4], c[4], d[4];
        transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
        transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
        transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
        transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
        uint8_t* dst = yT + (int64_t)c0 * rows + r0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mi::st8<MI_NT_YT>(dst + (int64_t)j * rows, a[j], b[j]);
          mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * rows, c[j], d[j]);
        }
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

// RMSNorm backward.  dy = grad w.r.t. the normalised output (the dgrad GEMM's bf16 result), xhat = x * rstd:
//   dx[r,c]  = rstd[r] * (dy*g - xhat * mean_c(dy*g*xhat))          (+ dres[r,c] if a residual gradient is given)
//   dgp[b,c] = sum over block b's rows of dy * xhat                  (fp32 partials, fixed order; caller adds the blocks)
// WPR waves share a row (each keeps its cols / WPR slice of the row in registers between the two passes and exchanges its part
// of the row's dot product through LDS: one barrier per row group); 4 / WPR rows are in flight per block.  With one wave per
// row, hidden 3072 needs 236 VGPRs (2 waves per SIMD) and hidden 4096 spills into AGPRs at 1 wave per SIMD -- too little
// memory-level parallelism for a streaming kernel; two waves per row halve gamma, the dgamma accumulators and the row copy.
constexpr int kNormMaxVec = 16;  // 16 x 8 x 64 = 8192 columns
template <int NVEC, int WPR>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                          const float* __restrict__ rstd, const uint16_t* __restrict__ gamma,
                                                          const uint16_t* __restrict__ dres, uint16_t* __restrict__ dx,
                                                          float* __restrict__ dgp, int rows, int rows_per_block) {
  static_assert(NVEC % WPR == 0 && (WPR == 1 || WPR == 2 || WPR == 4), "row split");
  extern __shared__ float s_dg[];  // [4 / WPR][cols]
  __shared__ float s_dot[2][4];    // [parity][wave]: per-wave parts of the dot products of the rows in flight
  constexpr int cols = NVEC * 512;  // 512 columns per wave-wide vector step
  constexpr int NV = NVEC / WPR, R = 4 / WPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave / WPR, part = wave % WPR;
  const int col0 = part * NV * 512 + lane * 8;
  v4i gq[NV];          // gamma, packed bf16
  float dgacc[NV][8];  // this wave's share of dgamma
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gq[v] = *reinterpret_cast<const v4i*>(gamma + col0 + v * 512);
#pragma unroll
    for (int j = 0; j < 8; ++j) dgacc[v][j] = 0.0f;
  }
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(rows, row_begin + rows_per_block);
  const int iters = (row_end - row_begin + R - 1) / R;  // the same for every wave of the block: the loop holds a barrier
  for (int it = 0; it < iters; ++it) {
    const int row = row_begin + it * R + slot;
    const bool valid = row < row_end;
    const float rs = valid ? rstd[row] : 0.0f;
    v4i dq[NV], xq[NV];
    float dot = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      dq[v] = (v4i){0, 0, 0, 0};
      xq[v] = (v4i){0, 0, 0, 0};
      if (valid) {
        const int64_t off = (int64_t)row * cols + col0 + v * 512;
        dq[v] = *reinterpret_cast<const v4i*>(dy + off);
        xq[v] = *reinterpret_cast<const v4i*>(x + off);
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = __uint_as_float((u32)dq[v][j] << 16), d1 = __uint_as_float((u32)dq[v][j] & 0xFFFF0000u);
        const float x0 = __uint_as_float((u32)xq[v][j] << 16) * rs, x1 = __uint_as_float((u32)xq[v][j] & 0xFFFF0000u) * rs;
        const float g0 = __uint_as_float((u32)gq[v][j] << 16), g1 = __uint_as_float((u32)gq[v][j] & 0xFFFF0000u);
        dgacc[v][2 * j] += d0 * x0;
        dgacc[v][2 * j + 1] += d1 * x1;
        dot += (d0 * g0) * x0 + (d1 * g1) * x1;
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
    if (WPR > 1) {
      if (lane == 0) s_dot[it & 1][wave] = dot;
      __syncthreads();
      dot = s_dot[it & 1][slot * WPR];
#pragma unroll
      for (int q = 1; q < WPR; ++q) dot += s_dot[it & 1][slot * WPR + q];  // fixed order: every wave of the row gets the same sum
    }
    const float c = dot / (float)cols;
    if (valid) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int64_t off = (int64_t)row * cols + col0 + v * 512;
        v4i rv = {0, 0, 0, 0};
        if (dres) rv = *reinterpret_cast<const v4i*>(dres + off);
        v4i ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d0 = __uint_as_float((u32)dq[v][j] << 16), d1 = __uint_as_float((u32)dq[v][j] & 0xFFFF0000u);
          const float x0 = __uint_as_float((u32)xq[v][j] << 16) * rs, x1 = __uint_as_float((u32)xq[v][j] & 0xFFFF0000u) * rs;
          const float g0 = __uint_as_float((u32)gq[v][j] << 16), g1 = __uint_as_float((u32)gq[v][j] & 0xFFFF0000u);
          float o0 = rs * (d0 * g0 - x0 * c), o1 = rs * (d1 * g1 - x1 * c);
          if (dres) {
            o0 += __uint_as_float((u32)rv[j] << 16);
            o1 += __uint_as_float((u32)rv[j] & 0xFFFF0000u);
          }
          ov[j] = (int)pack_bf16x2(o0, o1);
        }
        *reinterpret_cast<v4i*>(dx + off) = ov;
      }
    }
  }
  // block-level dgamma partial: the R row slots -> LDS -> one row of dgp (fixed order)
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) s_dg[slot * cols + col0 + v * 512 + j] = dgacc[v][j];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    float a = s_dg[c];
#pragma unroll
    for (int q = 1; q < R; ++q) a += s_dg[q * cols + c];
    dgp[(int64_t)blockIdx.x * cols + c] = a;
  }
}

// ------------------------------------------------------------------------------------------------ cross-entropy
// The loss the reference trains with (HF ForCausalLMLoss: mean token cross-entropy, ignore_index = -100) on the bf16
// logits of the FP8 lm_head, without the fp32 copy of the [tokens, vocab] logits: one workgroup per row.
//   forward : lse[r] = log sum_c exp(x[r,c]) (online max/sum, fp32), loss[r] = lse[r] - x[r,label] (0 if ignored)
//   backward: dx[r,c] = (exp(x[r,c] - lse[r]) - [c == label]) * *gscale   (0 for ignored rows)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const uint16_t* __restrict__ x, const int64_t* __restrict__ labels,
                                                     float* __restrict__ lse, float* __restrict__ loss, int cols) {
  __shared__ float s_m[4], s_s[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const uint16_t* xr = x + (int64_t)row * cols;
  float m = -INFINITY, ssum = 0.0f;
  for (int c = tid * 8; c < cols; c += 2048) {
    const v4i v = *reinterpret_cast<const v4i*>(xr + c);
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = __uint_as_float((u32)v[j] << 16);
      f[2 * j + 1] = __uint_as_float((u32)v[j] & 0xFFFF0000u);
    }
    float vm = f[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) vm = fmaxf(vm, f[j]);
    const float nm = fmaxf(m, vm);
    float add = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) add += __expf(f[j] - nm);
    ssum = ssum * __expf(m - nm) + add;
    m = nm;
  }
  // combine (m, ssum) across the wave, then across the 4 waves
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float om = __shfl_xor(m, o), os = __shfl_xor(ssum, o);
    const float nm = fmaxf(m, om);
    ssum = (nm == -INFINITY) ? 0.0f : ssum * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
  }
  if ((tid & 63) == 0) { s_m[tid >> 6] = m; s_s[tid >> 6] = ssum; }
  __syncthreads();
  if (tid == 0) {
    float M = s_m[0], S = s_s[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float nm = fmaxf(M, s_m[w]);
      S = S * __expf(M - nm) + s_s[w] * __expf(s_m[w] - nm);
      M = nm;
    }
    const float l = M + __logf(S);
    lse[row] = l;
    const int64_t lab = labels[row];
    loss[row] = (lab >= 0 && lab < cols) ? l - bf16_bits_to_float(xr[lab]) : 0.0f;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const uint16_t* __restrict__ x, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ gscale,
                                                     uint16_t* __restrict__ dx, int cols) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const int64_t lab = labels[row];
  const bool valid = lab >= 0 && lab < cols;
  const float gs = valid ? *gscale : 0.0f;
  const float l = lse[row];
  const uint16_t* xr = x + (int64_t)row * cols;
  uint16_t* dr = dx + (int64_t)row * cols;
  for (int c = tid * 8; c < cols; c += 2048) {
    const v4i v = *reinterpret_cast<const v4i*>(xr + c);
    v4i o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float p0 = __expf(__uint_as_float((u32)v[j] << 16) - l), p1 = __expf(__uint_as_float((u32)v[j] & 0xFFFF0000u) - l);
      if (c + 2 * j == lab) p0 -= 1.0f;
      if (c + 2 * j + 1 == lab) p1 -= 1.0f;
      o[j] = (int)pack_bf16x2(p0 * gs, p1 * gs);
    }
    *reinterpret_cast<v4i*>(dr + c) = o;
  }
}

// Cross-entropy backward fused with the FP8 cast of its result: d(logits) [rows, cols] is grad_output of the lm_head
// Linear (2.1 GB in bf16 for Llama-3.2-3B at 8192 tokens), whose backward would read it once more just to quantise it.
// Same arithmetic as ce_bwd_kernel, rounded to bf16, then mi_cast_amax's: the FP8 bytes and the amax equal the two-kernel
// sequence bit for bit.  One 128 x 128 tile per workgroup, 8 x 8 block per lane (the tiling of norm_cast_kernel).
template <int FMT, bool WRITE_Y, bool WRITE_T>
__global__ __launch_bounds__(256) void ce_bwd_cast_kernel(const uint16_t* __restrict__ x, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ lse, const float* __restrict__ gscale,
                                                          uint8_t* __restrict__ y, uint8_t* __restrict__ yT,
                                                          const float* __restrict__ scale_p, float* amax_out, int rows, int cols,
                                                          int tiles_c) {
  __shared__ float s_amax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x % tiles_c;
  const int r0 = tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int c0 = tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  const float scale = *scale_p, g0 = *gscale;
  float amax = 0.0f;
  if ((r0 < rows) && (c0 < cols)) {
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t lab = labels[r0 + i];
      const bool valid = lab >= 0 && lab < cols;
      const float gs = valid ? g0 : 0.0f;
      const float l = lse[r0 + i];
      const v4i raw = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(x + (int64_t)(r0 + i) * cols + c0));
      float f[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p0 = __expf(__uint_as_float((u32)raw[j] << 16) - l), p1 = __expf(__uint_as_float((u32)raw[j] & 0xFFFF0000u) - l);
        if (c0 + 2 * j == lab) p0 -= 1.0f;
        if (c0 + 2 * j + 1 == lab) p1 -= 1.0f;
        f[2 * j] = __uint_as_float(float_to_bf16_bits(p0 * gs) << 16);
        f[2 * j + 1] = __uint_as_float(float_to_bf16_bits(p1 * gs) << 16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, (f[j] != f[j]) ? 0.0f : fabsf(f[j]));
      lo[i] = cvt4_fp8<FMT>(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
      hi[i] = cvt4_fp8<FMT>(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
    }
    if (WRITE_Y) {
      uint8_t* dst = y + (int64_t)r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) mi::st8<MI_NT_Y>(dst + (int64_t)i * cols, lo[i], hi[i]);
    }
    if (WRITE_T) {
      u32 a[4], b[4], c[4], d[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], a[0], a[1], a[2], a[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], b[0], b[1], b[2], b[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], c[0], c[1], c[2], c[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], d[0], d[1], d[2], d[3]);
      uint8_t* dst = yT + (int64_t)c0 * rows + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mi::st8<MI_NT_YT>(dst + (int64_t)j * rows, a[j], b[j]);
        mi::st8<MI_NT_YT>(dst + (int64_t)(j + 4) * rows, c[j], d[j]);
      }
    }
  }
  if (amax_out != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      const float m = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
      // only a workgroup that would raise the value goes to the atomic unit: same-address atomics are served one at a time and a
      // wave cannot retire before its atomic has returned (thousands of workgroups per launch)
      if (m > 0.0f && m > __builtin_nontemporal_load(amax_out)) atomicMax(reinterpret_cast<unsigned int*>(amax_out), __float_as_uint(m));
    }
  }
}

}  // namespace mi

extern "C" int mi_rope_qkv(void* fused_bf16, void* q_bf16, void* k_bf16, void* v_bf16, const float* cos_tab,
                           const float* sin_tab, int64_t rows, int64_t seq, int n_q_heads, int n_kv_heads, int head_dim,
                           int backward, void* stream) {
  MI_CHECK_ARG(fused_bf16 && q_bf16 && k_bf16 && v_bf16 && cos_tab && sin_tab, "mi_rope_qkv: null pointer");
  MI_CHECK_ARG(rows >= 0 && seq > 0 && n_q_heads > 0 && n_kv_heads > 0, "mi_rope_qkv: bad shape");
  MI_CHECK_ARG(head_dim > 0 && head_dim % 16 == 0, "mi_rope_qkv: head_dim must be a multiple of 16");
  MI_CHECK_ARG(((uintptr_t)fused_bf16 % 16) == 0 && ((uintptr_t)q_bf16 % 16) == 0 && ((uintptr_t)k_bf16 % 16) == 0 &&
                   ((uintptr_t)v_bf16 % 16) == 0 && ((uintptr_t)cos_tab % 16) == 0 && ((uintptr_t)sin_tab % 16) == 0,
               "mi_rope_qkv: pointers must be 16-byte aligned");
  MI_CHECK_ARG(rows < (1LL << 31), "mi_rope_qkv: too many rows");
  if (rows == 0) return MI_OK;
  const int64_t per_row = (int64_t)(n_q_heads + n_kv_heads) * (head_dim / 16) + (int64_t)n_kv_heads * head_dim / 8;
  int64_t blocks = (rows * per_row + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (backward)
    hipLaunchKernelGGL(mi::rope_qkv_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)fused_bf16,
                       (uint16_t*)q_bf16, (uint16_t*)k_bf16, (uint16_t*)v_bf16, cos_tab, sin_tab, (int)rows, (int)seq, n_q_heads,
                       n_kv_heads, head_dim);
  else
    hipLaunchKernelGGL(mi::rope_qkv_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)fused_bf16,
                       (uint16_t*)q_bf16, (uint16_t*)k_bf16, (uint16_t*)v_bf16, cos_tab, sin_tab, (int)rows, (int)seq, n_q_heads,
                       n_kv_heads, head_dim);
  MI_CHECK_LAUNCH("mi_rope_qkv launch");
  return MI_OK;
}

extern "C" int mi_rope_qkv_bwd_cast(const void* dq_bf16, const void* dk_bf16, const void* dv_bf16, const float* cos_tab,
                                    const float* sin_tab, void* y_fp8, void* yT_fp8, const float* scale, float* amax, int64_t rows,
                                    int64_t seq, int n_q_heads, int n_kv_heads, int head_dim, int fmt, void* stream) {
  MI_CHECK_ARG(dq_bf16 && dk_bf16 && dv_bf16 && cos_tab && sin_tab && scale, "mi_rope_qkv_bwd_cast: null pointer");
  MI_CHECK_ARG(y_fp8 || yT_fp8, "mi_rope_qkv_bwd_cast: at least one of y, yT must be non-null");
  MI_CHECK_ARG(head_dim == 128, "mi_rope_qkv_bwd_cast: head_dim %d not supported (128)", head_dim);
  MI_CHECK_ARG(n_q_heads >= 1 && n_kv_heads >= 1 && seq >= 1 && rows >= 0 && rows % 8 == 0 && rows < (1LL << 31),
               "mi_rope_qkv_bwd_cast: bad shape (rows a multiple of 8)");
  MI_CHECK_ARG(((uintptr_t)dq_bf16 % 16) == 0 && ((uintptr_t)dk_bf16 % 16) == 0 && ((uintptr_t)dv_bf16 % 16) == 0 &&
                   ((uintptr_t)cos_tab % 16) == 0 && ((uintptr_t)sin_tab % 16) == 0 && ((uintptr_t)y_fp8 % 8) == 0 &&
                   ((uintptr_t)yT_fp8 % 8) == 0, "mi_rope_qkv_bwd_cast: misaligned pointer");
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "mi_rope_qkv_bwd_cast: bad fmt %d", fmt);
  if (rows == 0) return MI_OK;
  const int64_t units = ((rows + 63) / 64) * (int64_t)(n_q_heads + 3 * n_kv_heads);
  dim3 grid((unsigned)((units + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MI_RC(FMTv)                                                                                                          \
  if (y_fp8 && yT_fp8)                                                                                                       \
    hipLaunchKernelGGL((mi::rope_bwd_cast_kernel<FMTv, true, true>), grid, block, 0, st, (const uint16_t*)dq_bf16,           \
                       (const uint16_t*)dk_bf16, (const uint16_t*)dv_bf16, cos_tab, sin_tab, (uint8_t*)y_fp8, (uint8_t*)yT_fp8, \
                       scale, amax, (int)rows, (int)seq, n_q_heads, n_kv_heads);                                             \
  else if (y_fp8)                                                                                                            \
    hipLaunchKernelGGL((mi::rope_bwd_cast_kernel<FMTv, true, false>), grid, block, 0, st, (const uint16_t*)dq_bf16,          \
                       (const uint16_t*)dk_bf16, (const uint16_t*)dv_bf16, cos_tab, sin_tab, (uint8_t*)y_fp8, (uint8_t*)yT_fp8, \
                       scale, amax, (int)rows, (int)seq, n_q_heads, n_kv_heads);                                             \
  else                                                                                                                       \
    hipLaunchKernelGGL((mi::rope_bwd_cast_kernel<FMTv, false, true>), grid, block, 0, st, (const uint16_t*)dq_bf16,          \
                       (const uint16_t*)dk_bf16, (const uint16_t*)dv_bf16, cos_tab, sin_tab, (uint8_t*)y_fp8, (uint8_t*)yT_fp8, \
                       scale, amax, (int)rows, (int)seq, n_q_heads, n_kv_heads);
  if (fmt == MI_FMT_E4M3) { MI_RC(MI_FMT_E4M3) } else { MI_RC(MI_FMT_E5M2) }
#undef MI_RC
  MI_CHECK_LAUNCH("mi_rope_qkv_bwd_cast launch");
  return MI_OK;
}

static int swiglu_common_check(const char* who, const void* h, const void* y, const void* yT, const float* scale,
                               int64_t rows, int64_t F, int fmt) {
  MI_CHECK_ARG(h && scale, "%s: null input", who);
  MI_CHECK_ARG(y || yT, "%s: at least one of y, yT must be non-null", who);
  MI_CHECK_ARG(rows >= 0 && F >= 0 && rows % 8 == 0 && F % 8 == 0, "%s: rows and F must be multiples of 8", who);
  MI_CHECK_ARG(rows < (1LL << 31) && 2 * F < (1LL << 31), "%s: shape too large", who);
  MI_CHECK_ARG(((uintptr_t)h % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)yT % 8) == 0, "%s: misaligned pointer", who);
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "%s: bad fmt %d", who, fmt);
  return MI_OK;
}

extern "C" int mi_swiglu_cast(const void* h_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                              int64_t rows, int64_t F, int fmt, void* stream) {
  int rc = swiglu_common_check("mi_swiglu_cast", h_bf16, y_fp8, yT_fp8, scale, rows, F, fmt);
  if (rc != MI_OK) return rc;
  if (rows == 0 || F == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3) return mi::launch_swiglu<MI_FMT_E4M3, 0>(h_bf16, nullptr, y_fp8, yT_fp8, scale, amax, nullptr, rows, F, st);
  return mi::launch_swiglu<MI_FMT_E5M2, 0>(h_bf16, nullptr, y_fp8, yT_fp8, scale, amax, nullptr, rows, F, st);
}

extern "C" int mi_dswiglu_cast(const void* h_bf16, const void* dact_bf16, void* y_fp8, void* yT_fp8, const float* scale,
                               float* amax, float* colsum, int64_t rows, int64_t F, int fmt, void* stream) {
  int rc = swiglu_common_check("mi_dswiglu_cast", h_bf16, y_fp8, yT_fp8, scale, rows, F, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(dact_bf16 && ((uintptr_t)dact_bf16 % 16) == 0, "mi_dswiglu_cast: dact must be non-null and 16-byte aligned");
  if (rows == 0 || F == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3)
    return mi::launch_swiglu<MI_FMT_E4M3, 1>(h_bf16, dact_bf16, y_fp8, yT_fp8, scale, amax, colsum, rows, F, st);
  return mi::launch_swiglu<MI_FMT_E5M2, 1>(h_bf16, dact_bf16, y_fp8, yT_fp8, scale, amax, colsum, rows, F, st);
}

// mi_swiglu_cast / mi_dswiglu_cast with the fc1 bias (bf16 [2F], 16-byte aligned) added to h inside the kernel: `h` is the fc1
// GEMM's output WITHOUT bias (te_llama.py:58-63: LayerNormMLP keeps TE's default bias=True; TE's own bias + activation fusion)
extern "C" int mi_swiglu_cast_bias(const void* h_bf16, const void* bias_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                                   int64_t rows, int64_t F, int fmt, void* stream) {
  int rc = swiglu_common_check("mi_swiglu_cast_bias", h_bf16, y_fp8, yT_fp8, scale, rows, F, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(bias_bf16 && ((uintptr_t)bias_bf16 % 16) == 0, "mi_swiglu_cast_bias: bias must be non-null and 16-byte aligned");
  if (rows == 0 || F == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3) return mi::launch_swiglu<MI_FMT_E4M3, 0>(h_bf16, nullptr, y_fp8, yT_fp8, scale, amax, nullptr, rows, F, st, bias_bf16);
  return mi::launch_swiglu<MI_FMT_E5M2, 0>(h_bf16, nullptr, y_fp8, yT_fp8, scale, amax, nullptr, rows, F, st, bias_bf16);
}

extern "C" int mi_dswiglu_cast_bias(const void* h_bf16, const void* bias_bf16, const void* dact_bf16, void* y_fp8, void* yT_fp8,
                                    const float* scale, float* amax, float* colsum, int64_t rows, int64_t F, int fmt, void* stream) {
  int rc = swiglu_common_check("mi_dswiglu_cast_bias", h_bf16, y_fp8, yT_fp8, scale, rows, F, fmt);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(dact_bf16 && ((uintptr_t)dact_bf16 % 16) == 0, "mi_dswiglu_cast_bias: dact must be non-null and 16-byte aligned");
  MI_CHECK_ARG(bias_bf16 && ((uintptr_t)bias_bf16 % 16) == 0, "mi_dswiglu_cast_bias: bias must be non-null and 16-byte aligned");
  if (rows == 0 || F == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (fmt == MI_FMT_E4M3)
    return mi::launch_swiglu<MI_FMT_E4M3, 1>(h_bf16, dact_bf16, y_fp8, yT_fp8, scale, amax, colsum, rows, F, st, bias_bf16);
  return mi::launch_swiglu<MI_FMT_E5M2, 1>(h_bf16, dact_bf16, y_fp8, yT_fp8, scale, amax, colsum, rows, F, st, bias_bf16);
}

namespace mi {
// out = a + b (fp32 sum, one bf16 rounding: what torch's bf16 add yields) and rstd of the ROUNDED sum in the same pass: the
// residual add of a decoder layer feeds the next RMSNorm, whose statistics pass would re-read `out` from HBM.
// `bias` (optional, bf16 [cols]): b is the fc2 GEMM's output WITHOUT its bias; out = a + (b + bias), the inner sum in fp32.
__global__ __launch_bounds__(256) void add_rmsnorm_stats_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                                uint16_t* __restrict__ out, float* __restrict__ rstd, int rows,
                                                                int cols, float eps, const uint16_t* __restrict__ bias) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t off = (int64_t)row * cols;
  float acc = 0.0f;
  for (int c = lane * 8; c < cols; c += 512) {
    const v4i va = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(a + off + c));
    const v4i vb = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b + off + c));
    v4i vbias = {0, 0, 0, 0};
    if (bias != nullptr) vbias = *reinterpret_cast<const v4i*>(bias + c);
    v4i vo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 wa = (u32)va[j], wb = (u32)vb[j];
      float b0 = __uint_as_float(wb << 16), b1 = __uint_as_float(wb & 0xFFFF0000u);
      if (bias != nullptr) {
        b0 += __uint_as_float((u32)vbias[j] << 16);
        b1 += __uint_as_float((u32)vbias[j] & 0xFFFF0000u);
      }
      const u32 o = pack_bf16x2(__uint_as_float(wa << 16) + b0, __uint_as_float(wa & 0xFFFF0000u) + b1);
      const float lo = __uint_as_float(o << 16), hi = __uint_as_float(o & 0xFFFF0000u);
      acc += lo * lo + hi * hi;
      vo[j] = (int)o;
    }
    *reinterpret_cast<v4i*>(out + off + c) = vo;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) rstd[row] = rsqrtf(acc / (float)cols + eps);
}
}  // namespace mi

extern "C" int mi_add_rmsnorm_stats(const void* a_bf16, const void* b_bf16, void* out_bf16, float* rstd, int64_t rows, int64_t cols,
                                    float eps, void* stream) {
  MI_CHECK_ARG(a_bf16 && b_bf16 && out_bf16 && rstd, "mi_add_rmsnorm_stats: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31), "mi_add_rmsnorm_stats: bad shape");
  MI_CHECK_ARG((((uintptr_t)a_bf16 | (uintptr_t)b_bf16 | (uintptr_t)out_bf16) % 16) == 0, "mi_add_rmsnorm_stats: operands must be 16-byte aligned");
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(mi::add_rmsnorm_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)a_bf16, (const uint16_t*)b_bf16, (uint16_t*)out_bf16, rstd, (int)rows, (int)cols, eps,
                     (const uint16_t*)nullptr);
  MI_CHECK_LAUNCH("mi_add_rmsnorm_stats launch");
  return MI_OK;
}

extern "C" int mi_add_bias_rmsnorm_stats(const void* a_bf16, const void* b_bf16, const void* bias_bf16, void* out_bf16, float* rstd,
                                         int64_t rows, int64_t cols, float eps, void* stream) {
  MI_CHECK_ARG(a_bf16 && b_bf16 && bias_bf16 && out_bf16 && rstd, "mi_add_bias_rmsnorm_stats: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31), "mi_add_bias_rmsnorm_stats: bad shape");
  MI_CHECK_ARG((((uintptr_t)a_bf16 | (uintptr_t)b_bf16 | (uintptr_t)out_bf16 | (uintptr_t)bias_bf16) % 16) == 0,
               "mi_add_bias_rmsnorm_stats: operands must be 16-byte aligned");
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(mi::add_rmsnorm_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)a_bf16, (const uint16_t*)b_bf16, (uint16_t*)out_bf16, rstd, (int)rows, (int)cols, eps,
                     (const uint16_t*)bias_bf16);
  MI_CHECK_LAUNCH("mi_add_bias_rmsnorm_stats launch");
  return MI_OK;
}

extern "C" int mi_rmsnorm_stats(const void* x_bf16, float* rstd, int64_t rows, int64_t cols, float eps, void* stream) {
  MI_CHECK_ARG(x_bf16 && rstd, "mi_rmsnorm_stats: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31), "mi_rmsnorm_stats: bad shape");
  MI_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0, "mi_rmsnorm_stats: x must be 16-byte aligned");
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(mi::rmsnorm_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x_bf16, rstd, (int)rows, (int)cols, eps);
  MI_CHECK_LAUNCH("mi_rmsnorm_stats launch");
  return MI_OK;
}

extern "C" int mi_norm_cast(const void* x_bf16, const float* rstd, const void* gamma_bf16, void* y_fp8, void* yT_fp8,
                            const float* scale, float* amax, int64_t rows, int64_t cols, int fmt, void* stream) {
  MI_CHECK_ARG(x_bf16 && rstd && gamma_bf16 && scale, "mi_norm_cast: null input");
  MI_CHECK_ARG(y_fp8 || yT_fp8, "mi_norm_cast: at least one of y, yT must be non-null");
  MI_CHECK_ARG(rows >= 0 && cols >= 0 && rows % 8 == 0 && cols % 8 == 0, "mi_norm_cast: rows and cols must be multiples of 8");
  MI_CHECK_ARG(rows < (1LL << 31) && cols < (1LL << 31), "mi_norm_cast: shape too large");
  MI_CHECK_ARG(((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)gamma_bf16 % 16) == 0 && ((uintptr_t)rstd % 16) == 0 &&
                   ((uintptr_t)y_fp8 % 8) == 0 && ((uintptr_t)yT_fp8 % 8) == 0, "mi_norm_cast: misaligned pointer");
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "mi_norm_cast: bad fmt %d", fmt);
  if (rows == 0 || cols == 0) return MI_OK;
  MI_CHECK_ARG(((rows + 63) / 64) * ((cols + 63) / 64) < (1LL << 31), "mi_norm_cast: shape too large");
  const int tiles_r = (int)((rows + 63) / 64), tiles_c = (int)((cols + 63) / 64);  // 64 x 64 tiles, one per wave and step
  const int64_t wgs = ((int64_t)tiles_r * tiles_c + 3) / 4;
  int ncu = 256, devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || ncu <= 0) ncu = 256;
  const int64_t cap = (int64_t)ncu * 2;  // two workgroups per CU, a whole multiple of the CU count (profiles/r03_cast_grid_sweep.txt)
  dim3 grid((unsigned)(wgs < cap ? wgs : cap)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const uint16_t *xp = (const uint16_t*)x_bf16, *gp = (const uint16_t*)gamma_bf16;
  uint8_t *yp = (uint8_t*)y_fp8, *tp = (uint8_t*)yT_fp8;
#define MI_NC(FMTv)                                                                                                        \
  if (y_fp8 && yT_fp8)                                                                                                     \
    hipLaunchKernelGGL((mi::norm_cast_kernel<FMTv, true, true>), grid, block, 0, st, xp, rstd, gp, yp, tp, scale, amax, (int)rows, (int)cols, tiles_c); \
  else if (y_fp8)                                                                                                          \
    hipLaunchKernelGGL((mi::norm_cast_kernel<FMTv, true, false>), grid, block, 0, st, xp, rstd, gp, yp, tp, scale, amax, (int)rows, (int)cols, tiles_c); \
  else                                                                                                                     \
    hipLaunchKernelGGL((mi::norm_cast_kernel<FMTv, false, true>), grid, block, 0, st, xp, rstd, gp, yp, tp, scale, amax, (int)rows, (int)cols, tiles_c);
  if (fmt == MI_FMT_E4M3) { MI_NC(MI_FMT_E4M3) } else { MI_NC(MI_FMT_E5M2) }
#undef MI_NC
  MI_CHECK_LAUNCH("mi_norm_cast launch");
  return MI_OK;
}

extern "C" int mi_rmsnorm_bwd(const void* dy_bf16, const void* x_bf16, const float* rstd, const void* gamma_bf16,
                              const void* dres_bf16, void* dx_bf16, float* dgamma_partial, int n_partials, int64_t rows,
                              int64_t cols, void* stream) {
  MI_CHECK_ARG(dy_bf16 && x_bf16 && rstd && gamma_bf16 && dx_bf16 && dgamma_partial, "mi_rmsnorm_bwd: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 512 == 0 && cols <= 512 * mi::kNormMaxVec, "mi_rmsnorm_bwd: cols must be a multiple of 512, at most %d", 512 * mi::kNormMaxVec);
  MI_CHECK_ARG(n_partials >= 1 && rows < (1LL << 31), "mi_rmsnorm_bwd: bad sizes");
  MI_CHECK_ARG(((uintptr_t)dy_bf16 % 16) == 0 && ((uintptr_t)x_bf16 % 16) == 0 && ((uintptr_t)gamma_bf16 % 16) == 0 &&
                   ((uintptr_t)dres_bf16 % 16) == 0 && ((uintptr_t)dx_bf16 % 16) == 0, "mi_rmsnorm_bwd: misaligned pointer");
  const int rows_per_block = (int)((rows + n_partials - 1) / n_partials);
  const int rpb = rows_per_block < 1 ? 1 : rows_per_block;
  const size_t shm = (size_t)(4 * cols * sizeof(float));
#define MI_RB(NV, WPRv)                                                                                                   \
  case NV:                                                                                                                \
    hipLaunchKernelGGL((mi::rmsnorm_bwd_kernel<NV, WPRv>), dim3(n_partials), dim3(256), shm, (hipStream_t)stream,         \
                       (const uint16_t*)dy_bf16, (const uint16_t*)x_bf16, rstd, (const uint16_t*)gamma_bf16,              \
                       (const uint16_t*)dres_bf16, (uint16_t*)dx_bf16, dgamma_partial, (int)rows, rpb);                   \
    break;
  switch ((int)(cols / 512)) {
    MI_RB(1, 1) MI_RB(2, 1) MI_RB(3, 1) MI_RB(4, 2) MI_RB(5, 1) MI_RB(6, 2) MI_RB(7, 1) MI_RB(8, 2) MI_RB(10, 2) MI_RB(12, 4)
    MI_RB(14, 2) MI_RB(16, 4)
    default:
      mi::set_error("mi_rmsnorm_bwd: unsupported width %lld (cols/512 must be 1-8, 10, 12, 14 or 16)", (long long)cols);
      return MI_ERR_SHAPE;
  }
#undef MI_RB
  MI_CHECK_LAUNCH("mi_rmsnorm_bwd launch");
  return MI_OK;
}

extern "C" int mi_ce_forward(const void* logits_bf16, const int64_t* labels, float* lse, float* loss_rows, int64_t rows,
                             int64_t cols, void* stream) {
  MI_CHECK_ARG(logits_bf16 && labels && lse && loss_rows, "mi_ce_forward: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31), "mi_ce_forward: bad shape (cols % 8)");
  MI_CHECK_ARG(((uintptr_t)logits_bf16 % 16) == 0, "mi_ce_forward: logits must be 16-byte aligned");
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(mi::ce_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits_bf16, labels,
                     lse, loss_rows, (int)cols);
  MI_CHECK_LAUNCH("mi_ce_forward launch");
  return MI_OK;
}

extern "C" int mi_ce_backward(const void* logits_bf16, const int64_t* labels, const float* lse, const float* gscale,
                              void* dlogits_bf16, int64_t rows, int64_t cols, void* stream) {
  MI_CHECK_ARG(logits_bf16 && labels && lse && gscale && dlogits_bf16, "mi_ce_backward: null pointer");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31), "mi_ce_backward: bad shape (cols % 8)");
  MI_CHECK_ARG(((uintptr_t)logits_bf16 % 16) == 0 && ((uintptr_t)dlogits_bf16 % 16) == 0, "mi_ce_backward: misaligned pointer");
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(mi::ce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits_bf16, labels,
                     lse, gscale, (uint16_t*)dlogits_bf16, (int)cols);
  MI_CHECK_LAUNCH("mi_ce_backward launch");
  return MI_OK;
}

extern "C" int mi_ce_backward_cast(const void* logits_bf16, const int64_t* labels, const float* lse, const float* gscale,
                                   void* y_fp8, void* yT_fp8, const float* scale, float* amax, int64_t rows, int64_t cols, int fmt,
                                   void* stream) {
  MI_CHECK_ARG(logits_bf16 && labels && lse && gscale && scale, "mi_ce_backward_cast: null pointer");
  MI_CHECK_ARG(y_fp8 || yT_fp8, "mi_ce_backward_cast: at least one of y, yT must be non-null");
  MI_CHECK_ARG(rows >= 0 && cols > 0 && rows % 8 == 0 && cols % 8 == 0 && rows < (1LL << 31) && cols < (1LL << 31),
               "mi_ce_backward_cast: rows and cols must be multiples of 8");
  MI_CHECK_ARG(((uintptr_t)logits_bf16 % 16) == 0 && ((uintptr_t)y_fp8 % 8) == 0 && ((uintptr_t)yT_fp8 % 8) == 0,
               "mi_ce_backward_cast: misaligned pointer");
  MI_CHECK_ARG(fmt == MI_FMT_E4M3 || fmt == MI_FMT_E5M2, "mi_ce_backward_cast: bad fmt %d", fmt);
  if (rows == 0) return MI_OK;
  const int tiles_r = (int)((rows + 127) / 128), tiles_c = (int)((cols + 127) / 128);
  MI_CHECK_ARG((int64_t)tiles_r * tiles_c < (1LL << 31), "mi_ce_backward_cast: shape too large");
  dim3 grid((unsigned)(tiles_r * tiles_c)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MI_CC(FMTv, WY, WT)                                                                                              \
  hipLaunchKernelGGL((mi::ce_bwd_cast_kernel<FMTv, WY, WT>), grid, block, 0, st, (const uint16_t*)logits_bf16, labels, lse, \
                     gscale, (uint8_t*)y_fp8, (uint8_t*)yT_fp8, scale, amax, (int)rows, (int)cols, tiles_c)
#define MI_CC_F(FMTv)                                    \
  if (y_fp8 && yT_fp8) MI_CC(FMTv, true, true);          \
  else if (y_fp8) MI_CC(FMTv, true, false);              \
  else MI_CC(FMTv, false, true);
  if (fmt == MI_FMT_E4M3) { MI_CC_F(MI_FMT_E4M3) } else { MI_CC_F(MI_FMT_E5M2) }
#undef MI_CC_F
#undef MI_CC
  MI_CHECK_LAUNCH("mi_ce_backward_cast launch");
  return MI_OK;
}
