// Optimiser step of the reference's loop (train_fp8.py:288-291: clip_grad_norm_(1.0) -> AdamW(fused=True).step()),
// HBM-bound: the clip coefficient is folded into the AdamW pass (no separate in-place scaling of the gradients) and the
// squared-norm reduction is one streaming pass with fixed-order partial sums (bitwise reproducible).
//   mi_sumsq_bf16   partial[b] = sum over block b's elements of g^2 (fp32), b < n_partials
//   mi_adamw_bf16   torch.optim.AdamW semantics for bf16 parameters with bf16 exp_avg / exp_avg_sq, fp32 math
#include "mi_common.h"

namespace mi {

__global__ __launch_bounds__(256) void sumsq_kernel(const uint16_t* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float s_red[4];
  const int tid = threadIdx.x;
  float acc = 0.0f;
  const int64_t nvec = n >> 3;
  const bool aligned = ((uintptr_t)g & 15) == 0;
  if (aligned) {
    // four independent 16-byte loads in flight per lane and iteration (the pass is latency-bound otherwise)
    const int64_t stride = (int64_t)gridDim.x * 256;
    float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t i = (int64_t)blockIdx.x * 256 + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      v4i v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i + u * stride);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 w = (u32)v[u][j];
          const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xFFFF0000u);
          acc4[u] += a * a + b * b;
        }
    }
    for (; i < nvec; i += stride) {
      const v4i v = reinterpret_cast<const v4i*>(g)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 w = (u32)v[j];
        const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xFFFF0000u);
        acc4[0] += a * a + b * b;
      }
    }
    acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    for (int64_t i = (nvec << 3) + (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
      const float a = bf16_bits_to_float(g[i]);
      acc += a * a;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
      const float a = bf16_bits_to_float(g[i]);
      acc += a * a;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  p *= (1.0f - a.lr * a.weight_decay);
  m = m + (1.0f - a.beta1) * (g - m);
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(uint16_t* __restrict__ p, const uint16_t* __restrict__ g,
                                                    uint16_t* __restrict__ m, uint16_t* __restrict__ v, int64_t n,
                                                    const float* __restrict__ grad_scale, AdamArgs a) {
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const int64_t nvec = n >> 3;
  const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (aligned) {
    for (int64_t i = t0; i < nvec; i += stride) {
      v4i pv = reinterpret_cast<const v4i*>(p)[i];
      const v4i gv = reinterpret_cast<const v4i*>(g)[i];
      v4i mv = reinterpret_cast<const v4i*>(m)[i];
      v4i vv = reinterpret_cast<const v4i*>(v)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
        const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
        float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
        float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
        adam_one(pl, gl, ml, vl, a);
        adam_one(ph, gh, mh, vh, a);
        pv[j] = (int)pack_bf16x2(pl, ph);
        mv[j] = (int)pack_bf16x2(ml, mh);
        vv[j] = (int)pack_bf16x2(vl, vh);
      }
      reinterpret_cast<v4i*>(p)[i] = pv;
      reinterpret_cast<v4i*>(m)[i] = mv;
      reinterpret_cast<v4i*>(v)[i] = vv;
    }
  }
  for (int64_t i = (aligned ? (nvec << 3) : 0) + t0; i < n; i += stride) {
    float pf = bf16_bits_to_float(p[i]), mf = bf16_bits_to_float(m[i]), vf = bf16_bits_to_float(v[i]);
    adam_one(pf, gs * bf16_bits_to_float(g[i]), mf, vf, a);
    p[i] = (uint16_t)float_to_bf16_bits(pf);
    m[i] = (uint16_t)float_to_bf16_bits(mf);
    v[i] = (uint16_t)float_to_bf16_bits(vf);
  }
}

// ---- multi-tensor forms: ONE launch over every tensor of a parameter group.  `tab` [5, T] int64 on the device: rows of
// p / g / exp_avg / exp_avg_sq addresses and element counts; `chunks` [nchunks] = (tensor index, chunk index inside it); a
// workgroup owns one chunk of `chunk_elems` elements.  (A 3B model has ~280 parameter tensors: 2 x 280 launches per step,
// most of them a few microseconds long, become 2.)
struct ChunkRef {
  int tensor, chunk;
};

__global__ __launch_bounds__(256) void sumsq_multi_kernel(const int64_t* __restrict__ tab, int T, const ChunkRef* __restrict__ chunks,
                                                          int chunk_elems, double* __restrict__ partial) {
  // The squares of bf16 values are exact in fp32 (8 x 8 significand bits); they are ACCUMULATED in fp64, so a chunk's partial does
  // not depend on how the elements are dealt to lanes, and the total does not depend on how the tensors are cut into chunks,
  // groups or row shards beyond 2^-52: distributed.ShardedFP8DP's clip coefficient is then the replicated run's, bit for bit
  // (in fp32 the chunking moved the last bit of the norm in a few percent of the steps).  The kernel stays HBM-bound.
  __shared__ double s_red[4];
  const ChunkRef cr = chunks[blockIdx.x];
  const uint16_t* g = reinterpret_cast<const uint16_t*>(tab[(int64_t)1 * T + cr.tensor]);
  const int64_t n = tab[(int64_t)4 * T + cr.tensor];
  const int64_t lo = (int64_t)cr.chunk * chunk_elems, hi = min(n, lo + chunk_elems);
  const int tid = threadIdx.x;
  double acc0 = 0.0, acc1 = 0.0;
  if ((((uintptr_t)g) & 15) == 0) {  // chunk_elems is a multiple of 8: chunk starts stay 16-byte aligned
    const int64_t v0 = lo >> 3, v1 = hi >> 3;
    int64_t i = v0 + tid;
    for (; i + 256 < v1; i += 512) {
      const v4i a = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i);
      const v4i b = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i + 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 wa = (u32)a[j], wb = (u32)b[j];
        const float a0 = __uint_as_float(wa << 16), a1 = __uint_as_float(wa & 0xFFFF0000u);
        const float b0 = __uint_as_float(wb << 16), b1 = __uint_as_float(wb & 0xFFFF0000u);
        acc0 += (double)(a0 * a0) + (double)(a1 * a1);
        acc1 += (double)(b0 * b0) + (double)(b1 * b1);
      }
    }
    for (; i < v1; i += 256) {
      const v4i a = reinterpret_cast<const v4i*>(g)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 wa = (u32)a[j];
        const float a0 = __uint_as_float(wa << 16), a1 = __uint_as_float(wa & 0xFFFF0000u);
        acc0 += (double)(a0 * a0) + (double)(a1 * a1);
      }
    }
    for (int64_t k = (v1 << 3) + tid; k < hi; k += 256) {
      const float a = bf16_bits_to_float(g[k]);
      acc0 += (double)(a * a);
    }
  } else {
    for (int64_t k = lo + tid; k < hi; k += 256) {
      const float a = bf16_bits_to_float(g[k]);
      acc0 += (double)(a * a);
    }
  }
  double acc = acc0 + acc1;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const int64_t* __restrict__ tab, int T, const ChunkRef* __restrict__ chunks,
                                                          int chunk_elems, const float* __restrict__ grad_scale, AdamArgs a) {
  const ChunkRef cr = chunks[blockIdx.x];
  uint16_t* p = reinterpret_cast<uint16_t*>(tab[cr.tensor]);
  const uint16_t* g = reinterpret_cast<const uint16_t*>(tab[(int64_t)1 * T + cr.tensor]);
  uint16_t* m = reinterpret_cast<uint16_t*>(tab[(int64_t)2 * T + cr.tensor]);
  uint16_t* v = reinterpret_cast<uint16_t*>(tab[(int64_t)3 * T + cr.tensor]);
  const int64_t n = tab[(int64_t)4 * T + cr.tensor];
  const int64_t lo = (int64_t)cr.chunk * chunk_elems, hi = min(n, lo + chunk_elems);
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const int tid = threadIdx.x;
  const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  int64_t done = lo;
  if (aligned) {
    const int64_t v0 = lo >> 3, v1 = hi >> 3;
    for (int64_t i = v0 + tid; i < v1; i += 256) {
      v4i pv = reinterpret_cast<const v4i*>(p)[i];
      const v4i gv = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i);
      v4i mv = reinterpret_cast<const v4i*>(m)[i];
      v4i vv = reinterpret_cast<const v4i*>(v)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
        const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
        float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
        float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
        adam_one(pl, gl, ml, vl, a);
        adam_one(ph, gh, mh, vh, a);
        pv[j] = (int)pack_bf16x2(pl, ph);
        mv[j] = (int)pack_bf16x2(ml, mh);
        vv[j] = (int)pack_bf16x2(vl, vh);
      }
      reinterpret_cast<v4i*>(p)[i] = pv;
      reinterpret_cast<v4i*>(m)[i] = mv;
      reinterpret_cast<v4i*>(v)[i] = vv;
    }
    done = v1 << 3;
  }
  for (int64_t k = done + tid; k < hi; k += 256) {
    float pf = bf16_bits_to_float(p[k]), mf = bf16_bits_to_float(m[k]), vf = bf16_bits_to_float(v[k]);
    adam_one(pf, gs * bf16_bits_to_float(g[k]), mf, vf, a);
    p[k] = (uint16_t)float_to_bf16_bits(pf);
    m[k] = (uint16_t)float_to_bf16_bits(mf);
    v[k] = (uint16_t)float_to_bf16_bits(vf);
  }
}

// AdamW + FP8 weight cast in one pass (the weight-cast hand-off): under delayed scaling the scale the NEXT
// forward quantises a weight with is already final when the optimiser runs (the forward arena was updated at the end of this
// step's forward), and the optimiser streams every weight anyway.  For tensors with a "sink" (cols > 0) the chunk is a 128 x 128
// tile of the [rows, cols] weight (the tiling of cast_amax_kernel: 8 x 8 block per lane, in-register byte transpose): each
// element is updated, rounded to bf16, stored -- and that ROUNDED value is quantised exactly as mi_cast_amax would
// (y = sat(float(bf16) * scale), amax = max |bf16|), into y [rows, ld_y] and the transposed copy yT [cols, ld_yT].
// Tensors without a sink (cols == 0) take the flat path of adamw_multi_kernel.  Table rows (int64 each, T columns):
//   0 p  1 g  2 exp_avg  3 exp_avg_sq  4 numel  5 cols  6 y  7 yT  8 ld_y  9 ld_yT  10 scale ptr  11 amax ptr
__global__ __launch_bounds__(256) void adamw_cast_multi_kernel(const int64_t* __restrict__ tab, int T, const ChunkRef* __restrict__ chunks,
                                                               int chunk_elems, const float* __restrict__ grad_scale, AdamArgs a) {
  __shared__ float s_amax[4];
  const ChunkRef cr = chunks[blockIdx.x];
  uint16_t* p = reinterpret_cast<uint16_t*>(tab[cr.tensor]);
  const uint16_t* g = reinterpret_cast<const uint16_t*>(tab[(int64_t)1 * T + cr.tensor]);
  uint16_t* m = reinterpret_cast<uint16_t*>(tab[(int64_t)2 * T + cr.tensor]);
  uint16_t* v = reinterpret_cast<uint16_t*>(tab[(int64_t)3 * T + cr.tensor]);
  const int64_t n = tab[(int64_t)4 * T + cr.tensor];
  const int64_t cols = tab[(int64_t)5 * T + cr.tensor];
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const int tid = threadIdx.x;
  if (cols == 0) {  // flat chunk (same arithmetic and traversal as adamw_multi_kernel)
    const int64_t lo = (int64_t)cr.chunk * chunk_elems, hi = min(n, lo + chunk_elems);
    const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    int64_t done = lo;
    if (aligned) {
      const int64_t v0 = lo >> 3, v1 = hi >> 3;
      for (int64_t i = v0 + tid; i < v1; i += 256) {
        v4i pv = reinterpret_cast<const v4i*>(p)[i];
        const v4i gv = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i);
        v4i mv = reinterpret_cast<const v4i*>(m)[i];
        v4i vv = reinterpret_cast<const v4i*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
          const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
          float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
          float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
          adam_one(pl, gl, ml, vl, a);
          adam_one(ph, gh, mh, vh, a);
          pv[j] = (int)pack_bf16x2(pl, ph);
          mv[j] = (int)pack_bf16x2(ml, mh);
          vv[j] = (int)pack_bf16x2(vl, vh);
        }
        reinterpret_cast<v4i*>(p)[i] = pv;
        reinterpret_cast<v4i*>(m)[i] = mv;
        reinterpret_cast<v4i*>(v)[i] = vv;
      }
      done = v1 << 3;
    }
    for (int64_t k = done + tid; k < hi; k += 256) {
      float pf = bf16_bits_to_float(p[k]), mf = bf16_bits_to_float(m[k]), vf = bf16_bits_to_float(v[k]);
      adam_one(pf, gs * bf16_bits_to_float(g[k]), mf, vf, a);
      p[k] = (uint16_t)float_to_bf16_bits(pf);
      m[k] = (uint16_t)float_to_bf16_bits(mf);
      v[k] = (uint16_t)float_to_bf16_bits(vf);
    }
    return;
  }
  // tile of a weight with an FP8 sink
  uint8_t* y = reinterpret_cast<uint8_t*>(tab[(int64_t)6 * T + cr.tensor]);
  uint8_t* yT = reinterpret_cast<uint8_t*>(tab[(int64_t)7 * T + cr.tensor]);
  const int64_t ld_y = tab[(int64_t)8 * T + cr.tensor], ld_yT = tab[(int64_t)9 * T + cr.tensor];
  const float scale = *reinterpret_cast<const float*>(tab[(int64_t)10 * T + cr.tensor]);
  float* amax_out = reinterpret_cast<float*>(tab[(int64_t)11 * T + cr.tensor]);
  const int64_t rows = n / cols;
  const int tiles_c = (int)((cols + 127) >> 7);
  const int lane = tid & 63, wave = tid >> 6;
  const int tile_r = cr.chunk / tiles_c, tile_c = cr.chunk % tiles_c;
  const int64_t r0 = (int64_t)tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int64_t c0 = (int64_t)tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  float amax = 0.0f;
  if (r0 < rows && c0 < cols) {
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t idx = ((r0 + i) * cols + c0) >> 3;
      v4i pv = reinterpret_cast<const v4i*>(p)[idx];
      const v4i gv = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + idx);
      v4i mv = reinterpret_cast<const v4i*>(m)[idx];
      v4i vv = reinterpret_cast<const v4i*>(v)[idx];
      float f[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
        const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
        float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
        float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
        adam_one(pl, gl, ml, vl, a);
        adam_one(ph, gh, mh, vh, a);
        const u32 pw = pack_bf16x2(pl, ph);
        pv[j] = (int)pw;
        mv[j] = (int)pack_bf16x2(ml, mh);
        vv[j] = (int)pack_bf16x2(vl, vh);
        f[2 * j] = __uint_as_float(pw << 16);          // the ROUNDED weight: what the next forward's cast would read
        f[2 * j + 1] = __uint_as_float(pw & 0xFFFF0000u);
      }
      reinterpret_cast<v4i*>(p)[idx] = pv;
      reinterpret_cast<v4i*>(m)[idx] = mv;
      reinterpret_cast<v4i*>(v)[idx] = vv;
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, (f[j] != f[j]) ? 0.0f : fabsf(f[j]));
      lo[i] = cvt4_fp8<MI_FMT_E4M3>(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
      hi[i] = cvt4_fp8<MI_FMT_E4M3>(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
    }
    if (y != nullptr) {
      uint8_t* dst = y + r0 * ld_y + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // nontemporal: the copies are read again only by the NEXT forward; as write-back lines the 64-byte row segments cost
        // 4.0 ms per step on the 3B parameter set, streamed 1.3 ms (tools/bench_adamw.py)
        typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
        const v2u_ w = {lo[i], hi[i]};
        __builtin_nontemporal_store(w, reinterpret_cast<v2u_*>(dst + (int64_t)i * ld_y));
      }
    }
    if (yT != nullptr) {
      u32 ta[4], tb[4], tc[4], td[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], ta[0], ta[1], ta[2], ta[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], tb[0], tb[1], tb[2], tb[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], tc[0], tc[1], tc[2], tc[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], td[0], td[1], td[2], td[3]);
      uint8_t* dst = yT + c0 * ld_yT + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
        const v2u_ w0 = {ta[j], tb[j]}, w1 = {tc[j], td[j]};
        __builtin_nontemporal_store(w0, reinterpret_cast<v2u_*>(dst + (int64_t)j * ld_yT));
        __builtin_nontemporal_store(w1, reinterpret_cast<v2u_*>(dst + (int64_t)(j + 4) * ld_yT));
      }
    }
  }
  if (amax_out != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      const float mx = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
      if (mx > 0.0f && mx > __builtin_nontemporal_load(amax_out)) atomicMax(reinterpret_cast<unsigned int*>(amax_out), __float_as_uint(mx));
    }
  }
}

// AdamW + MXFP8 weight quantisation in one pass: the MXFP8 form of the weight-cast hand-off.  Block scaling has no state -- the
// E8M0 scale of a 32-element block comes from the block itself -- so the copies the NEXT forward needs (row-wise blocks for
// fprop, column-wise blocks, stored transposed, for dgrad) can be emitted as soon as the weight is updated.  Same tile walk as
// adamw_cast_multi_kernel (128 x 128 tile per workgroup, 8 x 8 block per lane); the quantisation of the ROUNDED bf16 weight is
// mxfp8_quant_kernel's, operation for operation (bitwise the bytes mi_mxfp8_quantize would produce from the updated weight).
// Table rows (int64 each, T columns):
//   0 p  1 g  2 exp_avg  3 exp_avg_sq  4 numel  5 cols (0 = no sink: flat path)  6 y_row  7 s_row  8 y_colT  9 s_colT
//   10 ldr (rows of the whole operand the tensor is a row-block of)  11 unused
// y_row / s_row / y_colT / s_colT point at the tensor's first row inside the operand's buffers: y_row [rows, cols],
// s_row [cols/32, ldr], y_colT [cols, ldr], s_colT [rows/32, cols].
__global__ __launch_bounds__(256) void adamw_mxcast_multi_kernel(const int64_t* __restrict__ tab, int T, const ChunkRef* __restrict__ chunks,
                                                                 int chunk_elems, const float* __restrict__ grad_scale, AdamArgs a) {
  const ChunkRef cr = chunks[blockIdx.x];
  uint16_t* p = reinterpret_cast<uint16_t*>(tab[cr.tensor]);
  const uint16_t* g = reinterpret_cast<const uint16_t*>(tab[(int64_t)1 * T + cr.tensor]);
  uint16_t* m = reinterpret_cast<uint16_t*>(tab[(int64_t)2 * T + cr.tensor]);
  uint16_t* v = reinterpret_cast<uint16_t*>(tab[(int64_t)3 * T + cr.tensor]);
  const int64_t n = tab[(int64_t)4 * T + cr.tensor];
  const int64_t cols = tab[(int64_t)5 * T + cr.tensor];
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const int tid = threadIdx.x;
  if (cols == 0) {  // flat chunk (same arithmetic and traversal as adamw_multi_kernel)
    const int64_t lo = (int64_t)cr.chunk * chunk_elems, hi = min(n, lo + chunk_elems);
    const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    int64_t done = lo;
    if (aligned) {
      const int64_t v0 = lo >> 3, v1 = hi >> 3;
      for (int64_t i = v0 + tid; i < v1; i += 256) {
        v4i pv = reinterpret_cast<const v4i*>(p)[i];
        const v4i gv = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + i);
        v4i mv = reinterpret_cast<const v4i*>(m)[i];
        v4i vv = reinterpret_cast<const v4i*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
          const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
          float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
          float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
          adam_one(pl, gl, ml, vl, a);
          adam_one(ph, gh, mh, vh, a);
          pv[j] = (int)pack_bf16x2(pl, ph);
          mv[j] = (int)pack_bf16x2(ml, mh);
          vv[j] = (int)pack_bf16x2(vl, vh);
        }
        reinterpret_cast<v4i*>(p)[i] = pv;
        reinterpret_cast<v4i*>(m)[i] = mv;
        reinterpret_cast<v4i*>(v)[i] = vv;
      }
      done = v1 << 3;
    }
    for (int64_t k = done + tid; k < hi; k += 256) {
      float pf = bf16_bits_to_float(p[k]), mf = bf16_bits_to_float(m[k]), vf = bf16_bits_to_float(v[k]);
      adam_one(pf, gs * bf16_bits_to_float(g[k]), mf, vf, a);
      p[k] = (uint16_t)float_to_bf16_bits(pf);
      m[k] = (uint16_t)float_to_bf16_bits(mf);
      v[k] = (uint16_t)float_to_bf16_bits(vf);
    }
    return;
  }
  uint8_t* y_row = reinterpret_cast<uint8_t*>(tab[(int64_t)6 * T + cr.tensor]);
  uint8_t* s_row = reinterpret_cast<uint8_t*>(tab[(int64_t)7 * T + cr.tensor]);
  uint8_t* y_colT = reinterpret_cast<uint8_t*>(tab[(int64_t)8 * T + cr.tensor]);
  uint8_t* s_colT = reinterpret_cast<uint8_t*>(tab[(int64_t)9 * T + cr.tensor]);
  const int64_t ldr = tab[(int64_t)10 * T + cr.tensor];
  const int64_t rows = n / cols;
  const int tiles_c = (int)((cols + 127) >> 7);
  const int lane = tid & 63, wave = tid >> 6;
  const int tile_r = cr.chunk / tiles_c, tile_c = cr.chunk % tiles_c;
  const int64_t r0 = (int64_t)tile_r * 128 + (wave >> 1) * 64 + (lane >> 3) * 8;
  const int64_t c0 = (int64_t)tile_c * 128 + (wave & 1) * 64 + (lane & 7) * 8;
  const bool active = r0 < rows && c0 < cols;  // rows, cols multiples of 32: a 4-lane block group is all-active or all-inactive
  const float rcp = 1.0f / fp8_max_of<MI_FMT_E4M3>();
  float f[8][8];
  unsigned long long nanmask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (active) {
      const int64_t idx = ((r0 + i) * cols + c0) >> 3;
      v4i pv = reinterpret_cast<const v4i*>(p)[idx];
      const v4i gv = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g) + idx);
      v4i mv = reinterpret_cast<const v4i*>(m)[idx];
      v4i vv = reinterpret_cast<const v4i*>(v)[idx];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pl = __uint_as_float((u32)pv[j] << 16), ph = __uint_as_float((u32)pv[j] & 0xFFFF0000u);
        const float gl = gs * __uint_as_float((u32)gv[j] << 16), gh = gs * __uint_as_float((u32)gv[j] & 0xFFFF0000u);
        float ml = __uint_as_float((u32)mv[j] << 16), mh = __uint_as_float((u32)mv[j] & 0xFFFF0000u);
        float vl = __uint_as_float((u32)vv[j] << 16), vh = __uint_as_float((u32)vv[j] & 0xFFFF0000u);
        adam_one(pl, gl, ml, vl, a);
        adam_one(ph, gh, mh, vh, a);
        const u32 pw = pack_bf16x2(pl, ph);
        pv[j] = (int)pw;
        mv[j] = (int)pack_bf16x2(ml, mh);
        vv[j] = (int)pack_bf16x2(vl, vh);
        f[i][2 * j] = __uint_as_float(pw << 16);  // the ROUNDED weight: what the next forward's quantiser would read
        f[i][2 * j + 1] = __uint_as_float(pw & 0xFFFF0000u);
      }
      reinterpret_cast<v4i*>(p)[idx] = pv;
      reinterpret_cast<v4i*>(m)[idx] = mv;
      reinterpret_cast<v4i*>(v)[idx] = vv;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.0f;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (f[i][j] != f[i][j]) {  // as mxfp8_quant_kernel: a NaN does not enter the block amax and leaves as 0x7F
        nanmask |= 1ull << (8 * i + j);
        f[i][j] = 0.0f;
      }
  auto patch4 = [&](u32 w, int bit0) -> u32 {
    if (__builtin_expect(nanmask != 0, 0)) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((nanmask >> (bit0 + k)) & 1) w = (w & ~(0xFFu << (8 * k))) | (0x7Fu << (8 * k));
    }
    return w;
  };
  typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
  {  // row-wise blocks (32 columns = 4 lanes)
    u32 sbytes[8], lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float am = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(f[i][j]));
      am = fmaxf(am, __shfl_xor(am, 1));
      am = fmaxf(am, __shfl_xor(am, 2));
      const u32 e = e8m0_roundup(am * rcp);
      const float inv = e8m0_inv(e);
      sbytes[i] = e;
      lo[i] = patch4(cvt4_fp8<MI_FMT_E4M3>(f[i][0] * inv, f[i][1] * inv, f[i][2] * inv, f[i][3] * inv), 8 * i);
      hi[i] = patch4(cvt4_fp8<MI_FMT_E4M3>(f[i][4] * inv, f[i][5] * inv, f[i][6] * inv, f[i][7] * inv), 8 * i + 4);
    }
    if (active) {
      uint8_t* dst = y_row + r0 * cols + c0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v2u_ w = {lo[i], hi[i]};
        __builtin_nontemporal_store(w, reinterpret_cast<v2u_*>(dst + (int64_t)i * cols));
      }
      if ((lane & 3) == 0) {
        const v2u_ w = {sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24),
                        sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24)};
        *reinterpret_cast<v2u_*>(s_row + (c0 / 32) * ldr + r0) = w;
      }
    }
  }
  {  // column-wise blocks (32 rows = 4 lane rows), stored transposed
    u32 sbytes[8];
    float inv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float am = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(f[i][j]));
      am = fmaxf(am, __shfl_xor(am, 8));
      am = fmaxf(am, __shfl_xor(am, 16));
      const u32 e = e8m0_roundup(am * rcp);
      sbytes[j] = e;
      inv[j] = e8m0_inv(e);
    }
    u32 lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      lo[i] = patch4(cvt4_fp8<MI_FMT_E4M3>(f[i][0] * inv[0], f[i][1] * inv[1], f[i][2] * inv[2], f[i][3] * inv[3]), 8 * i);
      hi[i] = patch4(cvt4_fp8<MI_FMT_E4M3>(f[i][4] * inv[4], f[i][5] * inv[5], f[i][6] * inv[6], f[i][7] * inv[7]), 8 * i + 4);
    }
    if (active) {
      u32 ta[4], tb[4], tc[4], td[4];
      transpose4x4(lo[0], lo[1], lo[2], lo[3], ta[0], ta[1], ta[2], ta[3]);
      transpose4x4(lo[4], lo[5], lo[6], lo[7], tb[0], tb[1], tb[2], tb[3]);
      transpose4x4(hi[0], hi[1], hi[2], hi[3], tc[0], tc[1], tc[2], tc[3]);
      transpose4x4(hi[4], hi[5], hi[6], hi[7], td[0], td[1], td[2], td[3]);
      uint8_t* dst = y_colT + c0 * ldr + r0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v2u_ w0 = {ta[j], tb[j]}, w1 = {tc[j], td[j]};
        __builtin_nontemporal_store(w0, reinterpret_cast<v2u_*>(dst + (int64_t)j * ldr));
        __builtin_nontemporal_store(w1, reinterpret_cast<v2u_*>(dst + (int64_t)(j + 4) * ldr));
      }
      if (((lane >> 3) & 3) == 0) {
        const v2u_ w = {sbytes[0] | (sbytes[1] << 8) | (sbytes[2] << 16) | (sbytes[3] << 24),
                        sbytes[4] | (sbytes[5] << 8) | (sbytes[6] << 16) | (sbytes[7] << 24)};
        *reinterpret_cast<v2u_*>(s_colT + (r0 / 32) * cols + c0) = w;
      }
    }
  }
}

// Embedding weight gradient added IN PLACE into an existing [V, H] bf16 gradient (the tied lm_head / embedding table already
// holds the lm_head wgrad): grad[id, :] += alpha * sum over the tokens with that id of dY[token, :].  Replaces
// aten::embedding_dense_backward (zero-fill of a dense [V, H] + scatter) followed by a dense add -- 2 x 788 MB written and
// 3 x 788 MB read for Llama-3.2-3B -- by a pass over the touched rows only.  Deterministic: ids are sorted (stable) by the
// caller, the workgroup at the head of a run of equal ids sums that run in order in fp32 and is the only writer of the row.
__global__ __launch_bounds__(256) void embedding_grad_add_kernel(uint16_t* __restrict__ grad, const uint16_t* __restrict__ dy,
                                                                 const int64_t* __restrict__ sorted_ids,
                                                                 const int64_t* __restrict__ perm, int T, int H, int64_t V,
                                                                 float alpha, int64_t padding_idx) {
  const int i0 = blockIdx.x;
  const int64_t id = sorted_ids[i0];
  if (i0 > 0 && sorted_ids[i0 - 1] == id) return;  // not the head of its run
  if (id < 0 || id >= V || id == padding_idx) return;
  int i1 = i0 + 1;
  while (i1 < T && sorted_ids[i1] == id) ++i1;
  for (int c = threadIdx.x * 8; c < H; c += 2048) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = i0; i < i1; ++i) {
      const v4i v = *reinterpret_cast<const v4i*>(dy + perm[i] * (int64_t)H + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __uint_as_float((u32)v[j] << 16);
        acc[2 * j + 1] += __uint_as_float((u32)v[j] & 0xFFFF0000u);
      }
    }
    uint16_t* gp = grad + id * (int64_t)H + c;
    const v4i g = *reinterpret_cast<const v4i*>(gp);
    v4i o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = (int)pack_bf16x2(__uint_as_float((u32)g[j] << 16) + alpha * acc[2 * j],
                              __uint_as_float((u32)g[j] & 0xFFFF0000u) + alpha * acc[2 * j + 1]);
    *reinterpret_cast<v4i*>(gp) = o;
  }
}

}  // namespace mi

static mi::AdamArgs make_adam_args(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step) {
  mi::AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  return a;
}

extern "C" int mi_sumsq_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                   double* partial, void* stream) {
  MI_CHECK_ARG(table && chunks && partial, "mi_sumsq_bf16_multi: null pointer");
  MI_CHECK_ARG(n_tensors >= 1 && n_chunks >= 1 && chunk_elems >= 8 && chunk_elems % 8 == 0, "mi_sumsq_bf16_multi: bad sizes");
  hipLaunchKernelGGL(mi::sumsq_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, n_tensors,
                     (const mi::ChunkRef*)chunks, chunk_elems, partial);
  MI_CHECK_LAUNCH("mi_sumsq_bf16_multi launch");
  return MI_OK;
}

extern "C" int mi_adamw_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                   const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   int64_t step, void* stream) {
  MI_CHECK_ARG(table && chunks, "mi_adamw_bf16_multi: null pointer");
  MI_CHECK_ARG(n_tensors >= 1 && n_chunks >= 1 && chunk_elems >= 8 && chunk_elems % 8 == 0 && step >= 1, "mi_adamw_bf16_multi: bad sizes");
  hipLaunchKernelGGL(mi::adamw_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, n_tensors,
                     (const mi::ChunkRef*)chunks, chunk_elems, grad_scale, make_adam_args(lr, beta1, beta2, eps, weight_decay, step));
  MI_CHECK_LAUNCH("mi_adamw_bf16_multi launch");
  return MI_OK;
}

extern "C" int mi_adamw_cast_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                        const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int64_t step, void* stream) {
  MI_CHECK_ARG(table && chunks, "mi_adamw_cast_bf16_multi: null pointer");
  MI_CHECK_ARG(n_tensors >= 1 && n_chunks >= 1 && chunk_elems >= 8 && chunk_elems % 8 == 0 && step >= 1, "mi_adamw_cast_bf16_multi: bad sizes");
  hipLaunchKernelGGL(mi::adamw_cast_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, n_tensors,
                     (const mi::ChunkRef*)chunks, chunk_elems, grad_scale, make_adam_args(lr, beta1, beta2, eps, weight_decay, step));
  MI_CHECK_LAUNCH("mi_adamw_cast_bf16_multi launch");
  return MI_OK;
}

extern "C" int mi_adamw_mxcast_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                          const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                                          int64_t step, void* stream) {
  MI_CHECK_ARG(table && chunks, "mi_adamw_mxcast_bf16_multi: null pointer");
  MI_CHECK_ARG(n_tensors >= 1 && n_chunks >= 1 && chunk_elems >= 8 && chunk_elems % 8 == 0 && step >= 1, "mi_adamw_mxcast_bf16_multi: bad sizes");
  hipLaunchKernelGGL(mi::adamw_mxcast_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, n_tensors,
                     (const mi::ChunkRef*)chunks, chunk_elems, grad_scale, make_adam_args(lr, beta1, beta2, eps, weight_decay, step));
  MI_CHECK_LAUNCH("mi_adamw_mxcast_bf16_multi launch");
  return MI_OK;
}

extern "C" int mi_sumsq_bf16(const void* g_bf16, int64_t n, float* partial, int n_partials, void* stream) {
  MI_CHECK_ARG(g_bf16 && partial, "mi_sumsq_bf16: null pointer");
  MI_CHECK_ARG(n >= 0 && n_partials >= 1 && n_partials <= 65535, "mi_sumsq_bf16: bad sizes");
  hipLaunchKernelGGL(mi::sumsq_kernel, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)g_bf16, n, partial);
  MI_CHECK_LAUNCH("mi_sumsq_bf16 launch");
  return MI_OK;
}

extern "C" int mi_adamw_bf16(void* p_bf16, const void* g_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16, int64_t n,
                             const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t step, void* stream) {
  MI_CHECK_ARG(p_bf16 && g_bf16 && exp_avg_bf16 && exp_avg_sq_bf16, "mi_adamw_bf16: null pointer");
  MI_CHECK_ARG(n >= 0 && step >= 1, "mi_adamw_bf16: bad n / step");
  if (n == 0) return MI_OK;
  const mi::AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, step);
  int64_t blocks = ((n >> 3) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mi::adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)p_bf16,
                     (const uint16_t*)g_bf16, (uint16_t*)exp_avg_bf16, (uint16_t*)exp_avg_sq_bf16, n, grad_scale, a);
  MI_CHECK_LAUNCH("mi_adamw_bf16 launch");
  return MI_OK;
}

extern "C" int mi_embedding_grad_add(void* grad_bf16, const void* dy_bf16, const int64_t* sorted_ids, const int64_t* perm,
                                     int64_t tokens, int64_t hidden, int64_t vocab, float alpha, int64_t padding_idx, void* stream) {
  MI_CHECK_ARG(grad_bf16 && dy_bf16 && sorted_ids && perm, "mi_embedding_grad_add: null pointer");
  MI_CHECK_ARG(tokens >= 0 && tokens < (1LL << 31) && hidden > 0 && hidden % 8 == 0 && hidden < (1LL << 31) && vocab > 0,
               "mi_embedding_grad_add: bad shape (hidden a multiple of 8)");
  MI_CHECK_ARG(((uintptr_t)grad_bf16 % 16) == 0 && ((uintptr_t)dy_bf16 % 16) == 0, "mi_embedding_grad_add: misaligned pointer");
  if (tokens == 0) return MI_OK;
  hipLaunchKernelGGL(mi::embedding_grad_add_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (uint16_t*)grad_bf16,
                     (const uint16_t*)dy_bf16, sorted_ids, perm, (int)tokens, (int)hidden, vocab, alpha, padding_idx);
  MI_CHECK_LAUNCH("mi_embedding_grad_add launch");
  return MI_OK;
}
