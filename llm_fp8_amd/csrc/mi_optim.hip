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

}  // namespace mi

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
  mi::AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  int64_t blocks = ((n >> 3) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mi::adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)p_bf16,
                     (const uint16_t*)g_bf16, (uint16_t*)exp_avg_bf16, (uint16_t*)exp_avg_sq_bf16, n, grad_scale, a);
  MI_CHECK_LAUNCH("mi_adamw_bf16 launch");
  return MI_OK;
}
