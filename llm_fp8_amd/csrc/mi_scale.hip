// K3: amax-history roll + delayed-scaling scale update for every slot of a recipe group in ONE
// launch (latency-bound; the Python side keeps all modules' meta of one recipe in one
// [H, S] buffer so a whole model costs one launch per direction per step).
// Replaces TE's fused amax-and-scale update reached from fp8_autocast exit
// (te_llama.py:76,79).
#include "mi_common.h"
#include <float.h>

namespace mi {

constexpr int kMaxPerThread = 16;  // H <= 256 * 16

__global__ __launch_bounds__(256) void scale_update_kernel(float* __restrict__ hist, float* __restrict__ scale,
                                                           float* __restrict__ scale_inv,
                                                           const float* __restrict__ fp8_max, int H, int64_t S,
                                                           float inv_margin_pow, int algo) {
  __shared__ float s_red[4];
  const int s = blockIdx.x, tid = threadIdx.x;
  float vals[kMaxPerThread];
  float m = 0.0f;
  // read rolled source: new[h] = old[(h+1) % H]
#pragma unroll
  for (int i = 0; i < kMaxPerThread; ++i) {
    int h = tid + i * 256;
    if (h < H) {
      int src = (h + 1 == H) ? 0 : h + 1;
      float v = hist[(int64_t)src * S + s];
      vals[i] = v;
      m = fmaxf(m, v);  // NaN ignored
    }
  }
  const float first = hist[s];  // old row 0 (most recent)
  m = wave_max(m);
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
  __syncthreads();  // also orders all reads before the writes below
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
#pragma unroll
  for (int i = 0; i < kMaxPerThread; ++i) {
    int h = tid + i * 256;
    if (h < H) hist[(int64_t)h * S + s] = (h == 0) ? 0.0f : vals[i];
  }
  if (tid == 0) {
    float amax = (algo == MI_AMAX_ALGO_MAX) ? m : first;
    float old = scale[s];
    float sf = (fp8_max[s] / amax) * inv_margin_pow;  // exact: inv_margin_pow is a power of two
    if (!(amax > 0.0f) || !isfinite(amax)) sf = old;
    else if (isinf(sf)) sf = FLT_MAX;
    scale[s] = sf;
    scale_inv[s] = 1.0f / sf;
  }
}

}  // namespace mi

extern "C" int mi_scale_update(float* amax_history, float* scale, float* scale_inv, const float* fp8_max, int H,
                               int S, int64_t ld, int margin, int algo, void* stream) {
  MI_CHECK_ARG(amax_history && scale && scale_inv && fp8_max, "mi_scale_update: null pointer");
  MI_CHECK_ARG(H >= 1 && H <= 256 * mi::kMaxPerThread, "mi_scale_update: H=%d out of range [1,%d]", H,
               256 * mi::kMaxPerThread);
  MI_CHECK_ARG(S >= 0 && ld >= S, "mi_scale_update: need 0 <= S <= ld");
  MI_CHECK_ARG(margin > -100 && margin < 100, "mi_scale_update: margin out of range");
  MI_CHECK_ARG(algo == MI_AMAX_ALGO_MAX || algo == MI_AMAX_ALGO_MOST_RECENT, "mi_scale_update: bad algo %d", algo);
  if (S == 0) return MI_OK;
  float inv_margin_pow = ldexpf(1.0f, -margin);
  hipLaunchKernelGGL(mi::scale_update_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, amax_history, scale,
                     scale_inv, fp8_max, H, ld, inv_margin_pow, algo);
  MI_CHECK_LAUNCH("mi_scale_update launch");
  return MI_OK;
}
