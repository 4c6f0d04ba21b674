// Bare MFMA-rate probe (GPU box): 16x16x128 vs 32x32x64 f8f6f4 on random / zero operands, registers only,
// 2 waves per SIMD (512-thread blocks, 1 per CU), reports PFLOP/s and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const v8i* src, float* out, unsigned long long* clk, int iters) {
  const int l = threadIdx.x;
  v8i a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(l + 64 * i) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = src[(l + 64 * (i + 4)) % 4096];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0;
  if (SHAPE == 16) {
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
  } else if (SHAPE == 17) {  // 16x16x128 with per-lane E8M0 scales in VGPRs (the MXFP8 form)
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
    int sa[4], sb[2];
    for (int i = 0; i < 4; ++i) sa[i] = 0x7c + ((a[i][0] >> 3) & 7);
    for (int i = 0; i < 2; ++i) sb[i] = 0x7c + ((b[i][0] >> 5) & 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 2 + j], 0, 0, 0, sa[i], 0, sb[j]);
    }
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
  } else if (SHAPE == 116 || SHAPE == 132) {
    // operands re-read from LDS every iteration (ds_read_b128 x 2 per fragment, conflict-free: lane l reads 32 contiguous bytes
    // at 32 l + 2048 f), same LDS bytes per FLOP for both shapes: 6 fragments of 2 KiB per 8 x 16x16x128 = 4 x 32x32x64
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * 2048 * 8];
    const int w = l >> 6, ln = l & 63;
    for (int i = ln; i < 6 * 2048 / 16; i += 64) reinterpret_cast<int4*>(lds + w * 6 * 2048)[i] = reinterpret_cast<const int4*>(src)[(i + w * 768) % 8192];
    __syncthreads();
    const v8i* f = reinterpret_cast<const v8i*>(lds + w * 6 * 2048) + ln;
    if (SHAPE == 116) {
      v4f acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
        v8i aa[4], bb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) aa[i] = f[64 * i];
#pragma unroll
        for (int i = 0; i < 2; ++i) bb[i] = f[64 * (4 + i)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa[i], bb[j], acc[i * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
    } else {
      v16f acc[2];
      for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
      for (int it = 0; it < iters; ++it) {
        v8i aa[4], bb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) aa[i] = f[64 * i];
#pragma unroll
        for (int i = 0; i < 2; ++i) bb[i] = f[64 * (4 + i)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aa[kk * 2 + j], bb[kk], acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      for (int i = 0; i < 2; ++i) sum += acc[i][0] + acc[i][7];
    }
  } else {
    v16f acc[2];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    for (int it = 0; it < iters; ++it) {
      // same FLOPs per iteration: 8 x (16x16x128) = 2 x 2 x (32x32x64)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[kk * 2 + j], b[kk], acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    for (int i = 0; i < 2; ++i) sum += acc[i][0] + acc[i][7];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + l] = sum;
  if (l == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  v8i* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 4096 * 32); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
  unsigned char* h = (unsigned char*)malloc(4096 * 32);
  for (int fill = 0; fill < 2; ++fill) {
    for (int i = 0; i < 4096 * 32; ++i) { unsigned char v = rand() & 0xff; if ((v & 0x7f) >= 0x78) v &= 0x3f; h[i] = fill ? 0 : v; }
    hipMemcpy(src, h, 4096 * 32, hipMemcpyHostToDevice);
    for (int shape : {16, 17, 32, 116, 132}) {
      const int iters = 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int q = 0; q < 8; ++q) {
          if (shape == 16) hipLaunchKernelGGL(k<16>, 256, 512, 0, 0, src, out, clk, iters);
          else if (shape == 17) hipLaunchKernelGGL(k<17>, 256, 512, 0, 0, src, out, clk, iters);
          else if (shape == 116) hipLaunchKernelGGL(k<116>, 256, 512, 0, 0, src, out, clk, iters);
          else if (shape == 132) hipLaunchKernelGGL(k<132>, 256, 512, 0, 0, src, out, clk, iters);
          else hipLaunchKernelGGL(k<32>, 256, 512, 0, 0, src, out, clk, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[512]; hipMemcpy(hc, clk, 4096, hipMemcpyDeviceToHost);
        double flop = 8.0 * 256 * 8 * iters * 8 * 2.0 * 16 * 16 * 128;
        printf("%s shape %d (1xx = operands re-read from LDS): %.3f PFLOP/s  cycles/iter %.1f (ideal 256 per wave-pair... ) clock %.0f MHz\n", fill ? "zeros " : "random", shape,
               flop / (ms * 1e-3) / 1e15, (double)hc[0] / iters, (double)hc[0] / hc[1] * 100.0);
      }
    }
  }
  return 0;
}
