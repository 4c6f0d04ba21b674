// LDS-fed MFMA probe (GPU box): does a 128x128 wave tile at ONE wave per SIMD (half the LDS->VGPR bytes per FLOP... -33 % against
// 128x64 at two waves per SIMD) hold a higher clock / rate on random FP8 bytes?  No global traffic, no barriers: every wave
// re-reads its A / B fragments of a 256x256x128 tile image from LDS each K-step (read_frag's conflict-free layout) and issues its
// MFMAs; the reads of step s+1 are issued before the MFMAs of step s (two register sets).
//   mode A: 8 waves (2 per SIMD), wave tile 128x64:  8 + 4 fragments -> 32 MFMAs per K-step   (the shipped kernel's shape)
//   mode B: 4 waves (1 per SIMD), wave tile 128x128: 8 + 8 fragments -> 64 MFMAs per K-step
// hipcc -O3 --offload-arch=gfx950 -I llm_fp8_amd/csrc tools/probe_wave_tile.hip -o tools/bin/probe_wave_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz_f(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ v8i read_frag(const uint8_t* lds_tile, int g, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const int f = swz_f(r);
  const uint8_t* base = lds_tile + g * 2048 + (r >> 3) * 1024 + (r & 7) * 128;
  v4i lo = *reinterpret_cast<const v4i*>(base + ((q ^ f) << 4));
  v4i hi = *reinterpret_cast<const v4i*>(base + (((4 + q) ^ f) << 4));
  return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NB>  // B fragments per wave: 4 (mode A) or 8 (mode B)
__global__ __launch_bounds__(NB == 4 ? 512 : 256, NB == 4 ? 2 : 1) void k(const v4i* src, float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * 256 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 256 * 128 / 16; i += blockDim.x) reinterpret_cast<v4i*>(lds)[i] = src[i];
  __syncthreads();
  const int wr = NB == 4 ? wave >> 2 : wave >> 1, wc = NB == 4 ? wave & 3 : wave & 1;
  const uint8_t* at = lds, *bt = lds + 256 * 128;
  v4f acc[8][NB];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = (v4f){0, 0, 0, 0};
  v8i a0[8], b0[NB], a1[8], b1[NB];
#pragma unroll
  for (int i = 0; i < 8; ++i) a0[i] = read_frag(at, wr * 8 + i, lane);
#pragma unroll
  for (int j = 0; j < NB; ++j) b0[j] = read_frag(bt, wc * NB + j, lane);
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[i] = read_frag(at, wr * 8 + i, lane);
#pragma unroll
    for (int j = 0; j < NB; ++j) b1[j] = read_frag(bt, wc * NB + j, lane);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0[i], b0[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
    for (int i = 0; i < 8; ++i) a0[i] = read_frag(at, wr * 8 + i, lane);
#pragma unroll
    for (int j = 0; j < NB; ++j) b0[j] = read_frag(bt, wc * NB + j, lane);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1[i], b1[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < NB; ++j) sum += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + tid] = sum;
  if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  v4i* src; float* out; unsigned long long* clk;
  (void)hipMalloc(&src, 65536); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 16);
  unsigned char* h = (unsigned char*)malloc(65536);
  for (int fill = 0; fill < 2; ++fill) {
    for (int i = 0; i < 65536; ++i) { unsigned char v = rand() & 0xff; if ((v & 0x7f) >= 0x78) v &= 0x3f; h[i] = fill ? 0 : v; }
    (void)hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
      const int iters = 20000;
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int q = 0; q < 8; ++q) {
          if (mode == 0) hipLaunchKernelGGL(k<4>, 256, 512, 0, 0, src, out, clk, iters);
          else hipLaunchKernelGGL(k<8>, 256, 256, 0, 0, src, out, clk, iters);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[512]; (void)hipMemcpy(hc, clk, 4096, hipMemcpyDeviceToHost);
        const double flop = 8.0 * 256 * (double)iters * 2.0 * 256 * 256 * 128;  // one 256x256x128 tile step per workgroup and iteration
        printf("%s %s: %.3f PFLOP/s  cycles per K-step %.1f (2048 = MFMA-bound)  clock %.0f MHz\n", fill ? "zeros " : "random",
               mode == 0 ? "A 2 waves/SIMD 128x64 " : "B 1 wave/SIMD  128x128", flop / (ms * 1e-3) / 1e15, (double)hc[0] / iters, (double)hc[0] / hc[1] * 100.0);
      }
    }
  }
  return 0;
}
