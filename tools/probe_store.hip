// Store-path probe (MI355X): what does one CU sustain when 8 waves stream 16-byte-per-lane buffer stores, by access shape,
// cache policy and number of CUs storing at once?  Answers whether the GEMM epilogue (128 KiB per CU and tile) is bound by the
// CU's own store path or by the chip's write bandwidth, and whether whole-line shapes are cheaper than 16 rows x 64 B.
//   build: hipcc --offload-arch=gfx950 -O2 tools/probe_store.hip -o tools/bin/probe_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// SHAPE 0: 16 rows x 64 B per instruction (the GEMM epilogue's shape), 1: 8 rows x 128 B, 2: 4 rows x 256 B, 3: 1 KiB contiguous
template <int SHAPE, int AUX>
__global__ __launch_bounds__(512) void store_kernel(uint8_t* out, long long bytes, int reps, int row_stride, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)bytes, 0x00020000);
  int voff;
  if (SHAPE == 0) voff = (lane & 15) * row_stride + (lane >> 4) * 16;
  else if (SHAPE == 1) voff = (lane >> 3) * row_stride + (lane & 7) * 16;
  else if (SHAPE == 2) voff = (lane >> 4) * row_stride + (lane & 15) * 16;
  else voff = lane * 16;
  const int rows_per = SHAPE == 0 ? 16 : SHAPE == 1 ? 8 : SHAPE == 2 ? 4 : 1;
  const int width = 1024 / rows_per;  // bytes per row per instruction
  // a "tile" = 256 rows x 512 B per workgroup (128 KiB); wave w owns rows 128*(w>>2).. and a 128-B column strip (w&3)
  v4u v = {(unsigned)lane, (unsigned)wave, 0x3f803f80u, 0x40004000u};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    const long long tile = ((long long)blockIdx.x * reps + r);
    const int tile_off = (int)((tile * 256 * (long long)row_stride) % (bytes - 256LL * row_stride - 4096));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // 16 instructions cover the wave's strip in pieces of rows_per rows x width bytes (128-B strips; 256-B strips for SHAPE 2)
      const int ppg = width >= 128 ? 1 : 128 / width;  // pieces per row group
      const int piece_row = (i / ppg) * rows_per, piece_col = (i % ppg) * width;
      int soff;
      if (SHAPE == 3) soff = tile_off + (wave * 16 + i) * 1024;
      else if (SHAPE == 2) soff = tile_off + ((wave >> 2) * 64 + i * 4) * row_stride + (wave & 3) * 256;
      else soff = tile_off + ((wave >> 2) * 128 + piece_row) * row_stride + (wave & 3) * 128 + piece_col;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, AUX);
      asm volatile("s_nop 1" ::"v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int AUX>
static void run(const char* name, uint8_t* buf, long long bytes, int grid, int reps, int row_stride, unsigned long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((store_kernel<SHAPE, AUX>), dim3(grid), dim3(512), 0, 0, buf, bytes, reps, row_stride, dcyc);
  hipEventRecord(e0);
  const int iters = 5;
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL((store_kernel<SHAPE, AUX>), dim3(grid), dim3(512), 0, 0, buf, bytes, reps, row_stride, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid);
  hipMemcpy(c.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto x : c) mean += (double)x;
  mean /= grid;
  const double tot = (double)grid * reps * 131072.0 * iters;
  printf("%-28s grid %3d stride %5d: %7.2f TB/s chip, %6.1f GB/s per CU, %6.1f cycles per 1-KiB store instruction per CU (%.1f B/clk/CU)\n", name, grid,
         row_stride, tot / (ms * 1e-3) / 1e12, tot / (ms * 1e-3) / 1e9 / grid, mean / (reps * 128.0), reps * 131072.0 / mean);
}

int main() {
  const long long bytes = 1LL << 30;  // 1 GiB target (beyond the Infinity Cache)
  uint8_t* buf;
  unsigned long long* dcyc;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&dcyc, 4096 * 8) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  for (int grid : {16, 64, 256}) {
    for (int stride : {6144, 16384}) {
      const int reps = 24;
      run<0, 16>("16 rows x 64 B, sc1", buf, bytes, grid, reps, stride, dcyc);
      run<0, 0>("16 rows x 64 B, plain", buf, bytes, grid, reps, stride, dcyc);
      run<0, 2>("16 rows x 64 B, nt", buf, bytes, grid, reps, stride, dcyc);
      run<1, 16>("8 rows x 128 B, sc1", buf, bytes, grid, reps, stride, dcyc);
      run<1, 0>("8 rows x 128 B, plain", buf, bytes, grid, reps, stride, dcyc);
      run<1, 2>("8 rows x 128 B, nt", buf, bytes, grid, reps, stride, dcyc);
      run<2, 16>("4 rows x 256 B, sc1", buf, bytes, grid, reps, stride, dcyc);
      run<3, 16>("1 KiB contiguous, sc1", buf, bytes, grid, reps, stride, dcyc);
      run<3, 0>("1 KiB contiguous, plain", buf, bytes, grid, reps, stride, dcyc);
    }
  }
  return 0;
}
