// What fits in the shadow of a wave's OWN MFMA (GPU box)?  One wave per SIMD issues v_mfma_scale_f32_16x16x128_f8f6f4 back to back
// (32 matrix-pipe cycles each) with a filler after every MFMA; the filler that leaves the cycles per MFMA at 32 is free.
// This decides how the four-wave GEMM (llm_fp8_amd/csrc/mi_gemm_w4.hip) spreads its epilogue: its chunks are made of exactly
// these instructions.  Fillers (independent of the MFMAs unless said otherwise):
//   0 none   1 4 v_mul_f32   2 8 v_mul_f32   3 4 v_accvgpr_read (AGPRs no MFMA of the loop touches)   4 4 v_accvgpr_read + 4 v_mul
//   5 2 v_pk_mul_f32   6 2 v_cvt_pk_bf16_f32   7 2 v_permlane16_swap   8 4 v_mov_b32 DPP row_ror:8   9 8 v_mov_b32 DPP
//   10 1 ds_read_b128   11 1 buffer_store_dwordx4 (one wave of four per slot)   12 full conversion chunk (4 read, 4 mul, 2 cvt)
//   13 4 v_accvgpr_read of an accumulator the PREVIOUS MFMA wrote (dependent: the interlock's worst case)
//   14 12 v_mul_f32   15 16 v_mul_f32
// hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_shadow.hip -o tools/bin/probe_mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define MFMA(acc) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(a), "v"(b), "v"(unit))

template <int F>
__global__ __launch_bounds__(256, 1) void k(const v4i* src, float* out, unsigned long long* clk, int iters, uint8_t* sink) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 1024; i += 256) reinterpret_cast<v4i*>(lds)[i] = src[i];
  __syncthreads();
  v8i a, b;
  {
    const v4i x = src[lane], y = src[64 + lane], z = src[128 + lane], w = src[192 + lane];
    a = (v8i){x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    b = (v8i){z[0], z[1], z[2], z[3], w[0], w[1], w[2], w[3]};
  }
  int unit = 0x7f7f7f7f;
  asm volatile("" : "+v"(unit));
  v4f acc[8], spare[4];
  for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { spare[i] = (v4f){1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+a"(spare[i])); }
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + i + lane;
  v4i d = {lane, 1, 2, 3};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, 1 << 26, 0x00020000);
  const unsigned ldsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + lane * 16;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      MFMA(acc[m]);
      if (F == 1 || F == 2 || F == 14 || F == 15 || F == 16 || F == 17) {
        constexpr int n = F == 1 ? 4 : F == 2 ? 8 : F == 14 ? 12 : F == 15 ? 16 : F == 16 ? 5 : 6;
#pragma unroll
        for (int e = 0; e < n; ++e) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[e]) : "v"(r[(e + 1) & 15]));
      } else if (F == 3 || F == 4 || F == 12) {
        float t0, t1, t2, t3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %4\n\tv_accvgpr_read_b32 %2, %4\n\tv_accvgpr_read_b32 %3, %4"
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "a"(spare[m & 3][0]));
        if (F == 4 || F == 12) {
          asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
                       : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(r[0]));
        }
        if (F == 12) {
          asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        }
        asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
      } else if (F == 13) {
        float t0, t1, t2, t3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %4\n\tv_accvgpr_read_b32 %2, %4\n\tv_accvgpr_read_b32 %3, %4"
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "a"(acc[(m + 7) & 7][0]));
        asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
      } else if (F == 5) {
        v2f p = {r[0], r[1]}, q = {r[2], r[3]}, s = {r[4], r[5]};
        asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %2" : "+v"(p), "+v"(q) : "v"(s));
        r[0] = p[0]; r[1] = p[1]; r[2] = q[0]; r[3] = q[1];
      } else if (F == 6) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
      } else if (F == 7) {
        asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
      } else if (F == 8 || F == 9) {
#pragma unroll
        for (int e = 0; e < (F == 8 ? 4 : 8); ++e)
          asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(d[e & 3]) : "v"(d[(e + 1) & 3]));
      } else if (F == 10) {
        v4i t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ldsaddr));
        asm volatile("s_waitcnt lgkmcnt(4)" ::"v"(t));
      } else if (F == 18) {  // 4 v_mul + 2 cvt_pk
        asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
                     "v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(r[4]));
      } else if (F == 19) {  // 2 cvt_pk + 2 permlane16_swap
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %2, %2, %3" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
        asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
      } else if (F == 20 || F == 21 || F == 22) {  // ONE store per 32 MFMA slots and workgroup (~ the real kernel's rate x 4)
        if (F == 21) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(d[e & 3]) : "v"(d[(e + 1) & 3]));
        }
        if (m == 0 && (it & 3) == wave) {
          if (F == 22) __builtin_amdgcn_raw_buffer_store_b64((__attribute__((ext_vector_type(2))) unsigned){(unsigned)d[0], (unsigned)d[1]}, rs, lane * 8, ((blockIdx.x * 64 + (it & 63)) * 1024) & ((1 << 26) - 1), 16);
          else __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, ((blockIdx.x * 64 + (it & 63)) * 1024) & ((1 << 26) - 1), 16);
        }
      } else if (F >= 30 && F < 40) {  // store issue cost: one store per 32 (F < 35) or 128 (F >= 35) MFMA slots and WAVE; policy by F % 5
        constexpr int every = F < 35 ? 4 : 16;
        constexpr int pol = (F % 5) == 0 ? 16 : (F % 5) == 1 ? 0 : (F % 5) == 2 ? 2 : (F % 5) == 3 ? 18 : 1;
        if (m == 0 && (it % every) == (wave % every)) {
          __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, ((blockIdx.x * 64 + (it & 63)) * 1024) & ((1 << 26) - 1), pol);
        }
      } else if (F == 40 || F == 41 || F == 42) {
        // 40: every wave executes a store in EVERY 4th slot, all with EXEC = 0 (what does a masked-off store cost?)
        // 41: every wave executes a store in every 4th slot; in slot m only wave (m / 4 + it) % 4 has EXEC != 0: ONE real store
        //     per 4 slots and workgroup, the other three waves' stores are masked off
        // 42: the lockstep reference: all four waves store (EXEC full) in the same slot, one slot in 16 (same bytes as 41)
        if ((m & 3) == 0) {
          const bool mine = F == 41 ? (((m >> 2) + it) & 3) == wave : false;
          if (F == 42) {
            if (m == 0 && (it & 1) == 0)
              __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, ((blockIdx.x * 64 + (it & 63)) * 4096 + wave * 1024) & ((1 << 26) - 1), 16);
          } else {
            const int mask = __builtin_amdgcn_readfirstlane(mine ? -1 : 0);
            const int soff = ((blockIdx.x * 64 + (it & 63)) * 4096 + wave * 1024) & ((1 << 26) - 1);
            asm volatile("s_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %0" ::"s"(mask) : "memory");
            __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, soff, 16);
            asm volatile("s_mov_b64 exec, -1" ::: "memory");
          }
        }
      } else if (F >= 50 && F < 60) {
        // store-cost curve without branches: every wave executes ONE store per 8 slots (m == 0), real (EXEC full) every P-th
        // iteration, masked off otherwise; lockstep (all four waves in the same slot).  F = 50 + log2(P); 59: never real
        constexpr int P = (F == 59 || F < 50) ? 0 : 1 << ((F >= 50 && F < 59) ? F - 50 : 0);
        if (m == 0) {
          const int mask = __builtin_amdgcn_readfirstlane((P != 0 && (it & (P - 1)) == 0) ? -1 : 0);
          const int soff = ((blockIdx.x * 64 + (it & 63)) * 4096 + wave * 1024) & ((1 << 26) - 1);
          asm volatile("s_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %0" ::"s"(mask) : "memory");
          __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, soff, 16);
          asm volatile("s_mov_b64 exec, -1" ::: "memory");
        }
      } else if (F == 11) {
        if (wave == (m & 3)) {
          __builtin_amdgcn_raw_buffer_store_b128((__attribute__((ext_vector_type(4))) unsigned)d, rs, lane * 16, ((blockIdx.x * 8 + m) * 1024) & ((1 << 26) - 1), 16);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0;
  for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) sum += r[i];
  sum += (float)(d[0] + d[1] + d[2] + d[3]);
  out[blockIdx.x * 256 + tid] = sum;
  if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int F>
static void run(const char* name, v4i* src, float* out, unsigned long long* clk, uint8_t* sink) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<F>, 256, 256, 0, 0, src, out, clk, iters, sink);
  (void)hipDeviceSynchronize();
  unsigned long long hc[512];
  (void)hipMemcpy(hc, clk, 4096, hipMemcpyDeviceToHost);
  printf("filler %2d %-58s cycles per MFMA %6.1f   clock %4.0f MHz\n", F, name, (double)hc[0] / (iters * 8.0), (double)hc[0] / hc[1] * 100.0);
}

int main() {
  v4i* src; float* out; unsigned long long* clk; uint8_t* sink;
  (void)hipMalloc(&src, 65536); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&clk, 256 * 16); (void)hipMalloc(&sink, 1 << 26);
  unsigned char* h = (unsigned char*)malloc(65536);
  for (int i = 0; i < 65536; ++i) { unsigned char v = rand() & 0xff; if ((v & 0x7f) >= 0x78) v &= 0x3f; h[i] = v; }
  (void)hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
  run<0>("none", src, out, clk, sink);
  run<1>("4 v_mul_f32", src, out, clk, sink);
  run<2>("8 v_mul_f32", src, out, clk, sink);
  run<14>("12 v_mul_f32", src, out, clk, sink);
  run<15>("16 v_mul_f32", src, out, clk, sink);
  run<3>("4 v_accvgpr_read (untouched AGPRs)", src, out, clk, sink);
  run<13>("4 v_accvgpr_read (accumulator of the previous MFMA)", src, out, clk, sink);
  run<4>("4 v_accvgpr_read + 4 v_mul_f32", src, out, clk, sink);
  run<12>("conversion chunk: 4 read + 4 mul + 2 cvt_pk_bf16", src, out, clk, sink);
  run<5>("2 v_pk_mul_f32", src, out, clk, sink);
  run<6>("2 v_cvt_pk_bf16_f32", src, out, clk, sink);
  run<7>("2 v_permlane16_swap_b32", src, out, clk, sink);
  run<8>("4 v_mov_b32 DPP row_ror:8", src, out, clk, sink);
  run<9>("8 v_mov_b32 DPP row_ror:8", src, out, clk, sink);
  run<10>("1 ds_read_b128", src, out, clk, sink);
  run<11>("1 buffer_store_dwordx4 sc1 per slot and workgroup", src, out, clk, sink);
  run<16>("5 v_mul_f32", src, out, clk, sink);
  run<17>("6 v_mul_f32", src, out, clk, sink);
  run<18>("4 v_mul_f32 + 2 v_cvt_pk_bf16_f32", src, out, clk, sink);
  run<19>("2 v_cvt_pk_bf16_f32 + 2 v_permlane16_swap", src, out, clk, sink);
  run<20>("1 buffer_store_dwordx4 sc1 per 32 slots and workgroup", src, out, clk, sink);
  run<21>("... + 4 DPP in every slot", src, out, clk, sink);
  run<22>("1 buffer_store_dwordx2 (half lines) per 32 slots", src, out, clk, sink);
  run<30>("store per 32 slots and wave, sc1", src, out, clk, sink);
  run<31>("store per 32 slots and wave, plain (write-back)", src, out, clk, sink);
  run<32>("store per 32 slots and wave, nt", src, out, clk, sink);
  run<33>("store per 32 slots and wave, sc1 nt", src, out, clk, sink);
  run<34>("store per 32 slots and wave, sc0", src, out, clk, sink);
  run<35>("store per 128 slots and wave, sc1", src, out, clk, sink);
  run<36>("store per 128 slots and wave, plain", src, out, clk, sink);
  run<37>("store per 128 slots and wave, nt", src, out, clk, sink);
  run<40>("store with EXEC = 0 every 4th slot (all waves)", src, out, clk, sink);
  run<41>("store every 4th slot, one wave real, three masked off", src, out, clk, sink);
  run<42>("lockstep: all four waves store in one slot of 16", src, out, clk, sink);
  run<59>("masked-off store every 8 slots (baseline of the curve)", src, out, clk, sink);
  run<50>("lockstep store every 8 slots", src, out, clk, sink);
  run<51>("lockstep store every 16 slots", src, out, clk, sink);
  run<52>("lockstep store every 32 slots", src, out, clk, sink);
  run<53>("lockstep store every 64 slots", src, out, clk, sink);
  run<54>("lockstep store every 128 slots", src, out, clk, sink);
  run<55>("lockstep store every 256 slots", src, out, clk, sink);
  return 0;
}
