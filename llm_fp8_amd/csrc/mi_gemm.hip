// K4/K5/K6 (+K8): FP8 x FP8 -> bf16 "TN" GEMM on gfx950 MFMA.
//
//   D[M,N] = (A[M,K] . B[N,K]^T) * alpha (+ bias),  A/B fp8 (E4M3 or E5M2 each), fp32 accumulate.
//
// Instruction: v_mfma_scale_f32_16x16x128_f8f6f4 -- the only FP8 form that reaches the ~5 PFLOP/s
// dense peak (the MI300-era 16x16x32_fp8_fp8 runs at the bf16 rate).  Per-tensor (delayed) scaling
// passes unit E8M0 scales (0x7F) and applies alpha = sa_inv*sb_inv in the epilogue; MXFP8 passes the
// real per-32 E8M0 block scales.  Operand lane map (measured on MI355X with tools/probe_mfma.hip):
// lane l = (r = l & 15, q = l >> 4) holds row r; its registers 0-3 are K-bytes [16q, 16q+16) and
// registers 4-7 are K-bytes [64+16q, 64+16q+16) of the 128-deep step (two 16-B LDS reads, chunks q
// and 4+q of the 128-B row).  The E8M0 scale of lane (r, j) applies to row r, K-bytes [32j, 32j+32).
// Result: we issue mfma(B, A) so that a lane holds 4 consecutive n for one m
// (lane & 15 -> m, (lane >> 4)*4 + reg -> n).
// Accumulation (measured): products are summed in groups of 8 aligned to the group's largest
// product; a product more than 2^13 below it is dropped, so |err| <= 7*2^-14 * sum|a.b| worst case;
// across MFMA instructions the accumulator is plain fp32.
//
// Kernels
//   gemm_generic   64x64 tile, register-staged, fully predicated: any M,N (mult. of 8), K mult. of 16.
//   gemm_256_2ph   256x256x128 tile, 8 waves (2M x 4N, 128x64 per wave), direct-to-LDS staging
//                  (global_load_lds_dwordx4), 2 LDS buffers, one barrier per K-step.
// LDS image (both fast kernels): per operand tile 256 rows x 128 B, cut into 1-KiB pieces of
// 8 rows x 128 B -- exactly what one wave-wide global_load_lds_dwordx4 writes, and each row a full
// 128-B line of the source.  The 16-B chunk c of row r sits at chunk position c ^ f(r),
// f(r) = ((r>>1)&3)<<1: with it the two ds_read_b128 of a fragment (chunks q and 4+q of rows
// r = 0..15) hit 16 distinct 16-B bank slots in every 16-lane service group.
// The swizzle is applied on the global SOURCE address (LDS-DMA writes lane-linear) and again
// on the read address.
//
// Replaces the cuBLASLt FP8 GEMMs behind te.Linear / LayerNormLinear / LayerNormMLP on the
// reference path (te_llama.py:45-63,76-80; Appendix B shapes).
#include "mi_gemm_dev.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace mi {

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the kernels: lane holds acc[j] for (m, n0 + j), j = 0..3.
template <int OUT>
__device__ __forceinline__ void store4(void* D, int64_t ldd, int64_t m, int64_t n0, v4f v) {
  if (OUT == 0) {
    uint2 p = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(D) + m * ldd + n0) = p;
  } else {
    *reinterpret_cast<v4f*>(reinterpret_cast<float*>(D) + m * ldd + n0) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Generic path.  64x64 tile, 4 waves (2x2), each wave 32x32 = 2x2 MFMA tiles, BK = 128.
// LDS: A[64][128] + B[64][128] bytes, chunk position = chunk ^ (row & 7).
template <int FA, int FB, int OUT, bool MX>
__global__ __launch_bounds__(256) void gemm_generic(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                    void* __restrict__ D, const float* __restrict__ sa_inv,
                                                    const float* __restrict__ sb_inv,
                                                    const uint8_t* __restrict__ SA, const uint8_t* __restrict__ SB,
                                                    const uint16_t* __restrict__ bias, int M, int N, int K,
                                                    int64_t lda, int64_t ldb, int64_t ldd) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * 64 * 128];
  uint8_t* lA = lds;
  uint8_t* lB = lds + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int fr = lane & 15, fq = lane >> 4;
  v4f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int kb = K / 32;  // scale blocks per row (MX)
  for (int k0 = 0; k0 < K; k0 += 128) {
    // stage: 512 chunks per operand, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ch = tid + i * 256;
      int row = ch >> 3, c = ch & 7;
      v4i va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
      if (m0 + row < M && k0 + c * 16 < K) va = *reinterpret_cast<const v4i*>(A + (int64_t)(m0 + row) * lda + k0 + c * 16);
      if (n0 + row < N && k0 + c * 16 < K) vb = *reinterpret_cast<const v4i*>(B + (int64_t)(n0 + row) * ldb + k0 + c * 16);
      *reinterpret_cast<v4i*>(lA + row * 128 + ((c ^ (row & 7)) * 16)) = va;
      *reinterpret_cast<v4i*>(lB + row * 128 + ((c ^ (row & 7)) * 16)) = vb;
    }
    __syncthreads();
    v8i af[2], bf[2];
    int sa[2] = {kUnitScale, kUnitScale}, sb[2] = {kUnitScale, kUnitScale};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int ra = wr * 32 + t * 16 + fr, rb = wc * 32 + t * 16 + fr;
      v4i a0 = *reinterpret_cast<const v4i*>(lA + ra * 128 + ((fq ^ (ra & 7)) * 16));
      v4i a1 = *reinterpret_cast<const v4i*>(lA + ra * 128 + (((4 + fq) ^ (ra & 7)) * 16));
      v4i b0 = *reinterpret_cast<const v4i*>(lB + rb * 128 + ((fq ^ (rb & 7)) * 16));
      v4i b1 = *reinterpret_cast<const v4i*>(lB + rb * 128 + (((4 + fq) ^ (rb & 7)) * 16));
      af[t] = (v8i){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      bf[t] = (v8i){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      if (MX) {
        int blk = k0 / 32 + fq;
        // out-of-range rows / k-blocks multiply zero data: any finite scale works, use 2^0
        sa[t] = (m0 + ra < M && blk < kb) ? (int)SA[(int64_t)blk * M + m0 + ra] : 0x7F;  // block-major [K/32, M]
        sb[t] = (n0 + rb < N && blk < kb) ? (int)SB[(int64_t)blk * N + n0 + rb] : 0x7F;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma_ba<FA, FB>(af[i], bf[j], acc[i][j], sa[i], sb[j]);
    __syncthreads();
  }
  float alpha = 1.0f;
  if (!MX) alpha = (*sa_inv) * (*sb_inv);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int m = m0 + wr * 32 + i * 16 + fr;
      int n = n0 + wc * 32 + j * 16 + fq * 4;
      if (m < M && n < N) {  // N is a multiple of 4 (checked on the host): the 4-wide group is all-in or all-out
        v4f v = acc[i][j] * alpha;
        if (bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf16_bits_to_float(bias[n + e]);
        }
        store4<OUT>(D, ldd, m, n, v);
      }
    }
}

// Stage one operand tile (256 rows x 128 B) of K-step kt into LDS at `lds_tile`.
// wave w issues 4 x global_load_lds_dwordx4, pieces s = 4w .. 4w+3 (8 rows each).
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ g_tile_row0, int64_t ld, int k_byte,
                                           uint8_t* lds_tile, int wave, int lane) {
  const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int src_chunk = lc ^ swz_f(lr);
    const uint8_t* src = g_tile_row0 + (int64_t)(piece * 8 + lr) * ld + k_byte + src_chunk * 16;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <int FA, int FB, int OUT>
__global__ __launch_bounds__(512, 2) void gemm_256_2ph(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                       void* __restrict__ D, const float* __restrict__ sa_inv,
                                                       const float* __restrict__ sb_inv,
                                                       const uint16_t* __restrict__ bias, int M, int N, int K,
                                                       int64_t lda, int64_t ldb, int64_t ldd, int tiles_m,
                                                       int tiles_n) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // 2 (M) x 4 (N)
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const uint8_t* gA = A + (int64_t)tm * BM * lda;
  const uint8_t* gB = B + (int64_t)tn * BN * ldb;
  v4f acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
  stage_tile(gA, lda, 0, lds, wave, lane);
  stage_tile(gB, ldb, 0, lds + kTileBytes, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    uint8_t* cur = lds + (kt & 1) * kBufBytes;
    uint8_t* nxt = lds + ((kt + 1) & 1) * kBufBytes;
    if (kt + 1 < nk) {
      stage_tile(gA, lda, (kt + 1) * BK, nxt, wave, lane);
      stage_tile(gB, ldb, (kt + 1) * BK, nxt + kTileBytes, wave, lane);
    }
    v8i bf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = read_frag(cur + kTileBytes, wc * 4 + j, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v8i af = read_frag(cur, wr * 8 + i, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_ba<FA, FB>(af, bf[j], acc[i][j], kUnitScale, kUnitScale);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const float alpha = (*sa_inv) * (*sb_inv);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t n = (int64_t)tn * BN + wc * 64 + j * 16 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bf16_bits_to_float(bias[n + e]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t m = (int64_t)tm * BM + wr * 128 + i * 16 + fr;
      v4f v = acc[i][j] * alpha;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bv[e];
      store4<OUT>(D, ldd, m, n, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_256_8ph: 256x256x128 tile, 8 waves, two wave groups (waves 0-3 / 4-7 = the two waves of each
// SIMD) running half a phase apart ("ping-pong"): while one group issues its 8 MFMAs of a phase the
// other reads fragments from LDS and issues LDS-DMA prefetches.  Four phases per K-tile:
//     p0: read A0 (4 frags) + B0 (2)   mfma C[0][*][0][*]      stage (t+1, B1)
//     p1: read B1 (2)                  mfma C[0][*][1][*]      stage (t+1, A1)
//     p2: read A1 (4)                  mfma C[1][*][1][*]      stage (t+2, A0)
//     p3: --                           mfma C[1][*][0][*]      stage (t+2, B0)
// LDS: 2 K-tile buffers x {A0, A1, B0, B1} half-tiles (128 rows x 128 B = 16 pieces, 2 per wave).
// A wave's 128x64 output is 2x2 blocks of 64x32: rows 128*mh + 64*wr + [0,64), cols 128*nh + 32*wc + [0,32),
// so every half-tile is read in exactly ONE phase by all waves and can be restaged two phases later:
// each LDS-DMA has >= 5 phases (~2.5k cycles) to land.  Hazards (barrier slots, group 1 one slot late):
//   RAW  a half-tile first read in phase p+1 is retired by `s_waitcnt vmcnt(8)` (4 younger half-tile
//        stages may stay in flight) before the barrier that closes the load segment of phase p;
//   WAR  a half-tile read in phase p (both groups: slots 2p, 2p+1, retired by lgkmcnt(0) at the start
//        of the following slot) is restaged no earlier than phase p+2 (slot 2p+4).
// vmcnt is never drained inside the loop; barriers are raw s_barrier (a __syncthreads would drain it).
// Past the last K-tile the stages re-fetch tile nk-1 into a dead buffer so the counts stay uniform.
__device__ __forceinline__ void stage_half(const uint8_t* __restrict__ g_row0, int64_t ld, int k_byte, uint8_t* lds_half,
                                           int wave, int lane) {
  const int lr = lane >> 3, lc = lane & 7;
  const int src_chunk = lc ^ swz_f(lr);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;
    const uint8_t* src = g_row0 + (int64_t)(piece * 8 + lr) * ld + k_byte + src_chunk * 16;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_half + piece * 1024), 16, 0, 0);
  }
}

template <int FA, int FB, int OUT, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_256_8ph(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                       void* __restrict__ D, const float* __restrict__ sa_inv,
                                                       const float* __restrict__ sb_inv,
                                                       const uint16_t* __restrict__ bias, int M, int N, int K,
                                                       int64_t lda, int64_t ldb, int64_t ldd, int tiles_m,
                                                       int tiles_n) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const uint8_t* gA0 = A + (int64_t)tm * BM * lda;  // A half h starts 128*h rows further
  const uint8_t* gB0 = B + (int64_t)tn * BN * ldb;
  const uint8_t* gA1 = gA0 + 128 * lda;
  const uint8_t* gB1 = gB0 + 128 * ldb;
  v4f acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
  const int last = nk - 1;
  uint8_t* buf0 = lds;
  uint8_t* buf1 = lds + kBufBytes;
  // prologue: K-tile 0 complete + (1, A0), (1, B0) in flight
  stage_half(gA0, lda, 0, buf0 + kOffA0, wave, lane);
  stage_half(gB0, ldb, 0, buf0 + kOffB0, wave, lane);
  stage_half(gB1, ldb, 0, buf0 + kOffB1, wave, lane);
  stage_half(gA1, lda, 0, buf0 + kOffA1, wave, lane);
  {
    const int k1 = min(1, last) * BK;
    stage_half(gA0, lda, k1, buf1 + kOffA0, wave, lane);
    stage_half(gB0, ldb, k1, buf1 + kOffB0, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier slot behind group 0

  v8i af[4], b0f[2], b1f[2];
  auto ktile = [&](uint8_t* cur, uint8_t* oth, int t) {
    const int kb1 = min(t + 1, last) * BK, kb2 = min(t + 2, last) * BK;
    // ---- phase 0
#pragma unroll
    for (int j = 0; j < 2; ++j) b0f[j] = read_frag(cur + kOffB0, wc * 2 + j, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = read_frag(cur + kOffA0, wr * 4 + i, lane);
    stage_half(gB1, ldb, kb1, oth + kOffB1, wave, lane);
    MI_PHASE_SYNC_BEFORE_MFMA();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[0][i][0][j] = mfma_ba<FA, FB>(af[i], b0f[j], acc[0][i][0][j], kUnitScale, kUnitScale);
    MI_PHASE_END();
    // ---- phase 1
#pragma unroll
    for (int j = 0; j < 2; ++j) b1f[j] = read_frag(cur + kOffB1, wc * 2 + j, lane);
    stage_half(gA1, lda, kb1, oth + kOffA1, wave, lane);
    MI_PHASE_SYNC_BEFORE_MFMA();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[0][i][1][j] = mfma_ba<FA, FB>(af[i], b1f[j], acc[0][i][1][j], kUnitScale, kUnitScale);
    MI_PHASE_END();
    // ---- phase 2
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = read_frag(cur + kOffA1, wr * 4 + i, lane);
    stage_half(gA0, lda, kb2, cur + kOffA0, wave, lane);
    MI_PHASE_SYNC_BEFORE_MFMA();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[1][i][1][j] = mfma_ba<FA, FB>(af[i], b1f[j], acc[1][i][1][j], kUnitScale, kUnitScale);
    MI_PHASE_END();
    // ---- phase 3
    stage_half(gB0, ldb, kb2, cur + kOffB0, wave, lane);
    MI_PHASE_SYNC_BEFORE_MFMA();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[1][i][0][j] = mfma_ba<FA, FB>(af[i], b0f[j], acc[1][i][0][j], kUnitScale, kUnitScale);
    MI_PHASE_END();
  };
  unsigned long long c0 = 0, r0 = 0;
  if (ABL == 2) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  for (int t = 0; t < nk; t += 2) {
    ktile(buf0, buf1, t);
    if (t + 1 < nk) ktile(buf1, buf0, t + 1);
  }
  if (ABL == 2) {
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* dbg = (unsigned long long*)bias + (size_t)blockIdx.x * 2;
      dbg[0] = c1 - c0;
      dbg[1] = r1 - r0;
    }
    bias = nullptr;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the dead tail prefetches before the LDS is released

  const float alpha = (*sa_inv) * (*sb_inv);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = (int64_t)tn * BN + 128 * b + wc * 32 + j * 16 + fq * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bf16_bits_to_float(bias[n + e]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t m = (int64_t)tm * BM + 128 * a + wr * 64 + i * 16 + fr;
          v4f v = acc[a][i][b][j] * alpha;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
          if (ABL >= 1) {  // timing ablation: no stores (results are garbage)
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          } else {
            store4<OUT>(D, ldd, m, n, v);
          }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_256_p8: persistent form of the 8-phase kernel.  One workgroup per CU walks its list of output
// tiles; the (tile, K-tile) steps form ONE continuous pipeline, so the LDS-DMA prefetch of the next
// tile's first K-tiles is already in flight while the current tile's accumulators are converted and
// stored, and the stores drain behind the next tile's MFMAs (they are never waited for: the four waits
// after an epilogue use vmcnt(8 + 16 stores)).
// Ownership is contiguous per wave (the half-tiles are row GATHERS, free with per-lane LDS-DMA sources):
//   wave (wr, wc) owns rows 128*wr + 64*mh + [0,64) and columns 64*wc + 32*nh + [0,32), i.e. a 128x64
//   block whose rows are whole 128-B lines of the bf16 output.  In the epilogue the two 4-column pieces
//   a lane holds per (mh, i, nh) are widened to 8 contiguous columns with v_permlane16_swap and stored
//   as one dwordx4 (16 rows x 64 B per instruction, 16 stores per wave and tile).
// Tile shapes: a wave owns (4 + MA1) x (2 + NB1) MFMA tiles; MA1 in {4, 2} and NB1 in {2, 1} give workgroup tiles of
// 256 or 192 rows x 256 or 192 columns.  The 192 variants exist for wave quantisation: an 8192x3072 output is 384
// tiles of 256x256 = 1.5 rounds on 256 CUs (the second round half empty) but 512 tiles of 256x192 = exactly 2 rounds of
// 0.75-size tiles.  Same 4-phase pipeline; phases 1-2 (and 2-3) just issue fewer MFMAs and the second half-tiles
// (A1: 2*16*MA1 rows, B1: 4*16*NB1 rows) are smaller.

// Requires K % 256 == 0 (an even number of K-tiles, so every tile starts on LDS buffer 0) and
// operand / output footprints < 2^31 bytes (32-bit buffer offsets); the host dispatcher checks both.
//
// SK (stream-K): when the tile count is not a multiple of the CU count, the flattened (tile, K-tile) steps are cut into
// equal contiguous ranges, one per workgroup (range length U >= K-tiles per tile, even).  A range may START inside a tile:
// that tail part is computed FIRST, its fp32 accumulators go to workspace slot v (write-through stores, vmcnt(0), barrier,
// then one lane publishes flags[v] = epoch).  A range may END inside a tile: that head part is computed LAST; the
// workgroup then waits for flags[v + 1] == epoch (set long before, the neighbour produced it first thing), adds the
// neighbour's partial accumulators and runs the normal epilogue.  Producers never wait, so there is no cycle.
template <int FA, int FB, int ABL = 0, bool MX = false, bool BIAS = false, int MA1 = 4, int NB1 = 2, bool SK = false, bool DEPI = true>
__global__ __launch_bounds__(512, 2) void gemm_256_p8(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                      uint16_t* __restrict__ D, const float* __restrict__ sa_inv,
                                                      const float* __restrict__ sb_inv, int K, int lda, int ldb,
                                                      int ldd, int tiles_m, int tiles_n, int a_bytes, int b_bytes,
                                                      int d_bytes, const uint8_t* __restrict__ SA,
                                                      const uint8_t* __restrict__ SB, int M, int N,
                                                      const uint16_t* __restrict__ bias, float* __restrict__ sk_ws,
                                                      unsigned int* __restrict__ sk_flags, unsigned int sk_epoch, int sk_U) {
  constexpr int RA0 = 64, RA1 = 16 * MA1, RB0 = 32, RB1 = 16 * NB1;  // rows of A / B per wave in half 0 / 1
  constexpr int TBM = 2 * (RA0 + RA1), TBN = 4 * (RB0 + RB1);        // workgroup tile
  constexpr int nA1 = MA1 / 2, nB1 = NB1;                            // LDS-DMA pieces per wave of the second halves
  // extra LDS-DMA ops per K-tile at the head of phase 0: the bias window and, for MX, one 256-byte run of block scales per
  // wave (operand w>>2, k-block w&3) for the NEXT step; phase 3's wait retires them (see MI_WAIT_SYNC).
  // epilogue store policy (buffer aux bits: 1 = sc0, 2 = nt, 16 = sc1); ABL 4 / 12 / 13 are the timing A/B builds (algo 17 / 25 / 26)
  constexpr int kStoreAux = ABL == 4 ? 0 : ABL == 12 ? 2 : ABL == 13 ? 18 : 16;
  constexpr int EX = (MX ? 1 : 0) + (BIAS ? 1 : 0);
  constexpr int W = nA1 + nB1 + 4;                                   // younger ops allowed at the p3 wait (8 for 256x256)
  constexpr int NST = (4 + MA1) * 2;                                 // epilogue stores per wave and tile
  // MX: the A fragments of a half are ROW-INTERLEAVED (fragment i, lane row-slot s <-> row F*s + i of the half, F = its
  // fragment count) instead of blocked (16*i + s): the F block scales a lane needs per k-block are then F consecutive bytes,
  // one LDS read instead of F byte reads (12 ds_read_u8 per K-tile cost the block-scaled GEMM 10 %).  Free on both sides:
  // the rows are gathered by the LDS-DMA source addresses, and an epilogue store covers 16 rows either way.
  constexpr bool PERM = MX;
  // behind the operand buffers: 8 KiB for the E8M0 scales (MX: 2 slots of 4 KiB x {A, B} x 4 k-blocks x 256 rows, padded) and
  // 4 KiB of bias windows (BIAS: 2 slots x 8 waves x 256 B)
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes + ((MX || BIAS) ? 12288 + 256 : 0) + (ABL == 9 ? 16384 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // ABL == 9 (diagnostic build, algo 22): waves 0 and 4 of workgroup 0 stamp s_memtime at the start of every MFMA segment and at
  // the end of every phase into LDS (2 x 1024 stamps) and dump them to the u64 buffer passed as `bias` when the kernel ends
  int stamp_idx = 0;
  unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + kLdsBytes + ((MX || BIAS) ? 12288 + 256 : 0)) + (wave >> 2) * 1024;
  auto stamp = [&]() {
    if (ABL == 9) {
      if (blockIdx.x == 0 && (wave & 3) == 0 && stamp_idx < 1024) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamp_lds[stamp_idx] = t;
        ++stamp_idx;
      }
    }
  };
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int nk = K / BK;
  // SK: virtual index v = XCD-major order of the workgroups (neighbours in v share an L2), range [v U, (v + 1) U)
  int sk_v = 0, sk_t0 = 0, sk_k0 = 0, sk_total = 0;
  if (SK) {
    const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
    sk_v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int u0 = sk_v * sk_U, u1 = min(u0 + sk_U, ntiles * nk);
    sk_t0 = u0 / nk;
    sk_k0 = u0 - sk_t0 * nk;
    sk_total = u1 - u0;
  }
  const int my_tiles = SK ? 0 : (ntiles - bid + G - 1) / G;  // tiles bid, bid + G, ...
  const float alpha = MX ? 1.0f : (*sa_inv) * (*sb_inv);
  const int total_tiles_hint = SK ? (sk_total + sk_k0 + nk - 1) / nk : my_tiles;  // tiles this workgroup touches
  const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
  const rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)D, 0, d_bytes, 0x00020000);
  // MX scales are block-major [K/32, rows]; wave w stages the 256 scale bytes of (operand w>>2, k-block w&3)
  const bool s_is_b = wave >= 4;
  const int s_rows = s_is_b ? N : M;
  const rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(MX ? (s_is_b ? SB : SA) : A), 0, MX ? (K / 32) * s_rows : 0, 0x00020000);
  // [slot][operand][4 k-blocks][256 rows], the k-block rows 272 B apart: the four 16-lane groups of a scale read (k-blocks
  // 0..3 of rows r..r+15) then fall on disjoint banks; at a 256-B stride they were a 4-way conflict on an LDS that the
  // fragment reads already keep 75 % busy, and the block-scaled GEMM ran 10 % behind the per-tensor one
  // LDS: [2 slots][operand A, B][4 k-blocks][272]
  constexpr int kSK = 272, kSOp = 4 * kSK, kSSlot = 4096;  // slot stride: a power of two, so the read bases toggle by v_xor (2 * kSOp = 2176 B used)
  uint8_t* const sbuf = lds + kLdsBytes;
  auto stage_scales = [&](int slot, int kt, int row0a, int row0b) {
    if (MX) {
      const int soff = (kt * 4 + (wave & 3)) * s_rows + (s_is_b ? row0b : row0a);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, LDS_PTR(sbuf + slot * kSSlot + (wave >> 2) * kSOp + (wave & 3) * kSK), 4,
                                               lane * 4, soff, 0, 0);
    }
  };
  // bias: every K-tile each wave re-fetches the 256-byte window that starts at its own columns of cursor 1's tile
  // (one dword per lane; reads past N return 0 through the descriptor's range check) -> uniform vmcnt accounting
  const rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)(BIAS ? (const void*)bias : (const void*)A), 0, BIAS ? N * 2 : 0, 0x00020000);
  uint8_t* const bbuf = lds + kLdsBytes + 8192;  // [slot][wave][256 B]
  auto stage_bias = [&](int slot, int col0) {
    if (BIAS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(bbuf + (slot * 8 + wave) * 256), 4, lane * 4,
                                               (col0 + wc * (RB0 + RB1)) * 2, 0, 0);
  };
  v4f acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  // per-lane staging offsets.  Half-tile local row -> tile row: the rows of wave-row wr' (A) / wave-col wc' (B) of
  // half h are contiguous in the tile, so a wave's output block is contiguous (whole 128-B lines per row).
  // ONE per-lane offset per operand (plus one for a second half whose row map differs: MA1 == 2 / NB1 == 1); the second
  // 8-row piece of a wave and the second half of the tile are wave-uniform distances from it and ride in the SGPR offset of
  // the buffer load.  (Eight separate per-lane offsets pushed the 256x256 variants over the 256-VGPR limit: a spilled
  // fragment base was reloaded behind `s_waitcnt vmcnt(0)` once per tile, draining the prefetch pipeline.)
  int a0_v, b0_v, a1_v = 0, b1_v = 0;
  {
    const int lr = lane >> 3, lc = lane & 7;
    const int chunk = (lc ^ swz_f(lr)) * 16;
    {
      const int local = wave * 16 + lr;  // half 0: 128 rows, 16 pieces; piece 1 of the wave = 8 local rows further
      const int l0 = local % RA0;
      a0_v = ((local / RA0) * (RA0 + RA1) + (PERM ? 4 * (l0 & 15) + (l0 >> 4) : l0)) * lda + chunk;
      b0_v = ((local / RB0) * (RB0 + RB1) + local % RB0) * ldb + chunk;
    }
    if (MA1 != 4) {
      const int local = wave * 8 + lr;  // half 1 of A: 2*RA1 rows, one piece per wave
      const int l1 = local % RA1;
      a1_v = ((local / RA1) * (RA0 + RA1) + RA0 + (PERM ? MA1 * (l1 & 15) + (l1 >> 4) : l1)) * lda + chunk;
    }
    if (NB1 != 2) {
      const int local = wave * 8 + lr;  // half 1 of B: 4*RB1 rows, one piece per wave
      b1_v = ((local / RB1) * (RB0 + RB1) + RB0 + local % RB1) * ldb + chunk;
    }
  }
  const int stepA = (PERM ? 32 : 8) * lda, stepB = 8 * ldb;  // local row + 8 -> tile row + 8 (PERM: fragment row slot + 8 = 32 rows)
  auto dma = [&](rsrc_t rs, uint8_t* dst, int voff, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, voff, soff, 0, 0);
  };
  auto stage_a0 = [&](int soff, uint8_t* buf) __attribute__((always_inline)) {
    dma(rsA, buf + kOffA0 + (wave * 2) * 1024, a0_v, soff);
    dma(rsA, buf + kOffA0 + (wave * 2 + 1) * 1024, a0_v, soff + stepA);
  };
  auto stage_b0 = [&](int soff, uint8_t* buf) __attribute__((always_inline)) {
    dma(rsB, buf + kOffB0 + (wave * 2) * 1024, b0_v, soff);
    dma(rsB, buf + kOffB0 + (wave * 2 + 1) * 1024, b0_v, soff + stepB);
  };
  auto stage_a1 = [&](int soff, uint8_t* buf) __attribute__((always_inline)) {
    if (MA1 == 4) {  // same row map as half 0, RA0 tile rows further
      dma(rsA, buf + kOffA1 + (wave * 2) * 1024, a0_v, soff + RA0 * lda);
      dma(rsA, buf + kOffA1 + (wave * 2 + 1) * 1024, a0_v, soff + RA0 * lda + stepA);
    } else {
      dma(rsA, buf + kOffA1 + wave * 1024, a1_v, soff);
    }
  };
  auto stage_b1 = [&](int soff, uint8_t* buf) __attribute__((always_inline)) {
    if (NB1 == 2) {
      dma(rsB, buf + kOffB1 + (wave * 2) * 1024, b0_v, soff + RB0 * ldb);
      dma(rsB, buf + kOffB1 + (wave * 2 + 1) * 1024, b0_v, soff + RB0 * ldb + stepB);
    } else {
      dma(rsB, buf + kOffB1 + wave * 1024, b1_v, soff);
    }
  };

  // pipeline cursors (wave-uniform byte offsets of the (tile, K-tile) of step s+1 / s+2, clamped to the last step)
  const int total = SK ? sk_total : my_tiles * nk;
  auto tile_mn = [&](int ti, int& tm, int& tn) {
    if (SK) tile_of_flat(sk_t0 + ti, tiles_m, tiles_n, tm, tn);
    else tile_of_block(bid + ti * G, ntiles, tiles_m, tiles_n, tm, tn);
  };
  // Lane i keeps the origin (first row of A, first row of B) of this workgroup's i-th tile: ONE vector evaluation of the tile map
  // at kernel start.  The map costs three integer divisions; evaluated at every tile switch (twice per tile, once per prefetch
  // cursor, plus once for the output offset) it held both wave groups for ~300 cycles each time with the matrix pipes idle
  // (per-phase stamps, tools/bench_kernels.py --which stamps).  Tiles beyond the 64th fall back to the scalar evaluation.
  int tab;  // tile row index | tile column index << 16 (both < 2^15: the host checks M, N < 2^31 bytes of footprint)
  {
    int tm, tn;
    tile_mn(min(lane, max(total_tiles_hint - 1, 0)), tm, tn);
    tab = tm | (tn << 16);
  }
  // MX variants (no VGPR to spare: the table register was spilled and reloaded by a scratch load in EVERY K-tile's cursor
  // advance, an uncounted op in front of the counted waits): the table lives in LDS behind the bias windows instead and a tile
  // switch reads its entry with one uniform ds_read (+ lgkmcnt wait, once per cursor and tile).
  constexpr bool kTabLds = MX;
  const unsigned tab_lds = (unsigned)(size_t)LDS_PTR(lds + kLdsBytes + 8192 + 4096);  // 256 B; the MX / BIAS region is 12288 + 256 B
  if (kTabLds) {
    if (wave == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(tab_lds + lane * 4), "v"(tab) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all 8 waves: the first tile_rc follows at once
  }
  auto tile_rc = [&](int ti, int& ra, int& rb) {
    if (ti < 64) {
      int t;
      if (kTabLds) {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tab_lds + ti * 4) : "memory");
        t = __builtin_amdgcn_readfirstlane(v);
      } else {
        t = __builtin_amdgcn_readlane(tab, ti);
      }
      ra = (t & 0xFFFF) * TBM;
      rb = (int)((unsigned)t >> 16) * TBN;
    } else {
      int tm, tn;
      tile_mn(ti, tm, tn);
      ra = tm * TBM;
      rb = tn * TBN;
    }
  };
  auto tile_origin = [&](int ti, int& oa, int& ob, int& ra, int& rb) {
    tile_rc(ti, ra, rb);
    if (ABL == 7) ra = rb = 0;  // timing ablation: every tile streams the panels of tile (0, 0) -> all operand reads hit L2
    oa = ra * lda;
    ob = rb * ldb;
  };
  int ra_1 = 0, rb_1 = 0;
  int oa_1, ob_1, oa_2, ob_2, ti_1 = 0, kt_1 = SK ? sk_k0 : 0, ti_2 = 0, kt_2 = 0;
  int oa_0, ob_0, ra_0, rb_0, ra_2, rb_2;
  tile_origin(0, oa_0, ob_0, ra_0, rb_0);
  oa_1 = oa_2 = oa_0;
  ob_1 = ob_2 = ob_0;
  ra_1 = ra_2 = ra_0;
  rb_1 = rb_2 = rb_0;
  auto advance = [&](int& ti, int& kt, int& oa, int& ob, int& ra, int& rb, int step) {
    if (step < total) {
      if (++kt == nk) {
        kt = 0;
        ++ti;
        tile_origin(ti, oa, ob, ra, rb);
      }
    }
  };
  advance(ti_1, kt_1, oa_1, ob_1, ra_1, rb_1, 1);
  ti_2 = ti_1; kt_2 = kt_1; oa_2 = oa_1; ob_2 = ob_1; ra_2 = ra_1; rb_2 = rb_1;
  advance(ti_2, kt_2, oa_2, ob_2, ra_2, rb_2, 2);

  // ABL 17 (algo 30, timing build): odd tiles walk K downwards ("serpentine"), so that a tile starts on the K window its
  // predecessor on this CU -- and the other CUs of the XCD, which share its A panels across rounds -- ended on
  auto keff = [&](int ti, int kt) -> int { return (ABL == 17 && (ti & 1)) ? nk - 1 - kt : kt; };
  uint8_t* const buf0 = lds;
  uint8_t* const buf1 = lds + kBufBytes;
  // prologue: step 0 complete, (step 1: A0, B0) in flight
  const int kb0 = SK ? sk_k0 * BK : 0;  // first step's K offset (bytes)
  stage_scales(0, SK ? sk_k0 : 0, ra_0, rb_0);
  stage_bias(0, rb_0);
  stage_a0(oa_0 + kb0, buf0);
  stage_b0(ob_0 + kb0, buf0);
  stage_b1(ob_0 + kb0, buf0);
  stage_a1(oa_0 + kb0, buf0);
  stage_a0(oa_1 + keff(ti_1, kt_1) * BK, buf1);
  stage_b0(ob_1 + keff(ti_1, kt_1) * BK, buf1);
  if (!SK && sk_U > 0) {
    // Start stagger (sk_U = units of 512 cycles per class, 0 = off): the CUs of an XCD start in 4 classes a few microseconds
    // apart, so their epilogue store bursts (128 KiB per CU and tile, all CUs at once = 32 MiB against ~6 TB/s of store
    // bandwidth) no longer coincide: a burst that has to drain chip-wide holds the next K-tiles' loads behind it.
    const int cls = (bid >> 3) & 3;
    for (int i = 0; i < cls * sk_U; ++i) __builtin_amdgcn_s_sleep(8);
  }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();

  v8i af[4], b0f[2], b1f[2];
  int as_[4] = {kUnitScale, kUnitScale, kUnitScale, kUnitScale}, b0s[2] = {kUnitScale, kUnitScale}, b1s[2] = {kUnitScale, kUnitScale};
  int as1[4] = {kUnitScale, kUnitScale, kUnitScale, kUnitScale};  // scales of the A1 half (phases 2-3)
  const int sfr = lane & 15, sfq = lane >> 4;
  // Per-lane byte addresses of this wave's first A0 / B0 fragment in the CURRENT LDS buffer (read_frag's address math: row
  // r = lane & 15 of the 16-row group, 16-B chunks q and 4 + q through the swizzle).  The other fragments and, with the common
  // row map (MA1 == 4 / NB1 == 2), the second halves are immediate offsets away; the buffer toggle is one v_xor per register at
  // the end of every K-tile, issued inside the last MFMA segment (kBufBytes = 64 KiB does not fit the 16-bit DS offset).
  int fa_lo, fa_hi, fb_lo, fb_hi, fa1_lo = 0, fa1_hi = 0, fb1_lo = 0, fb1_hi = 0;
  {
    const int frow = (sfr >> 3) * 1024 + (sfr & 7) * 128;
    const int flo = frow + ((sfq ^ swz_f(sfr)) << 4), fhi = frow + (((4 + sfq) ^ swz_f(sfr)) << 4);
    const int l0 = (int)(size_t)LDS_PTR(lds);
    fa_lo = l0 + kOffA0 + wr * 4 * 2048 + flo;
    fa_hi = l0 + kOffA0 + wr * 4 * 2048 + fhi;
    fb_lo = l0 + kOffB0 + wc * 2 * 2048 + flo;
    fb_hi = l0 + kOffB0 + wc * 2 * 2048 + fhi;
    if (MA1 != 4) {
      fa1_lo = l0 + kOffA1 + wr * MA1 * 2048 + flo;
      fa1_hi = l0 + kOffA1 + wr * MA1 * 2048 + fhi;
    }
    if (NB1 != 2) {
      fb1_lo = l0 + kOffB1 + wc * NB1 * 2048 + flo;
      fb1_hi = l0 + kOffB1 + wc * NB1 * 2048 + fhi;
    }
  }
  // MX: per-lane LDS byte addresses of this lane's block scales in the CURRENT scale slot: A side = 4 (MA1) consecutive bytes per
  // k-block (row slot sfr of the wave's fragment rows, see PERM), B side = one byte per fragment at +16 j.  Same reason as the
  // fragment bases: left to hipcc, every (slot, operand half) x (lane part) sum becomes a hoisted VGPR, and the 256x256 MX
  // variants spilled ten of them.
  int sc_a = 0, sc_a1 = 0, sc_b = 0;
  if (MX) {
    const int s0 = (int)(size_t)LDS_PTR(sbuf) + sfq * kSK;
    sc_a = s0 + wr * (RA0 + RA1) + 4 * sfr;
    sc_a1 = s0 + wr * (RA0 + RA1) + RA0 + (MA1 == 4 ? 4 : 2) * sfr;
    sc_b = s0 + kSOp + wc * (RB0 + RB1) + sfr;
  }
  auto frag_toggle = [&]() __attribute__((always_inline)) {
    if (MX) asm volatile("v_xor_b32 %0, 0x1000, %0\n\tv_xor_b32 %1, 0x1000, %1\n\tv_xor_b32 %2, 0x1000, %2" : "+v"(sc_a), "+v"(sc_a1), "+v"(sc_b));
    asm volatile("v_xor_b32 %0, 0x10000, %0\n\tv_xor_b32 %1, 0x10000, %1\n\tv_xor_b32 %2, 0x10000, %2\n\tv_xor_b32 %3, 0x10000, %3"
                 : "+v"(fa_lo), "+v"(fa_hi), "+v"(fb_lo), "+v"(fb_hi));
    if (MA1 != 4) asm volatile("v_xor_b32 %0, 0x10000, %0\n\tv_xor_b32 %1, 0x10000, %1" : "+v"(fa1_lo), "+v"(fa1_hi));
    if (NB1 != 2) asm volatile("v_xor_b32 %0, 0x10000, %0\n\tv_xor_b32 %1, 0x10000, %1" : "+v"(fb1_lo), "+v"(fb1_hi));
  };
  static_assert(kBufBytes == 0x10000 && kSSlot == 0x1000 && (kLdsBytes % 0x2000) == 0, "frag_toggle assumes 64-KiB buffers and 4-KiB scale slots at an 8-KiB boundary");
  int s = 0;  // current step
  // One K-tile = 4 phases; LDS-DMA issued per phase: p0 {EX, B1(step+1): nB1}, p1 {A1(step+1): nA1}, p2 {A0(step+2): 2},
  // p3 {B0(step+2): 2}; fragments read at the TOP of a phase (before its wait + barrier): p0 A0/B0, p1 B1, p2 A1.
  // Wave group 1 runs one barrier behind group 0, so what a phase reads at its top must be known landed at the stagers'
  // wait TWO phases earlier (one phase earlier would be the barrier group 0 passes only AFTER those reads).  Ops allowed
  // to stay in flight at the wait of phase p (the wait follows the phase's own issues):
  //   p0: A1(step) done   -> the 2 + 2 + EX + nB1 younger ones            = W + EX - nA1
  //   p1: nothing new     -> as loose as p0's successor allows            = W + EX
  //   p2: A0/B0(step+1)   -> EX + nB1 + nA1 + 2 younger                   = W + EX - 2
  //   p3: B1(step+1)      -> nA1 + 2 + 2 younger                          = W - nB1
  // Epilogue, WOVEN INTO THE MFMA SEGMENTS around a tile boundary (non-stream-K path; `mseg`): row half 0 of a tile (final after
  // phase 1 of its last K-tile) leaves in the MFMA segments of phases 2 and 3 of that K-tile (mode 3), row half 1 in those of phases 0
  // and 1 of the next tile's first K-tile (mode 1), whose zero-C MFMAs overwrite it in phases 2 and 3.  Stores per MFMA segment:
  // 4 | 4 | MA1 | MA1.  vmcnt retires in order, so a wait must also allow the stores issued after the load it awaits: the load
  // awaited at the wait of (global) phase P was issued in the load segment of phase P - 3 (p1's: P - 4, treated as P - 3, which
  // only makes that wait stricter), so the stores of the MFMA segments of phases P - 3 .. P - 1 may stay in flight:
  //   mode 3 (flag: always):                 p3 + 4
  //   mode 1, flag = a previous tile exists:  p0 + 8 | p1 + 8 + MA1 | p2 + 4 + 2 MA1 | p3 + 2 MA1
  //   mode 2, flag = behind such a K-tile:    p0 + MA1
  //   mode 0, flag = first K-tile behind a BLOCK epilogue (stream-K path, algo 46): p0-p2 + NST, p3 + 0 (awaits B1 issued after them)
  // A store is thus first covered by a wait 4 phases after its issue.
#define MI_WAIT_SYNC(MODE_, flag, PH)                                                                              \
  {                                                                                                                \
    constexpr int kAllow = (PH) == 0 ? W + EX - nA1 : (PH) == 1 ? W + EX : (PH) == 2 ? W + EX - 2 : W - nB1;       \
    /* stores that may stay in flight: woven = MFMA segments P-3..P-1, load-segment placement = load segments P-3..P */ \
    constexpr int kXw = (MODE_) == 1 ? ((PH) == 0 ? 8 : (PH) == 1 ? 8 + MA1 : (PH) == 2 ? 4 + 2 * MA1 : 2 * MA1)   \
                      : (MODE_) == 2 ? ((PH) == 0 ? MA1 : 0)                                                       \
                      : (MODE_) == 3 ? ((PH) == 3 ? 4 : 0)                                                         \
                                     : ((PH) == 0 ? MA1 : (PH) == 3 ? 4 : 0); /* mode 4 = modes 2 and 3 in one */  \
    constexpr int kXl = (MODE_) == 1 ? ((PH) == 0 ? 8 + MA1 : (PH) == 1 ? 8 + 2 * MA1 : (PH) == 2 ? 4 + 2 * MA1 : 2 * MA1) \
                      : (MODE_) == 2 ? ((PH) == 0 ? MA1 : 0)                                                       \
                      : (MODE_) == 3 ? ((PH) == 2 ? 4 : (PH) == 3 ? 8 : 0)                                         \
                                     : ((PH) == 0 ? MA1 : (PH) == 2 ? 4 : (PH) == 3 ? 8 : 0);                      \
    constexpr int kXh = (MODE_) == 1 ? ((PH) == 0 ? 8 : (PH) == 1 ? 8 + MA1 : (PH) == 2 ? 8 + 2 * MA1 : 4 + 2 * MA1) \
                      : (MODE_) == 2 ? ((PH) == 0 ? 2 * MA1 : (PH) == 1 ? MA1 : 0)                                 \
                                     : ((PH) == 3 ? 4 : 0);                                                        \
    constexpr int kX = (MODE_) == 0 ? ((PH) != 3 ? NST : 0) : kHybrid ? kXh : kWoven ? kXw : kXl;                  \
    constexpr bool kOwn = (MODE_) >= 3 && (PH) >= 2; /* this tile's own half 0: unconditional */                   \
    if (kX != 0 && ((flag) || kOwn)) {                                                                             \
      if ((MODE_) == 4 && (PH) == 0 && !(flag)) wait_vmcnt<kAllow>();                                              \
      else wait_vmcnt<kAllow + kX>();                                                                              \
    } else wait_vmcnt<kAllow>();                                                                                   \
  }                                                                                                                \
  __builtin_amdgcn_s_barrier();                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_setprio(1);
#define MI_PIN(NI, NJ, EXPR)                                           \
  _Pragma("unroll") for (int i = 0; i < NI; ++i)                       \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(EXPR));
  // fragment i of an A half: per-fragment scale register (per-tensor path, unit scales) or byte i of the packed one (MX)
  auto mfma_sel = [&](int i, const v8i& a_, const v8i& b_, v4f c_, const int (&sa_)[4], int sb_) -> v4f {
    if (!PERM) return mfma_ba<FA, FB>(a_, b_, c_, sa_[i], sb_);
    switch (i) {
      case 0: return mfma_ba_sel<FA, FB, 0>(a_, b_, c_, sa_[0], sb_);
      case 1: return mfma_ba_sel<FA, FB, 1>(a_, b_, c_, sa_[0], sb_);
      case 2: return mfma_ba_sel<FA, FB, 2>(a_, b_, c_, sa_[0], sb_);
      default: return mfma_ba_sel<FA, FB, 3>(a_, b_, c_, sa_[0], sb_);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int ecol = (fq & 1) * 16 + (fq >> 1) * 8;  // column of this lane's 8-wide piece after the permlane16 swap
  const int d_voff = ((wr * (RA0 + RA1) + (PERM ? 4 : 1) * fr) * ldd + wc * (RB0 + RB1)) * 2;  // bytes, within the tile (half 0)
  const int d_voff1 = (PERM && MA1 != 4) ? ((wr * (RA0 + RA1) + MA1 * fr) * ldd + wc * (RB0 + RB1)) * 2 : d_voff;  // half 1
  auto tile_d_off = [&](int ti) -> int {  // uniform byte offset of tile ti's output
    int ra, rb;
    tile_rc(ti, ra, rb);
    return (ra * ldd + rb) * 2;
  };
  // whole-line form (NB1 == 2): lane (m = fr, q = fq) of the first store of a fragment pair covers row m & 7, the second row
  // 8 + (m & 7); lanes m >= 8 carry the second 32-column block of those rows
  const int d_voff_line = ((wr * (RA0 + RA1) + (PERM ? 4 : 1) * (fr & 7)) * ldd + wc * (RB0 + RB1) + (fr >> 3) * RB0 + ecol) * 2;
  const int d_voff_line1 = (PERM && MA1 != 4) ? ((wr * (RA0 + RA1) + MA1 * (fr & 7)) * ldd + wc * (RB0 + RB1) + (fr >> 3) * RB0 + ecol) * 2 : d_voff_line;
  // Epilogue of one FRAGMENT ROW (a, i) = 16 tile rows x this wave's columns, in four sub-steps that the K-tiles around a tile
  // boundary weave between their MFMAs (see `mseg`): 0 / 1 = scale (+ bias), convert and gather column block 0 / 1 into
  // 16 B per lane (8 contiguous columns from ecol: two MFMA tiles, permlane16 swap), 2 = exchange for whole lines, 3 = stores.
  // A fragment pair gives 16 rows x 64 B per store instruction: half-line segments, which one CU stores at 33 GB/s whatever
  // the cache policy, against 72-118 GB/s for whole 128-B lines (tools/probe_store.hip).  With NB1 == 2 the two column blocks
  // of a fragment row are therefore exchanged between lanes m and m ^ 8 (DPP row_ror:8, bank-masked): store 1 = rows 0-7 x 128 B,
  // store 2 = rows 8-15 x 128 B.
  typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
  v2u_ bias_w[2][2];  // BIAS: this wave's bias window (bf16 x 4 per lane and MFMA tile), read once per segment
  v4i e_p[2][2];      // sub-step state of the rows in flight: [slot][column block] (slot 1: the hybrid placement keeps two rows)
  auto epi_begin = [&](int bslot) __attribute__((always_inline)) {
    if (BIAS) {
      // This wave's own LDS-DMA data, landed long ago (covered by the vmcnt waits of the previous K-tiles), so no vmcnt wait is
      // needed -- but hipcc puts `s_waitcnt vmcnt(0)` in front of a C++ read of LDS bytes that a dword LDS-DMA may have written
      // (it cannot see the counted waits in the inline asm), which drained the whole prefetch pipeline at every use.  The reads
      // are therefore inline asm too; epi_step waits for them (lgkmcnt) before the first use.
      const unsigned bp = (unsigned)(size_t)LDS_PTR(bbuf + (bslot * 8 + wave) * 256 + fq * 8);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < (b == 0 ? 2 : NB1); ++j)
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(bias_w[b][j]) : "v"(bp), "n"((b * RB0 + j * 16) * 2));
    }
  };
  auto epi_step = [&](auto a_c, int i, int st, int d_tile, bool zero, bool first, int slot = 0) __attribute__((always_inline)) {
    v4i& e_o0 = e_p[slot][0];
    v4i& e_o1 = e_p[slot][1];
    constexpr int a = decltype(a_c)::value;
    constexpr int F = a == 0 ? 4 : MA1;
    if (ABL == 15) return;  // timing build (algo 28): no epilogue at all -- neither conversion nor stores (wrong results)
    auto biased = [&](v4f v, int b, int j) __attribute__((always_inline)) -> v4f {
      if (BIAS) {
        v[0] += __uint_as_float(bias_w[b][j].x << 16);
        v[1] += __uint_as_float(bias_w[b][j].x & 0xFFFF0000u);
        v[2] += __uint_as_float(bias_w[b][j].y << 16);
        v[3] += __uint_as_float(bias_w[b][j].y & 0xFFFF0000u);
      }
      return v;
    };
    auto pack_blk = [&](int b) __attribute__((always_inline)) -> v4i {
      v4i r;
      if (b == 1 && NB1 == 1) {  // single-tile block of the 192-column shapes: 4 columns in .x/.y
        const v4f v0 = biased(acc[a][i][1][0] * alpha, 1, 0);
        r = (v4i){(int)pack_bf16x2(v0[0], v0[1]), (int)pack_bf16x2(v0[2], v0[3]), 0, 0};
      } else {
        const v4f v0 = biased(acc[a][i][b][0] * alpha, b, 0), v1 = biased(acc[a][i][b][1] * alpha, b, 1);
        u32 p0x = pack_bf16x2(v0[0], v0[1]), p0y = pack_bf16x2(v0[2], v0[3]);
        u32 p1x = pack_bf16x2(v1[0], v1[1]), p1y = pack_bf16x2(v1[2], v1[3]);
        auto sx = __builtin_amdgcn_permlane16_swap(p0x, p1x, false, false);
        auto sy = __builtin_amdgcn_permlane16_swap(p0y, p1y, false, false);
        r = (v4i){(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
      }
      if (zero) {
#pragma unroll
        for (int j = 0; j < (b == 0 ? 2 : NB1); ++j) acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
      return r;
    };
    constexpr bool LINES = NB1 == 2 && ABL != 11;
    if (st == 0) {
      if (BIAS && first) {  // the window reads of epi_begin: tie the registers to the wait so that no use is scheduled above it
        if (NB1 == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias_w[0][0]), "+v"(bias_w[0][1]), "+v"(bias_w[1][0]), "+v"(bias_w[1][1])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias_w[0][0]), "+v"(bias_w[0][1]), "+v"(bias_w[1][0])::"memory");
      }
      e_o0 = pack_blk(0);
    } else if (st == 1) {
      e_o1 = pack_blk(1);
    } else if (st == 2) {
      if (LINES) {
        // x = block 0 of rows m < 8 | block 1 of rows m - 8;  y = block 0 of rows m + 8 | block 1 of rows m >= 8: lane m takes
        // from lane m ^ 8 (row_ror:8) in the banks (groups of 4 lanes of a 16-lane row) where it holds the other block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int keep = e_o1[e];
          e_o1[e] = __builtin_amdgcn_update_dpp(e_o1[e], e_o0[e], 0x128, 0xF, 0x3, false);  // banks 0-1 (m < 8) <- block 0 of row m + 8
          e_o0[e] = __builtin_amdgcn_update_dpp(e_o0[e], keep, 0x128, 0xF, 0xC, false);     // banks 2-3 (m >= 8) <- block 1 of row m - 8
        }
      }
    } else {
      // hipcc (ROCm 7.2) lets the next VALU overwrite the data registers of a 16-byte store (SGPR-offset form) with no wait
      // state: lanes 12-15 of every 16-lane row then stored the NEXT block's unconverted fp32 (seen on MI355X).  Every store is
      // followed by `s_nop 1` that keeps its registers live across the required wait states.
      // aux 16 = sc1: write-through, the line is not kept in this XCD's L2.  A tile's 128 KiB of output per CU (4 MiB per XCD =
      // its whole L2) would otherwise evict the A/B panels the next tile streams (measured -5..6 % kernel time at K = 2048-4096).
      const int rowoff = d_tile + ((a * RA0 + (PERM ? i : i * 16)) * ldd) * 2;
      if (ABL == 1) {
        asm volatile("" ::"v"(e_o0), "v"(e_o1));
      } else if (LINES) {
        const int dvo = a == 0 ? d_voff_line : d_voff_line1;
        constexpr int kRow8 = 8 * (PERM ? F : 1);
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)e_o0, rsD, dvo, rowoff, kStoreAux);
        asm volatile("s_nop 1" ::"v"(e_o0) : "memory");
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)e_o1, rsD, dvo, rowoff + kRow8 * ldd * 2, kStoreAux);
        asm volatile("s_nop 1" ::"v"(e_o1) : "memory");
      } else {  // half-line stores: 192-column shapes (NB1 == 1: 64 B + 32 B per row), and ABL == 11 (A/B baseline)
        const int dvo = a == 0 ? d_voff : d_voff1;
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)e_o0, rsD, dvo + ecol * 2, rowoff, kStoreAux);
        asm volatile("s_nop 1" ::"v"(e_o0) : "memory");
        if (NB1 == 2) {
          __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)e_o1, rsD, dvo + ecol * 2, rowoff + RB0 * 2, kStoreAux);
        } else {
          const v2u_ o2 = {(unsigned)e_o1[0], (unsigned)e_o1[1]};
          __builtin_amdgcn_raw_buffer_store_b64(o2, rsD, dvo + fq * 8, rowoff + RB0 * 2, 16);
        }
        asm volatile("s_nop 1" ::"v"(e_o1) : "memory");
      }
    }
  };
  // kHybrid (ABL 16, algo 29, timing build): conversion woven into the MFMA segments, stores one load segment later
  constexpr bool kHybrid = ABL == 16;
  using c0_t = std::integral_constant<int, 0>;
  using c1_t = std::integral_constant<int, 1>;
  using c2_t = std::integral_constant<int, 2>;
  using c3_t = std::integral_constant<int, 3>;
  using c4_t = std::integral_constant<int, 4>;
  // the fragment rows [r0, r0 + nr) of half a, back to back (after the tile walk; stream-K / block-epilogue path)
  auto epi_rows = [&](auto a_c, int r0, int nr, int d_tile, int bslot, bool zero) __attribute__((always_inline)) {
    epi_begin(bslot);
#pragma unroll
    for (int r = 0; r < nr; ++r)
#pragma unroll
      for (int st = 0; st < 4; ++st) epi_step(a_c, r0 + r, st, d_tile, zero, r == 0);
  };
  auto epilogue_block = [&](int ti, bool zero) __attribute__((always_inline)) {  // both halves at once (stream-K / block-epilogue path)
    const int d_tile = tile_d_off(ti);
    epi_rows(c0_t{}, 0, 4, d_tile, ti & 1, zero);
    epi_rows(c1_t{}, 0, MA1, d_tile, ti & 1, zero);
  };
  // An MFMA segment of NI x NJ MFMAs (mf(i, j)) with, when `on`, the epilogue of NR fragment rows [r0, r0 + NR) of half a woven
  // in: after MFMA k the sub-steps [4 NR k / NM, 4 NR (k + 1) / NM).  The conversion VALU work then runs in the shadow of the
  // wave's OWN MFMAs (the matrix pipe is busy 32 cycles per MFMA, the wave's issue port 4-8), and the stores leave at the rate
  // the CU's store path takes them.  In the load segments -- where the quadrants were converted before -- every cycle of
  // epilogue work held the partner wave group at its barrier with the matrix pipe idle, and the two groups' shares came one
  // after the other: per-phase stamps showed ~8.7k cycles per tile boundary, 16 % of a K = 3072 tile.
  auto mseg = [&](auto ni_c, auto nj_c, auto mf, auto a_c, auto nr_c, int r0, bool on, int d_tile, int bslot, bool zero)
                  __attribute__((always_inline)) {
    constexpr int NI = decltype(ni_c)::value, NJ = decltype(nj_c)::value, NM = NI * NJ;
    constexpr int SPR = kHybrid ? 3 : 4;  // sub-steps per row woven here (hybrid: the stores follow in the NEXT load segment)
    constexpr int NR = decltype(nr_c)::value, NE = SPR * NR, MAXU = NR == 0 ? 0 : (NE + NM - 1) / NM;
    if (NR > 0 && on) epi_begin(bslot);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        mf(i, j);
        if (NR > 0) {
          const int k = i * NJ + j;
          if (on) {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
              const int e = k * NE / NM + u;
              if (e < (k + 1) * NE / NM) epi_step(a_c, r0 + e / SPR, e % SPR, d_tile, zero, e < SPR, kHybrid ? e / SPR : 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // The same rows in a LOAD segment instead (after the segment's own LDS-DMA issues, before its wait): the default placement.
  // Measured (same-process A/B, 3B decoder shapes) the woven form is 1.8 % SLOWER: a wave whose store cannot issue (the CU's store
  // path takes 22 cycles per 1-KiB store, and 4 waves store at once) also cannot issue its next MFMA, so the matrix pipe idles
  // either way, and in a load segment the partner group's MFMAs at least run undisturbed.  kWoven (ABL 14, algo 27) keeps the
  // woven form as a timing build.
  constexpr bool kWoven = ABL == 14 || ABL == 16;
  // hybrid: the stores of the NR rows [r0, r0 + NR) of half a that the previous MFMA segment converted (slots 0 .. NR - 1)
  auto lseg_store = [&](auto a_c, auto nr_c, int r0, bool on, int d_tile) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_c)::value;
    if (NR > 0 && kHybrid) {
      if (on) {
#pragma unroll
        for (int r = 0; r < NR; ++r) epi_step(a_c, r0 + r, 3, d_tile, false, false, r);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto lseg_epi = [&](auto a_c, auto nr_c, int r0, bool on, int d_tile, int bslot, bool zero) __attribute__((always_inline)) {
    constexpr int NR = decltype(nr_c)::value;
    if (NR > 0 && !kWoven) {
      if (on) epi_rows(a_c, r0, NR, d_tile, bslot, zero);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // mode 0: plain K-tile (flag: behind a block epilogue); mode 1: first K-tile of a tile -- zero-C MFMAs, and when `flag` row half 1
  // of the PREVIOUS tile (output offset d_tile, bias slot bslot) leaves woven into the MFMA segments of phases 0 and 1; mode 2: the
  // K-tile behind a mode-1 K-tile (flag: that one carried stores); mode 3: last K-tile of a tile -- row half 0 of THIS tile, final
  // after phase 1, leaves in the MFMA segments of phases 2 and 3; mode 4 (K = 256: two K-tiles per tile) = modes 2 and 3 in one.
  auto ktile = [&](auto mode_c, uint8_t* cur, uint8_t* oth, bool flag, int slot, int d_tile, int bslot) {
    constexpr int MODE = decltype(mode_c)::value;
    // (zero-C MFMAs through the builtin were tried: the allocator then stops accumulating in place, spills the fresh quadrants and
    // reloads them behind `s_waitcnt vmcnt(0)`; mfma_ba_zero ties the accumulator in inline asm.  MX has no unit-scale form: its
    // quadrants are zeroed by v_mov as they are converted.)
    constexpr bool ZC = MODE == 1 && !MX;  // first K-tile of a tile: accumulate onto zero (mfma_ba_zero), no v_mov zeroing
    using nr01_t = std::integral_constant<int, (MODE == 1 ? MA1 / 2 : 0)>;  // fragment rows leaving in phases 0, 1 (half 1, previous tile)
    using nr23_t = std::integral_constant<int, (MODE >= 3 ? 2 : 0)>;        // ... in phases 2, 3 (half 0, this tile)
    using wv01_t = std::integral_constant<int, (kWoven ? nr01_t::value : 0)>;  // ... of them woven into the MFMA segment
    using wv23_t = std::integral_constant<int, (kWoven ? nr23_t::value : 0)>;
    using ma1_t = std::integral_constant<int, MA1>;
    using nb1_t = std::integral_constant<int, NB1>;
    const int sa1 = oa_1 + keff(ti_1, kt_1) * BK, sb1 = ob_1 + keff(ti_1, kt_1) * BK;
    const int sa2 = oa_2 + keff(ti_2, kt_2) * BK, sb2 = ob_2 + keff(ti_2, kt_2) * BK;
    // Fragment reads: explicit per-lane bases (fa_* / fb_*, see their definition) + immediate offsets -- no address VALU in the
    // load segments (8 v_add per K-tile there cost 2 %: every VALU instruction delays the segment's LDS-DMA issue and barrier)
    // and no compiler-hoisted base per (operand half, buffer) (ten VGPRs, some spilled and reloaded behind `s_waitcnt vmcnt(0)`).
    auto frag2 = [&](int lo, int hi, int off) __attribute__((always_inline)) -> v8i {
      const v4i l = *reinterpret_cast<const __attribute__((address_space(3))) v4i*>((size_t)(unsigned)(lo + off));
      const v4i h = *reinterpret_cast<const __attribute__((address_space(3))) v4i*>((size_t)(unsigned)(hi + off));
      return (v8i){l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
    };
    auto frag = [&](int half_off, int, int g) __attribute__((always_inline)) -> v8i {
      if (half_off == kOffA0) return frag2(fa_lo, fa_hi, g * 2048);
      if (half_off == kOffB0) return frag2(fb_lo, fb_hi, g * 2048);
      if (half_off == kOffA1) return MA1 == 4 ? frag2(fa_lo, fa_hi, kHalfBytes + g * 2048) : frag2(fa1_lo, fa1_hi, g * 2048);
      return NB1 == 2 ? frag2(fb_lo, fb_hi, kHalfBytes + g * 2048) : frag2(fb1_lo, fb1_hi, g * 2048);
    };
    // ---- phase 0: C[0][*][0][*]
    // A load segment that carries epilogue rows issues its LDS-DMA first, then the rows, and reads its fragments LAST: the
    // fragments are then not live across the conversion (the 256x256 variants sit at the 256-VGPR limit), at the price of their
    // LDS latency once per such segment.
    constexpr bool kE01 = !kWoven && nr01_t::value > 0, kE23 = !kWoven && nr23_t::value > 0;
    // hybrid store segments: mode 3 p3 (half 0 rows 0-1 of this tile); mode 1 p0 (half 0 rows 2-3 of the previous tile), p1 and
    // p2 (half 1 of the previous tile)
    constexpr bool kH1 = kHybrid && MODE == 1, kH3 = kHybrid && MODE >= 3;
    using h2_t = std::integral_constant<int, 2>;
    using hm_t = std::integral_constant<int, MA1 / 2>;
    if (kE01 || kH1) {
      stage_scales(slot ^ 1, kt_1, ra_1, rb_1);
      stage_bias(ti_1 & 1, rb_1);
      stage_b1(sb1, oth);
      if (kH1) lseg_store(c0_t{}, h2_t{}, 2, flag, d_tile);
      else lseg_epi(c1_t{}, nr01_t{}, 0, flag, d_tile, bslot, MX);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) b0f[j] = frag(kOffB0, wc * 2, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag(kOffA0, wr * 4, i);
    if (MX && ABL != 5) {  // every scale of this K-tile up front: phases 1-3 then start on their fragment reads alone
      // (inline asm: the values are consumed behind MI_WAIT_SYNC's `s_waitcnt lgkmcnt(0)` + sched_barrier, like the fragments)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(b0s[j]) : "v"(sc_b), "n"(j * 16));
#pragma unroll
      for (int j = 0; j < NB1; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(b1s[j]) : "v"(sc_b), "n"(RB0 + j * 16));
      // A side: the lane's 4 (A0) and MA1 (A1) fragment scales as one dword / one u16 (sfr = row slot, see PERM)
      asm volatile("ds_read_b32 %0, %1" : "=v"(as_[0]) : "v"(sc_a));
      if (MA1 == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(as1[0]) : "v"(sc_a1));
      else asm volatile("ds_read_u16 %0, %1" : "=v"(as1[0]) : "v"(sc_a1));
      if (ABL == 6) {  // timing ablation: the scales are read but the MFMAs get unit scales
        asm volatile("" ::"v"(as_[0]), "v"(as1[0]));
        as_[0] = kUnitScale;
        as1[0] = kUnitScale;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          asm volatile("" ::"v"(b0s[j]), "v"(b1s[j]));
          b0s[j] = kUnitScale;
          b1s[j] = kUnitScale;
        }
      }
    }
    if (!(kE01 || kH1)) {
      stage_scales(slot ^ 1, kt_1, ra_1, rb_1);
      stage_bias(ti_1 & 1, rb_1);
      stage_b1(sb1, oth);
    }
    MI_WAIT_SYNC(MODE, flag, 0)
    stamp();
    mseg(c4_t{}, c2_t{}, [&](int i, int j) __attribute__((always_inline)) {
      if (ZC) mfma_ba_zero<FA, FB>(af[i], b0f[j], acc[0][i][0][j], b0s[j]);
      else acc[0][i][0][j] = mfma_sel(i, af[i], b0f[j], acc[0][i][0][j], as_, b0s[j]);
    }, c1_t{}, wv01_t{}, 0, flag, d_tile, bslot, MX);
    MI_PIN(4, 2, acc[0][i][0][j])
    MI_PHASE_END();
    stamp();
    // ---- phase 1: C[0][*][1][*]
    if (kE01 || kH1) {
      stage_a1(sa1, oth);
      if (kH1) lseg_store(c1_t{}, hm_t{}, 0, flag, d_tile);
      else lseg_epi(c1_t{}, nr01_t{}, MA1 / 2, flag, d_tile, bslot, MX);
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) b1f[j] = frag(kOffB1, wc * NB1, j);
    if (!(kE01 || kH1)) stage_a1(sa1, oth);
    MI_WAIT_SYNC(MODE, flag, 1)
    stamp();
    mseg(c4_t{}, nb1_t{}, [&](int i, int j) __attribute__((always_inline)) {
      if (ZC) mfma_ba_zero<FA, FB>(af[i], b1f[j], acc[0][i][1][j], b1s[j]);
      else acc[0][i][1][j] = mfma_sel(i, af[i], b1f[j], acc[0][i][1][j], as_, b1s[j]);
    }, c1_t{}, wv01_t{}, MA1 / 2, flag, d_tile, bslot, MX);
    MI_PIN(4, NB1, acc[0][i][1][j])
    MI_PHASE_END();
    stamp();
    // ---- phase 2: C[1][*][1][*]
    if (kE23 || kH1) {
      stage_a0(sa2, cur);
      if (kH1) lseg_store(c1_t{}, hm_t{}, MA1 / 2, flag, d_tile);
      else lseg_epi(c0_t{}, nr23_t{}, 0, true, d_tile, bslot, MX);
    }
#pragma unroll
    for (int i = 0; i < MA1; ++i) af[i] = frag(kOffA1, wr * MA1, i);
    if (!(kE23 || kH1)) stage_a0(sa2, cur);
    MI_WAIT_SYNC(MODE, flag, 2)
    stamp();
    mseg(ma1_t{}, nb1_t{}, [&](int i, int j) __attribute__((always_inline)) {
      if (ZC) mfma_ba_zero<FA, FB>(af[i], b1f[j], acc[1][i][1][j], b1s[j]);
      else acc[1][i][1][j] = mfma_sel(i, af[i], b1f[j], acc[1][i][1][j], as1, b1s[j]);
    }, c0_t{}, wv23_t{}, 0, true, d_tile, bslot, MX);
    MI_PIN(MA1, NB1, acc[1][i][1][j])
    MI_PHASE_END();
    stamp();
    // ---- phase 3: C[1][*][0][*]
    stage_b0(sb2, cur);
    if (kH3) lseg_store(c0_t{}, h2_t{}, 0, true, d_tile);
    lseg_epi(c0_t{}, nr23_t{}, 2, true, d_tile, bslot, MX);
    MI_WAIT_SYNC(MODE, flag, 3)
    stamp();
    mseg(ma1_t{}, c2_t{}, [&](int i, int j) __attribute__((always_inline)) {
      if (ZC) mfma_ba_zero<FA, FB>(af[i], b0f[j], acc[1][i][0][j], b0s[j]);
      else acc[1][i][0][j] = mfma_sel(i, af[i], b0f[j], acc[1][i][0][j], as1, b0s[j]);
    }, c0_t{}, wv23_t{}, 2, true, d_tile, bslot, MX);
    frag_toggle();
    MI_PIN(MA1, 2, acc[1][i][0][j])
    MI_PHASE_END();
    stamp();
    // cursors follow the step
    ++s;
    advance(ti_1, kt_1, oa_1, ob_1, ra_1, rb_1, s + 1);
    advance(ti_2, kt_2, oa_2, ob_2, ra_2, rb_2, s + 2);
  };

  // SK: partial accumulators of slot `slot` (this wave's 32 KiB: [acc index][lane] x 16 B, whole 1-KiB lines per store)
  const rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(SK ? (void*)sk_ws : (void*)A), 0, SK ? G * (8 * 32 * 1024) : 0, 0x00020000);
  auto partial_store = [&](int slot) {
    const int base = (slot * 8 + wave) * (32 * 1024);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < (a == 0 ? 4 : MA1); ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < (b == 0 ? 2 : NB1); ++j) {
            const v4f v = acc[a][i][b][j];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mi::v4u, v), rsW, lane * 16, base + (((a * 4 + i) * 2 + b) * 2 + j) * 1024, 16);
            asm volatile("s_nop 1" ::"v"(v) : "memory");
            acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};
          }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores acknowledged (and the prefetches landed)
    __builtin_amdgcn_s_barrier();  // wave group 1 passes this one only after group 0 passed its own (group 0 is a phase ahead)
    if (wave == 4 && lane == 0) __hip_atomic_store(sk_flags + slot, sk_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto partial_add = [&](int slot) {
    if (lane == 0) {
      while (__hip_atomic_load(sk_flags + slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != sk_epoch) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_wave_barrier();
    const int base = (slot * 8 + wave) * (32 * 1024);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < (a == 0 ? 4 : MA1); ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < (b == 0 ? 2 : NB1); ++j) {
            const mi::v4u w = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane * 16, base + (((a * 4 + i) * 2 + b) * 2 + j) * 1024, 16);
            acc[a][i][b][j] += __builtin_bit_cast(v4f, w);
          }
  };

  unsigned long long dg_c0 = 0, dg_r0 = 0;
  if (ABL == 8) {  // diagnostic build: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the whole tile walk
    dg_c0 = __builtin_amdgcn_s_memtime();
    dg_r0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (SK || !DEPI) {  // block epilogue after each tile (stream-K; DEPI = false is kept as the A/B baseline, algo 46)
    int kt = SK ? sk_k0 : 0, ti = 0;
    bool after = false;  // previous step ended with an epilogue whose stores are still in flight
    for (int pair = 0; pair < total / 2; ++pair) {
      ktile(c0_t{}, buf0, buf1, after, 0, 0, 0);
      ktile(c0_t{}, buf1, buf0, false, 1, 0, 0);
      after = false;
      kt += 2;
      if (kt == nk) {
        if (SK && ti == 0 && sk_k0 != 0) {
          partial_store(sk_v);
        } else {
          epilogue_block(ti, true);
          after = true;
        }
        kt = 0;
        ++ti;
      }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SK && kt != 0) {  // the range ended inside a tile: fetch the rest from the neighbour and finish it
      partial_add(sk_v + 1);
      epilogue_block(ti, false);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    int d_prev = 0;
    using c4m_t = std::integral_constant<int, 4>;
    if (nk == 2) {
      for (int ti = 0; ti < my_tiles; ++ti) {
        const bool have_prev = ti > 0;
        ktile(c1_t{}, buf0, buf1, have_prev, 0, d_prev, (ti - 1) & 1);
        d_prev = tile_d_off(ti);
        ktile(c4m_t{}, buf1, buf0, have_prev, 1, d_prev, ti & 1);
      }
    } else {
      for (int ti = 0; ti < my_tiles; ++ti) {
        const bool have_prev = ti > 0;
        ktile(c1_t{}, buf0, buf1, have_prev, 0, d_prev, (ti - 1) & 1);
        ktile(c2_t{}, buf1, buf0, have_prev, 1, 0, 0);
        for (int pair = 2; pair < nk / 2; ++pair) {
          ktile(c0_t{}, buf0, buf1, false, 0, 0, 0);
          ktile(c0_t{}, buf1, buf0, false, 1, 0, 0);
        }
        d_prev = tile_d_off(ti);
        ktile(c0_t{}, buf0, buf1, false, 0, 0, 0);
        ktile(c3_t{}, buf1, buf0, true, 1, d_prev, ti & 1);
      }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_tiles > 0) {
      if (kHybrid) {  // half 0 rows 2-3 of the last tile: converted in its last MFMA segment, not stored yet
        epi_step(c0_t{}, 2, 3, d_prev, false, false, 0);
        epi_step(c0_t{}, 3, 3, d_prev, false, false, 1);
      }
      epi_rows(c1_t{}, 0, MA1, d_prev, (my_tiles - 1) & 1, false);  // row half 1 of the last tile
    }
    if (ABL == 9) {
      if (blockIdx.x == 0 && (wave & 3) == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        unsigned long long* dbg = (unsigned long long*)bias + (wave >> 2) * 1024;
        for (int i = lane; i < 1024; i += 64) dbg[i] = i < stamp_idx ? stamp_lds[i] : 0ull;
      }
    }
    if (ABL == 8) {  // `bias` is a u64[4 * gridDim.x] buffer of its own: {cycles, 100 MHz ticks, steps, xcc id}
      const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (tid == 0) {
        unsigned long long* dbg = (unsigned long long*)bias + (size_t)bid * 4;
        dbg[0] = c1 - dg_c0;
        dbg[1] = r1 - dg_r0;
        dbg[2] = (unsigned long long)total;
        dbg[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
      }
    }
  }
#undef MI_WAIT_SYNC
#undef MI_PIN
}


// Tile shape for the persistent kernel: minimise rounds x tile area / efficiency over the shapes that divide M, N.
// cfg: 0 = 256x256, 1 = 256x192, 2 = 192x256, 3 = 192x192.
static int pick_tile_cfg(int64_t M, int64_t N, int64_t K) {
  static const int bm[4] = {256, 256, 192, 192}, bn[4] = {256, 192, 256, 192};
  // measured relative efficiency of the shorter phases.  192-COLUMN tiles (NB1 == 1) store half lines (a wave's 48 columns = 64
  // + 32 B per row), 192-ROW tiles keep whole 128-byte lines: 6144x6144x4096 runs 126.5 us as 256x192 and 118.8 us as 192x256
  // tiles.  And half-line stores pay extra when the output lines are COLD, which in a training step they always are (the
  // output is a fresh 50-MB tensor; a benchmark loop rewrites the same one): 8192x3072x3072 as 256x192 tiles takes 74.9 us into
  // a warm output and 90.5 us rotating over 12 outputs, 256x256 tiles 78.0 / 77.4 us (tools/bench_kernels.py --which rotout).
  // The penalty is per output byte, i.e. ~1/K of the tile time: eff = 0.88 / (1 + 1150 / K) fits K = 3072 .. 16384.
  const double cold = 1.0 + 1150.0 / (double)(K > 0 ? K : 1);
  const double eff[4] = {1.0, 0.88 / cold, 0.92, 0.80 / cold};
  const int ncu = num_cus();
  int best = -1;
  double best_cost = 0;
  // Measured on MI355X with interleaved A/B timing (tools/bench_kernels.py --which tiles; back-to-back timing is biased by
  // clock drift): 8192x3072xK as 512 tiles of 256x192 beats 384 tiles of 256x256 by 5 % (K 3072) to 9 % (K 16384).
  for (int c = 0; c < 4; ++c) {
    if (M % bm[c] || N % bn[c]) continue;
    const int64_t tiles = (M / bm[c]) * (N / bn[c]);
    const int64_t rounds = (tiles + ncu - 1) / ncu;
    const double cost = (double)rounds * bm[c] * bn[c] / eff[c];
    if (best < 0 || cost < best_cost * 0.98) {  // prefer the larger tile unless the gain is > 2 % (in the step fc2 fprop, K = 8192: 170 us as 256x192, 178 us as 256x256 tiles)
      best = c;
      best_cost = cost;
    }
  }
  return best;
}

// stream-K workspace registered by the caller (mi_gemm_set_workspace): [4 KiB of flags][slots of 256 KiB]
struct SkWorkspace {
  unsigned int* flags = nullptr;
  float* slots = nullptr;
  int nslots = 0;
};
static SkWorkspace g_sk_ws[32];
static std::atomic<unsigned int> g_sk_epoch{1};

static const SkWorkspace* sk_workspace() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
  return g_sk_ws[dev].flags ? &g_sk_ws[dev] : nullptr;
}

// stream-K plan for 256x256 tiles: units per workgroup (0 = not applicable / not worth it)
static int sk_units(int64_t M, int64_t N, int64_t K, bool force) {
  if (M % 256 || N % 256 || K % 256) return 0;
  const SkWorkspace* ws = sk_workspace();
  const int ncu = num_cus();
  if (!ws || ws->nslots < ncu) return 0;
  const int64_t ntiles = (M / 256) * (N / 256), nk = K / BK;
  if (ntiles <= ncu || ntiles % ncu == 0) return 0;
  const int64_t rounds = (ntiles + ncu - 1) / ncu;
  const double waste = 1.0 - (double)ntiles / (double)(rounds * ncu);
  if (!force && waste < 0.06) return 0;
  int64_t U = (ntiles * nk + ncu - 1) / ncu;
  U += U & 1;
  return U >= nk ? (int)U : 0;
}

template <int FA, int FB, bool MX, bool BIAS, int ABL, int MA1, int NB1, bool DEPI = true>
static void launch_p8_cfg(const uint8_t* a, const uint8_t* b, uint16_t* D, const float* sa_inv, const float* sb_inv,
                          const uint8_t* SA, const uint8_t* SB, const uint16_t* bias, int64_t M, int64_t N, int64_t K,
                          int64_t lda, int64_t ldb, int64_t ldd, hipStream_t st, bool one_tile_per_wg = false) {
  constexpr int TBM = 2 * (64 + 16 * MA1), TBN = 4 * (32 + 16 * NB1);
  const int tiles_m = (int)(M / TBM), tiles_n = (int)(N / TBN);
  // one_tile_per_wg (algo 5): same kernel, one workgroup per tile, so the hardware dispatcher balances the tiles over whatever
  // CUs are free -- the form to use while other kernels (RCCL collectives) hold part of the chip
  int grid = (one_tile_per_wg || tiles_m * tiles_n < num_cus()) ? tiles_m * tiles_n : num_cus();
#ifdef MI_DIAG
  if ((ABL == 3 || ABL == 8 || ABL == 9) && getenv("MI_GEMM_GRID")) grid = std::max(1, std::min(grid, atoi(getenv("MI_GEMM_GRID"))));  // experiment knob
  const int stagger = ((ABL == 3) && getenv("MI_GEMM_STAGGER")) ? atoi(getenv("MI_GEMM_STAGGER")) : 0;  // experiment knob (algo 16)
#else
  const int stagger = 0;
#endif
  hipLaunchKernelGGL((gemm_256_p8<FA, FB, ABL, MX, BIAS, MA1, NB1, false, DEPI>), dim3(grid), dim3(512), 0, st, a, b, D, sa_inv, sb_inv, (int)K,
                     (int)lda, (int)ldb, (int)ldd, tiles_m, tiles_n, (int)(M * lda), (int)(N * ldb), (int)(M * ldd * 2), SA, SB,
                     (int)M, (int)N, bias, (float*)nullptr, (unsigned int*)nullptr, 0u, stagger);
}

template <int FA, int FB, bool MX, bool BIAS>
static void launch_p8_sk(const uint8_t* a, const uint8_t* b, uint16_t* D, const float* sa_inv, const float* sb_inv,
                         const uint8_t* SA, const uint8_t* SB, const uint16_t* bias, int64_t M, int64_t N, int64_t K,
                         int64_t lda, int64_t ldb, int64_t ldd, int U, hipStream_t st) {
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
  const int64_t total = (int64_t)tiles_m * tiles_n * (K / BK);
  const int grid = (int)((total + U - 1) / U);
  const SkWorkspace* ws = sk_workspace();
  const unsigned int epoch = g_sk_epoch.fetch_add(1);
  hipLaunchKernelGGL((gemm_256_p8<FA, FB, 0, MX, BIAS, 4, 2, true>), dim3(grid), dim3(512), 0, st, a, b, D, sa_inv, sb_inv, (int)K,
                     (int)lda, (int)ldb, (int)ldd, tiles_m, tiles_n, (int)(M * lda), (int)(N * ldb), (int)(M * ldd * 2), SA, SB,
                     (int)M, (int)N, bias, ws->slots, ws->flags, epoch == 0 ? g_sk_epoch.fetch_add(1) : epoch, U);
}

template <int FA, int FB>
static int launch_p8(const uint8_t* a, const uint8_t* b, uint16_t* D, const float* sa_inv, const float* sb_inv,
                     const uint8_t* SA, const uint8_t* SB, const uint16_t* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldb, int64_t ldd, int algo, bool mx, hipStream_t st, void* clock_stamps = nullptr) {
  // algo 44: stream-K on 256x256 tiles (needs a registered workspace).  It is NOT part of the automatic choice: measured
  // (interleaved A/B) it loses 3-15 % to the best whole-tile shape at K <= 8192 and wins 3.5 % only at K = 16384 -- the chip
  // is power-limited, so a half-empty last round costs less than its CU count suggests (the busy CUs clock higher) while the
  // partial-accumulator traffic is extra energy.  45 = the whole-tile picker (same as 4).
  if (algo == 44) {
    const int U = sk_units(M, N, K, algo == 44);
    if (U > 0) {
      if (mx) {
        if (bias) launch_p8_sk<FA, FB, true, true>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, U, st);
        else launch_p8_sk<FA, FB, true, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, U, st);
      } else {
        if (bias) launch_p8_sk<FA, FB, false, true>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, U, st);
        else launch_p8_sk<FA, FB, false, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, U, st);
      }
      MI_CHECK_LAUNCH("mi_gemm (stream-K) launch");
      return MI_OK;
    }
    if (algo == 44) {
      set_error("mi_gemm: stream-K needs a registered workspace, 256-aligned M/N/K and more tiles than CUs (%lld x %lld x %lld)",
                (long long)M, (long long)N, (long long)K);
      return MI_ERR_SHAPE;
    }
  }
  int cfg = (algo >= 40 && algo <= 43) ? algo - 40 : pick_tile_cfg(M, N, K);
  static const int bm[4] = {256, 256, 192, 192}, bn[4] = {256, 192, 256, 192};
  if (cfg < 0 || M % bm[cfg] || N % bn[cfg]) {
    set_error("mi_gemm: no persistent tile shape divides %lld x %lld", (long long)M, (long long)N);
    return MI_ERR_SHAPE;
  }
#define MI_P8(MXv, BIASv, ABLv, MA1v, NB1v) \
  launch_p8_cfg<FA, FB, MXv, BIASv, ABLv, MA1v, NB1v>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, st, algo == 5)
#define MI_P8_CFG(MXv, BIASv, ABLv)                                 \
  switch (cfg) {                                                    \
    case 0: MI_P8(MXv, BIASv, ABLv, 4, 2); break;                   \
    case 1: MI_P8(MXv, BIASv, ABLv, 4, 1); break;                   \
    case 2: MI_P8(MXv, BIASv, ABLv, 2, 2); break;                   \
    default: MI_P8(MXv, BIASv, ABLv, 2, 1); break;                  \
  }
#ifdef MI_DIAG
  if (algo == 46 || (algo >= 15 && algo <= 30)) {  // timing-only / diagnostic builds: E4M3 x E4M3 only (compile time)
    if constexpr (FA == 0 && FB == 0) {
      if (algo == 46) {  // A/B baseline: block epilogue after each tile (the round-1 form)
        switch (cfg) {
          case 0: launch_p8_cfg<FA, FB, false, false, 0, 4, 2, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, st); break;
          case 1: launch_p8_cfg<FA, FB, false, false, 0, 4, 1, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, st); break;
          case 2: launch_p8_cfg<FA, FB, false, false, 0, 2, 2, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, st); break;
          default: launch_p8_cfg<FA, FB, false, false, 0, 2, 1, false>(a, b, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, st); break;
        }
      } else if (algo == 15) {  // no stores
        MI_P8_CFG(false, false, 1)
      } else if (algo == 16) {  // start stagger (MI_GEMM_STAGGER)
        MI_P8_CFG(false, false, 3)
      } else if (algo == 17) {  // plain (write-back) stores
        MI_P8_CFG(false, false, 4)
      } else if (algo == 20) {  // every tile reads the operand panels of tile (0, 0) (L2-resident): wrong results
        MI_P8_CFG(false, false, 7)
      } else if (algo == 22) {  // `bias` is a u64[2048] buffer: per-phase s_memtime stamps of waves 0 and 4 of workgroup 0
        MI_P8_CFG(false, false, 9)
      } else if (algo == 21) {  // `bias` is a u64[4 * grid] stamp buffer (cycles, 100 MHz ticks, steps, XCC id)
        MI_P8_CFG(false, false, 8)
      } else if (algo == 30) {  // odd tiles walk K downwards (L2 reuse across tile boundaries; fp32 summation order differs per tile parity)
        MI_P8_CFG(false, false, 17)
      } else if (algo == 29) {  // conversion woven into the MFMA segments, stores one load segment later (timing A/B; K >= 512)
        if (K < 512) { set_error("mi_gemm: algo 29 needs K >= 512"); return MI_ERR_SHAPE; }
        MI_P8_CFG(false, false, 16)
      } else if (algo == 28) {  // no epilogue at all (timing only): what conversion + stores cost together
        MI_P8_CFG(false, false, 15)
      } else if (algo == 27) {  // epilogue woven into the MFMA segments (timing A/B)
        MI_P8_CFG(false, false, 14)
      } else if (algo == 25) {  // nt (streaming) stores
        MI_P8_CFG(false, false, 12)
      } else if (algo == 26) {  // sc1 + nt stores
        MI_P8_CFG(false, false, 13)
      } else if (algo == 24) {  // A/B baseline: half-line epilogue stores (16 rows x 64 B per instruction)
        MI_P8_CFG(false, false, 11)
      } else if (algo == 18 || algo == 19) {
        // (lab) the MX scale path without its effect: they read the block scales, so they exist for mi_gemm_mxfp8 only -- through
        // mi_gemm_fp8 the scale pointers are null and the kernel faults (it did once, from a sweep script)
        if (SA == nullptr || SB == nullptr) {
          set_error("mi_gemm: diagnostic algo %d reads MXFP8 block scales: call it through mi_gemm_mxfp8", algo);
          return MI_ERR_ARG;
        }
        if (algo == 18) {  // block scales staged into LDS but not read (unit scales): wrong results
          MI_P8_CFG(true, false, 5)
        } else {  // block scales staged and read, MFMAs still get unit scales: wrong results
          MI_P8_CFG(true, false, 6)
        }
      } else {
        set_error("mi_gemm: unknown diagnostic algo %d", algo);
        return MI_ERR_ARG;
      }
    } else {
      set_error("mi_gemm: diagnostic algo %d is built for E4M3 x E4M3 only", algo);
      return MI_ERR_ARG;
    }
  } else
#else
  if (algo == 46 || (algo >= 15 && algo <= 30)) {
    set_error("mi_gemm: algo %d is a timing / diagnostic build; it lives in the lab library (make -C llm_fp8_amd/csrc lab)", algo);
    return MI_ERR_ARG;
  } else
#endif
  if (clock_stamps != nullptr) {  // mi_gemm_fp8_clock: the production kernel's stamped build (E4M3 x E4M3 only)
    if constexpr (FA == 0 && FB == 0) {
      bias = (const uint16_t*)clock_stamps;
      MI_P8_CFG(false, false, 8)
    } else {
      set_error("mi_gemm_fp8_clock: built for E4M3 x E4M3 only");
      return MI_ERR_ARG;
    }
  } else if (mx) {
    if (bias) { MI_P8_CFG(true, true, 0) } else { MI_P8_CFG(true, false, 0) }
  } else {
    if (bias) { MI_P8_CFG(false, true, 0) } else { MI_P8_CFG(false, false, 0) }
  }
#undef MI_P8_CFG
#undef MI_P8
  MI_CHECK_LAUNCH("mi_gemm (persistent) launch");
  return MI_OK;
}

template <int FA, int FB, int OUT>
static int launch_fmt(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv,
                      const void* SA, const void* SB, const void* bias, int64_t M, int64_t N, int64_t K,
                      int64_t lda, int64_t ldb, int64_t ldd, int algo, bool mx, hipStream_t st, void* clock_stamps = nullptr) {
  const uint8_t *a = (const uint8_t*)A, *b = (const uint8_t*)B;
  const uint16_t* bp = (const uint16_t*)bias;
  if (algo == 2 && !mx) {
    int tiles_m = (int)(M / BM), tiles_n = (int)(N / BN);
    hipLaunchKernelGGL((gemm_256_2ph<FA, FB, OUT>), dim3(tiles_m * tiles_n), dim3(512), 0, st, a, b, D, sa_inv, sb_inv,
                       bp, (int)M, (int)N, (int)K, lda, ldb, ldd, tiles_m, tiles_n);
  } else if (algo == 3 && !mx) {
    int tiles_m = (int)(M / BM), tiles_n = (int)(N / BN);
    hipLaunchKernelGGL((gemm_256_8ph<FA, FB, OUT>), dim3(tiles_m * tiles_n), dim3(512), 0, st, a, b, D, sa_inv, sb_inv,
                       bp, (int)M, (int)N, (int)K, lda, ldb, ldd, tiles_m, tiles_n);
  } else if (algo == 4 || algo == 5 || (algo >= 15 && algo <= 30) || (algo >= 40 && algo <= 46)) {
    return launch_p8<FA, FB>(a, b, (uint16_t*)D, sa_inv, sb_inv, (const uint8_t*)SA, (const uint8_t*)SB, bp, M, N, K, lda, ldb, ldd,
                             algo, mx, st, clock_stamps);
  } else if (algo >= 6 && algo <= 12 && !mx) {
    // four-wave kernel (mi_gemm_w4.hip): one tile per workgroup 6 = product, 7 = no stores, 8 = clock stamps; persistent
    // 9 = product, 10 = no stores, 11 = clock stamps, 12 = no epilogue (7, 8, 10-12: lab library only)
    const int variant = algo <= 8 ? algo - 6 : algo + 1;
    if (bp != nullptr && algo != 8 && algo != 9 && algo != 11) {
      set_error("mi_gemm: of the four-wave kernels only the persistent one (algo 9) takes a bias (got algo %d)", algo);
      return MI_ERR_ARG;
    }
    if (clock_stamps != nullptr) return launch_w4(A, B, D, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, FA, FB, 12, clock_stamps, st);
    return launch_w4(A, B, D, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, FA, FB, variant, (algo == 8 || algo == 11) ? (void*)bias : nullptr, st,
                     algo == 9 ? (const void*)bp : nullptr);
#ifdef MI_DIAG
  } else if (algo >= 70 && algo <= 73 && !mx) {  // persistent four-wave kernel, epilogue store policy: 70 plain, 71 nt, 72 sc1 + nt; 73: per-K-tile stamps
    return launch_w4(A, B, D, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, FA, FB, algo - 56, algo == 73 ? (void*)bias : nullptr, st);
  } else if (algo >= 50 && algo < 70 && !mx) {  // four-wave kernel schedule sweep: 50 + 4 S + {0: product, 1: no stores, 2: stamps}
    return launch_w4(A, B, D, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, FA, FB, algo - 30, (algo % 4) == 0 ? (void*)bias : nullptr, st);
  } else if (algo == 13 && !mx) {
    int tiles_m = (int)(M / BM), tiles_n = (int)(N / BN);
    hipLaunchKernelGGL((gemm_256_8ph<FA, FB, OUT, 1>), dim3(tiles_m * tiles_n), dim3(512), 0, st, a, b, D, sa_inv, sb_inv,
                       bp, (int)M, (int)N, (int)K, lda, ldb, ldd, tiles_m, tiles_n);
  } else if (algo == 14 && !mx) {  // diagnostic: `bias` is a u64[2 * tiles] debug buffer (cycles, 100 MHz ticks)
    int tiles_m = (int)(M / BM), tiles_n = (int)(N / BN);
    hipLaunchKernelGGL((gemm_256_8ph<FA, FB, OUT, 2>), dim3(tiles_m * tiles_n), dim3(512), 0, st, a, b, D, sa_inv, sb_inv,
                       bp, (int)M, (int)N, (int)K, lda, ldb, ldd, tiles_m, tiles_n);
#endif
  } else {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    if (mx)
      hipLaunchKernelGGL((gemm_generic<FA, FB, OUT, true>), grid, dim3(256), 0, st, a, b, D, sa_inv, sb_inv,
                         (const uint8_t*)SA, (const uint8_t*)SB, bp, (int)M, (int)N, (int)K, lda, ldb, ldd);
    else
      hipLaunchKernelGGL((gemm_generic<FA, FB, OUT, false>), grid, dim3(256), 0, st, a, b, D, sa_inv, sb_inv,
                         (const uint8_t*)SA, (const uint8_t*)SB, bp, (int)M, (int)N, (int)K, lda, ldb, ldd);
  }
  MI_CHECK_LAUNCH("mi_gemm launch");
  return MI_OK;
}

static int dispatch(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, const void* SA,
                    const void* SB, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                    int64_t ldd, int fa, int fb, int out, int algo, bool mx, hipStream_t st, void* clock_stamps = nullptr) {
#define MI_CASE(FA_, FB_, OUT_)                                                                               \
  if (fa == FA_ && fb == FB_ && out == OUT_)                                                                  \
    return launch_fmt<FA_, FB_, OUT_>(A, B, D, sa_inv, sb_inv, SA, SB, bias, M, N, K, lda, ldb, ldd, algo, mx, st, clock_stamps);
  MI_CASE(0, 0, 0) MI_CASE(0, 1, 0) MI_CASE(1, 0, 0) MI_CASE(1, 1, 0)
  MI_CASE(0, 0, 1) MI_CASE(0, 1, 1) MI_CASE(1, 0, 1) MI_CASE(1, 1, 1)
#undef MI_CASE
  set_error("mi_gemm: unsupported format/out combination (%d,%d,%d)", fa, fb, out);
  return MI_ERR_ARG;
}

static int check_common(const char* who, const void* A, const void* B, void* D, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldb, int64_t ldd, int fa, int fb, int out) {
  MI_CHECK_ARG(A && B && D, "%s: null operand", who);
  MI_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "%s: negative shape", who);
  MI_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && K % 16 == 0, "%s: M,N must be multiples of 8 and K of 16 (got %lld,%lld,%lld)",
               who, (long long)M, (long long)N, (long long)K);
  MI_CHECK_ARG(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "%s: shape too large", who);
  MI_CHECK_ARG(lda >= K && ldb >= K && ldd >= N && lda % 16 == 0 && ldb % 16 == 0 && ldd % 4 == 0,
               "%s: bad leading dimensions", who);
  MI_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)D % 16) == 0,
               "%s: operands must be 16-byte aligned", who);
  MI_CHECK_ARG((fa == 0 || fa == 1) && (fb == 0 || fb == 1) && (out == 0 || out == 1), "%s: bad fmt/out_dtype", who);
  return MI_OK;
}

static int pick_algo(int algo, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int out, bool has_bias,
                     const char* who) {
  const bool fast_ok = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && M > 0 && N > 0 && K > 0;
  // persistent kernel: even K-tile count, bf16 output, 32-bit buffer offsets
  (void)has_bias;
  const bool p8_ok = (M % 256 == 0 || M % 192 == 0) && (N % 256 == 0 || N % 192 == 0) && M > 0 && N > 0 && K > 0 &&
                     (K % (2 * BK) == 0) && out == 0 && M * lda < (1LL << 31) && N * ldb < (1LL << 31) &&
                     M * ldd * 2 < (1LL << 31);
#ifndef MI_DIAG
  if (algo == 7 || algo == 8 || (algo >= 10 && algo <= 30) || algo == 46 || (algo >= 50 && algo <= 73)) {
    set_error("%s: algo %d is a timing / diagnostic build; it lives in the lab library (make -C llm_fp8_amd/csrc lab)", who, algo);
    return MI_ERR_ARG;
  }
#endif
  if (algo == 4 || algo == 5 || (algo >= 15 && algo <= 30) || (algo >= 40 && algo <= 46)) {
    if (!p8_ok) {
      set_error("%s: algo %d needs M,N %% 256 (or 192) == 0, K %% 256 == 0, bf16 output, operands < 2 GiB", who, algo);
      return MI_ERR_SHAPE;
    }
    return algo;
  }
  if ((algo >= 6 && algo <= 12) || (algo >= 50 && algo <= 73)) {
    if (!(p8_ok && M % 256 == 0 && N % 256 == 0)) {
      set_error("%s: algo %d needs M,N,K %% 256 == 0, bf16 output, operands < 2 GiB", who, algo);
      return MI_ERR_SHAPE;
    }
    return algo;
  }
  if (algo == 47) {
    // auto, four-wave kernel where it is the faster one (measured, profiles/r03_w4p_*): 256-multiples, K >= 512, and the
    // eight-wave kernel's own tile-shape choice is 256 x 256 (where it prefers 192-wide tiles the round count decides)
    if (p8_ok && M % 256 == 0 && N % 256 == 0 && K >= 512 && pick_tile_cfg(M, N, K) == 0 &&
        ((M / 256) * (N / 256) + num_cus() - 1) / num_cus() <= 64)
      return 9;
    algo = 0;
  }
  if (algo == 0) return p8_ok ? 4 : (fast_ok ? 3 : 1);
  if (algo == 1) return 1;
  if (algo == 2 || algo == 3 || algo == 13 || algo == 14) {
    if (!fast_ok) {
      set_error("%s: algo %d needs M,N %% 256 == 0 and K %% 128 == 0", who, algo);
      return MI_ERR_SHAPE;
    }
    return algo;
  }
  set_error("%s: unknown algo %d", who, algo);
  return MI_ERR_ARG;
}

}  // namespace mi

extern "C" int mi_gemm_fp8(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv,
                           const void* bias_bf16, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                           int64_t ldd, int fmt_a, int fmt_b, int out_dtype, int algo, void* stream) {
  int rc = mi::check_common("mi_gemm_fp8", A, B, D, M, N, K, lda, ldb, ldd, fmt_a, fmt_b, out_dtype);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(sa_inv && sb_inv, "mi_gemm_fp8: null scale pointer");
  if (M == 0 || N == 0) return MI_OK;
  int a = mi::pick_algo(algo, M, N, K, lda, ldb, ldd, out_dtype, bias_bf16 != nullptr, "mi_gemm_fp8");
  if (a < 0) return a;
  return mi::dispatch(A, B, D, sa_inv, sb_inv, nullptr, nullptr, bias_bf16, M, N, K, lda, ldb, ldd, fmt_a, fmt_b,
                      out_dtype, a, false, (hipStream_t)stream);
}

extern "C" int mi_gemm_fp8_clock(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, int64_t M,
                                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int algo, unsigned long long* stamps,
                                 void* stream) {
  int rc = mi::check_common("mi_gemm_fp8_clock", A, B, D, M, N, K, lda, ldb, ldd, 0, 0, 0);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(sa_inv && sb_inv && stamps, "mi_gemm_fp8_clock: null pointer");
  MI_CHECK_ARG(algo == 0 || algo == 4 || algo == 9, "mi_gemm_fp8_clock: algo must be 0, 4 or 9");
  if (M == 0 || N == 0) return MI_OK;
  int a = mi::pick_algo(algo, M, N, K, lda, ldb, ldd, 0, false, "mi_gemm_fp8_clock");
  if (a < 0) return a;
  if (a != 4 && a != 9) {
    mi::set_error("mi_gemm_fp8_clock: the shape does not run on a persistent kernel");
    return MI_ERR_SHAPE;
  }
  return mi::dispatch(A, B, D, sa_inv, sb_inv, nullptr, nullptr, nullptr, M, N, K, lda, ldb, ldd, 0, 0, 0, a, false,
                      (hipStream_t)stream, (void*)stamps);
}

extern "C" int mi_gemm_mxfp8(const void* A, const void* SA, const void* B, const void* SB, void* D,
                             const void* bias_bf16, int64_t M, int64_t N, int64_t K, int fmt_a, int fmt_b,
                             int out_dtype, int algo, void* stream) {
  int rc = mi::check_common("mi_gemm_mxfp8", A, B, D, M, N, K, K, K, N, fmt_a, fmt_b, out_dtype);
  if (rc != MI_OK) return rc;
  MI_CHECK_ARG(SA && SB, "mi_gemm_mxfp8: null scale pointer");
  MI_CHECK_ARG(K % 32 == 0, "mi_gemm_mxfp8: K must be a multiple of 32");
#ifdef MI_DIAG
  MI_CHECK_ARG(algo == 0 || algo == 1 || algo == 4 || algo == 5 || algo == 18 || algo == 19 || (algo >= 40 && algo <= 45), "mi_gemm_mxfp8: algo must be 0, 1, 4, 5, 18, 19 or 40-45");
#else
  MI_CHECK_ARG(algo == 0 || algo == 1 || algo == 4 || algo == 5 || (algo >= 40 && algo <= 45), "mi_gemm_mxfp8: algo must be 0, 1, 4, 5 or 40-45");
#endif
  if (M == 0 || N == 0) return MI_OK;
  int a = algo == 0 ? 4 : algo;
  if (a != 1) {
    a = mi::pick_algo(a, M, N, K, K, K, N, out_dtype, bias_bf16 != nullptr, "mi_gemm_mxfp8");
    if (a < 0) {
      if (algo != 0) return a;
      a = 1;  // auto: fall back to the generic kernel
    }
  }
  return mi::dispatch(A, B, D, nullptr, nullptr, SA, SB, bias_bf16, M, N, K, K, K, N, fmt_a, fmt_b, out_dtype, a, true,
                      (hipStream_t)stream);
}

extern "C" int64_t mi_gemm_workspace_bytes(void) { return 4096 + (int64_t)mi::num_cus() * (8 * 32 * 1024); }

extern "C" int mi_gemm_set_workspace(void* workspace, int64_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) {
    mi::set_error("mi_gemm_set_workspace: no current device");
    return MI_ERR_HIP;
  }
  if (workspace == nullptr) {
    mi::g_sk_ws[dev] = mi::SkWorkspace();
    return MI_OK;
  }
  MI_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "mi_gemm_set_workspace: pointer must be 256-byte aligned");
  MI_CHECK_ARG(bytes >= 4096 + 8 * 32 * 1024, "mi_gemm_set_workspace: %lld bytes is too small", (long long)bytes);
  mi::SkWorkspace w;
  w.flags = (unsigned int*)workspace;
  w.slots = (float*)((char*)workspace + 4096);
  w.nslots = (int)((bytes - 4096) / (8 * 32 * 1024));
  if (w.nslots > 1024) w.nslots = 1024;
  mi::g_sk_ws[dev] = w;
  return MI_OK;
}
