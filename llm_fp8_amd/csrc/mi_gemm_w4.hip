// gemm_w4: FP8 x FP8 -> bf16 "TN" GEMM, 256x256x128 workgroup tile, FOUR waves (one per SIMD), 128x128 wave tiles.
//
// Why a second main loop (round 3): the chip is power-limited and the LDS->VGPR fragment traffic is a third of
// the dynamic energy per MFMA.  A 128x128 wave tile needs 16 fragments per 64 MFMAs where the eight-wave kernel's 128x64
// wave tile needs 12 per 32 (-33 % LDS bytes per flop; tools/probe_wave_tile.hip: +6.5 % at a higher held clock).  With one
// wave per SIMD there is no partner wave to hide LDS latency, LDS-DMA issue and barriers behind, so the wave hides them
// behind its OWN MFMAs: an MFMA keeps the matrix pipe busy for 32 cycles and the wave's issue port for ~8, the rest is shadow.
//   * accumulators: 64 MFMA tiles = 256 registers (AGPRs); fragments are read ONE PHASE AHEAD into registers
//     (aA, aB: the two A halves; b0[2]: B half 0, double-buffered; b1: B half 1 = 160 VGPRs);
//   * a K-tile is 4 phases of 16 MFMAs (wave quadrants A0.B0, A0.B1, A1.B1, A1.B0), each phase also issues 8 ds_read_b128
//     (the 4 fragments the NEXT phase needs first) and 4 LDS-DMA (one half-tile of 16 KiB per workgroup and phase);
//   * one barrier per phase, issued right after the phase's first MFMA (its wait is the MFMA's shadow).
// LDS: 2 K-tile buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 128 B, the eight-wave kernel's swizzled image.
// Half-tile X(t) is read (into registers) in ONE phase and restaged with X(t + 2) in the following one:
//     phase      MFMAs            reads (for)            stages
//     p0(t)   aA . b0[t&1]     b1 <- B1(t)   (p1)     B0(t + 2)
//     p1(t)   aA . b1          aB <- A1(t)   (p2)     B1(t + 2)
//     p2(t)   aB . b1          aA <- A0(t+1) (p0')    A1(t + 2)
//     p3(t)   aB . b0[t&1]     b0[~t&1] <- B0(t+1)    A0(t + 3)
// so a stage has 7 phases (~3.5k cycles) to land; the wait before a phase's barrier is a uniform vmcnt(24): the half read
// in this phase was staged 7 phases ago, the 6 younger stages (4 DMA each per wave) may stay in flight.
//
// Replaces the same TE cuBLASLt FP8 GEMMs as mi_gemm.hip (te_llama.py:45-63,76-80).
#include "mi_gemm_dev.h"
#include <type_traits>

namespace mi {

// c += a . b with the accumulator pinned to the AGPR file ("+a"): the 64 accumulator tiles of a wave are exactly the 256
// AGPRs; through the builtin hipcc keeps 8 of them in VGPRs and copies each through a[4:7] around every MFMA (80 v_accvgpr
// moves + hazard nops per K-tile pair).  Unit scales (per-tensor path).
template <int FA, int FB>
__device__ __forceinline__ void mfma_acc(const v8i& a, const v8i& b, v4f& acc, int unit) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:%4 blgp:%5"
               : "+a"(acc)
               : "v"(b), "v"(a), "v"(unit), "n"(FB), "n"(FA));
}

// ABL: 0 = product, 1 = no stores (timing), 2 = clock stamps into `dbg` (u64[4 * grid]: cycles, 100 MHz ticks, K-tiles, xcc)
template <int FA, int FB, int ABL>
__global__ __launch_bounds__(256, 1) void gemm_w4(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                  uint16_t* __restrict__ D, const float* __restrict__ sa_inv,
                                                  const float* __restrict__ sb_inv, int K, int lda, int ldb, int ldd,
                                                  int tiles_m, int tiles_n, int a_bytes, int b_bytes, int d_bytes,
                                                  unsigned long long* __restrict__ dbg) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
  const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
  const int oa = tm * 256 * lda, ob = tn * 256 * ldb;
  const int nk = K / BK, last = nk - 1;

  // LDS-DMA: half h of an operand = the rows 128 w' + 64 h + [0, 64) of the two wave rows (columns) w' -- a row GATHER, free
  // with per-lane source addresses.  Wave w stages pieces 4w .. 4w+3 (8 rows each) of every half: local rows 32 w + 8 p + lr.
  int a_v, b_v;
  {
    const int lr = lane >> 3, lc = lane & 7;
    const int chunk = (lc ^ swz_f(lr)) * 16;
    const int row = (wave >> 1) * 128 + (wave & 1) * 32 + lr;
    a_v = row * lda + chunk;
    b_v = row * ldb + chunk;
  }
  uint8_t* const buf0 = lds;
  uint8_t* const buf1 = lds + kBufBytes;
  auto dma = [&](rsrc_t rs, uint8_t* dst, int voff, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, voff, soff, 0, 0);
  };
  // one of the 4 DMA of a half-tile stage: piece p of (operand, half h) of K-tile kt into `buf`
  auto stage_a = [&](int h, int kt, uint8_t* buf, int p) __attribute__((always_inline)) {
    dma(rsA, buf + (h ? kOffA1 : kOffA0) + (wave * 4 + p) * 1024, a_v, oa + min(kt, last) * BK + (h * 64 + p * 8) * lda);
  };
  auto stage_b = [&](int h, int kt, uint8_t* buf, int p) __attribute__((always_inline)) {
    dma(rsB, buf + (h ? kOffB1 : kOffB0) + (wave * 4 + p) * 1024, b_v, ob + min(kt, last) * BK + (h * 64 + p * 8) * ldb);
  };

  // fragment read bases (per lane): [buffer] x {A, B} x {chunk q, chunk 4 + q}; halves and fragments are immediate offsets
  int fa_lo[2], fa_hi[2], fb_lo[2], fb_hi[2];
  {
    const int r = lane & 15, q = lane >> 4;
    const int frow = (r >> 3) * 1024 + (r & 7) * 128;
    const int flo = frow + ((q ^ swz_f(r)) << 4), fhi = frow + (((4 + q) ^ swz_f(r)) << 4);
    const int l0 = (int)(size_t)LDS_PTR(lds);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      fa_lo[b] = l0 + b * kBufBytes + kOffA0 + wr * 4 * 2048 + flo;
      fa_hi[b] = l0 + b * kBufBytes + kOffA0 + wr * 4 * 2048 + fhi;
      fb_lo[b] = l0 + b * kBufBytes + kOffB0 + wc * 4 * 2048 + flo;
      fb_hi[b] = l0 + b * kBufBytes + kOffB0 + wc * 4 * 2048 + fhi;
    }
  }
  typedef const __attribute__((address_space(3))) v4i* lds_v4i_p;
  // n = 0..7: fragment n >> 1, 16-byte part n & 1 (registers 0-3 / 4-7 of the fragment)
  auto read_part = [&](v8i (&dst)[4], int lo, int hi, int half, int n) __attribute__((always_inline)) {
    const int f = n >> 1;
    const v4i v = *reinterpret_cast<lds_v4i_p>((size_t)(unsigned)(((n & 1) ? hi : lo) + half * kHalfBytes + f * 2048));
    if (n & 1) dst[f].hi = v;
    else dst[f].lo = v;
  };

  v4f acc[2][2][4][4];  // [A half][B half][fragment i][fragment j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][b][i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  v8i aA[4], aB[4], b0[2][4], b1[4];
  int unit = kUnitScale;
  asm volatile("" : "+v"(unit));

  // One phase: 16 MFMAs c[i][j] += af[i] . bf[j]; after MFMA 0 the wait + barrier; after MFMAs 1-4 one LDS-DMA each; after
  // MFMAs 4-11 one ds_read_b128 each.  `wait_lgkm`: the fragments read in the previous phase have landed (they were issued
  // >= 4 MFMAs ago) -- also the WAR guarantee for the stage that follows the barrier.
  auto phase = [&](const v8i (&af)[4], const v8i (&bf)[4], v4f (&c)[4][4], auto stage, auto read) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = k >> 2, j = k & 3;
      mfma_acc<FA, FB>(af[i], bf[j], c[i][j], unit);
      if (k == 0) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (k >= 1 && k <= 4) stage(k - 1);
      if (k >= 4 && k <= 11) read(k - 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto ktile = [&](auto par_c, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    uint8_t* const cur = par ? buf1 : buf0;
    uint8_t* const oth = par ? buf0 : buf1;
    phase(aA, b0[par], acc[0][0], [&](int p) __attribute__((always_inline)) { stage_b(0, t + 2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(b1, fb_lo[par], fb_hi[par], 1, n); });
    phase(aA, b1, acc[0][1], [&](int p) __attribute__((always_inline)) { stage_b(1, t + 2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(aB, fa_lo[par], fa_hi[par], 1, n); });
    phase(aB, b1, acc[1][1], [&](int p) __attribute__((always_inline)) { stage_a(1, t + 2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(aA, fa_lo[par ^ 1], fa_hi[par ^ 1], 0, n); });
    phase(aB, b0[par], acc[1][0], [&](int p) __attribute__((always_inline)) { stage_a(0, t + 3, oth, p); },
          [&](int n) __attribute__((always_inline)) { read_part(b0[par ^ 1], fb_lo[par ^ 1], fb_hi[par ^ 1], 0, n); });
  };

  // prologue: the stage sequence of the virtual phases before p0(0), then the two read-only phases p2(-1), p3(-1)
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_a(0, 0, buf0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_b(0, 0, buf0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_b(1, 0, buf0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_a(1, 0, buf0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_a(0, 1, buf1, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_b(0, 1, buf1, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_b(1, 1, buf1, p);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_a(1, 1, buf1, p);
#pragma unroll
  for (int n = 0; n < 8; ++n) read_part(aA, fa_lo[0], fa_hi[0], 0, n);
  asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 4; ++p) stage_a(0, 2, buf0, p);
#pragma unroll
  for (int n = 0; n < 8; ++n) read_part(b0[0], fb_lo[0], fb_hi[0], 0, n);
  __builtin_amdgcn_sched_barrier(0);

  unsigned long long c0 = 0, r0 = 0;
  if (ABL == 2) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  using par0_t = std::integral_constant<int, 0>;
  using par1_t = std::integral_constant<int, 1>;
  for (int t = 0; t < nk; t += 2) {
    ktile(par0_t{}, t);
    ktile(par1_t{}, t + 1);
  }
  if (ABL == 2) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* o = dbg + (size_t)blockIdx.x * 4;
      o[0] = c1 - c0;
      o[1] = r1 - r0;
      o[2] = (unsigned long long)nk;
      o[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // dead tail prefetches retired before the LDS is released

  const float alpha = (*sa_inv) * (*sb_inv);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t m = (int64_t)tm * 256 + wr * 128 + a * 64 + i * 16 + fr;
          const int64_t n = (int64_t)tn * 256 + wc * 128 + b * 64 + j * 16 + fq * 4;
          const v4f v = acc[a][b][i][j] * alpha;
          if (ABL != 0) {
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          } else {
            const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            *reinterpret_cast<uint2*>(D + m * ldd + n) = pk;
          }
        }
}

template <int FA, int FB>
static int launch_w4_fmt(const uint8_t* a, const uint8_t* b, uint16_t* D, const float* sa_inv, const float* sb_inv, int64_t M,
                         int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int variant, void* dbg, hipStream_t st) {
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
  const dim3 grid(tiles_m * tiles_n), block(256);
#define MI_W4(ABLv)                                                                                                      \
  hipLaunchKernelGGL((gemm_w4<FA, FB, ABLv>), grid, block, 0, st, a, b, D, sa_inv, sb_inv, (int)K, (int)lda, (int)ldb,    \
                     (int)ldd, tiles_m, tiles_n, (int)(M * lda), (int)(N * ldb), (int)(M * ldd * 2), (unsigned long long*)dbg)
  if (variant == 0) MI_W4(0);
  else if (variant == 1) MI_W4(1);
  else MI_W4(2);
#undef MI_W4
  MI_CHECK_LAUNCH("mi_gemm (w4) launch");
  return MI_OK;
}

// variant: 0 = product, 1 = no stores, 2 = clock stamps (dbg = u64[4 * tiles]).  Shapes: M, N % 256 == 0, K % 256 == 0,
// operands below 2 GiB (the dispatcher in mi_gemm.hip checks).
int launch_w4(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldd, int fa, int fb, int variant, void* dbg, hipStream_t st) {
  const uint8_t *a = (const uint8_t*)A, *b = (const uint8_t*)B;
  uint16_t* d = (uint16_t*)D;
  if (fa == 0 && fb == 0) return launch_w4_fmt<0, 0>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st);
  if (fa == 0 && fb == 1) return launch_w4_fmt<0, 1>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st);
  if (fa == 1 && fb == 0) return launch_w4_fmt<1, 0>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st);
  return launch_w4_fmt<1, 1>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st);
}

}  // namespace mi
