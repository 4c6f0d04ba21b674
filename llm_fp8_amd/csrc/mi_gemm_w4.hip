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
#include <utility>

namespace mi {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) -- indices that must be constant
// expressions for the FRONT END (register arrays indexed by a loop variable of a big unrolled nest end up in scratch memory)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// c += a . b with the accumulator pinned to the AGPR file ("+a"): the 64 accumulator tiles of a wave are exactly the 256
// AGPRs; through the builtin hipcc keeps 8 of them in VGPRs and copies each through a[4:7] around every MFMA (80 v_accvgpr
// moves + hazard nops per K-tile pair).  Unit scales (per-tensor path).
template <int FA, int FB>
__device__ __forceinline__ void mfma_acc(const v8i& a, const v8i& b, v4f& acc, int unit) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:%4 blgp:%5"
               : "+a"(acc)
               : "v"(b), "v"(a), "v"(unit), "n"(FB), "n"(FA));
}

// Slot tables of a phase (16 MFMA slots), by schedule id S -- lab sweep (tools/bench_w4.py --which sched):
//   kb   the wait + barrier follow MFMA kb
//   dm   LDS-DMA p of the phase's stage follows MFMA dm(p)      (> kb: the stage overwrites a half other waves were reading)
//   rd   ds_read_b128 n follows MFMA rd(n)                     (> = kb: the half read was staged by other waves too)
//   lag  1: a half read in phase P is restaged in P + 1 (7 phases to land, the wait must cover the reads of P: lgkmcnt(0));
//        2: restaged in P + 2 (6 phases to land; the reads of P - 1 may still be in flight at the barrier: lgkmcnt(8))
namespace w4 {
constexpr int sched_lag(int S) { return (S == 3 || S == 4) ? 2 : 1; }
constexpr int sched_kb(int S) { return 0; }
constexpr int sched_dm(int S, int p) { return S == 1 ? 1 + 4 * p : S == 3 ? 2 + 4 * p : 1 + p; }
constexpr int sched_rd(int S, int n) { return S == 3 ? 1 + 2 * n : 4 + n; }
}  // namespace w4

// ABL: 0 = product, 1 = no stores (timing), 2 = clock stamps into `dbg` (u64[4 * grid]: cycles, 100 MHz ticks, K-tiles, xcc)
template <int FA, int FB, int ABL, int S = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                  uint16_t* __restrict__ D, const float* __restrict__ sa_inv,
                                                  const float* __restrict__ sb_inv, int K, int lda, int ldb, int ldd,
                                                  int tiles_m, int tiles_n, int a_bytes, int b_bytes, int d_bytes,
                                                  unsigned long long* __restrict__ dbg) {
  constexpr int LAG = w4::sched_lag(S);
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

  // One phase: 16 MFMAs c[i][j] += af[i] . bf[j] with the wait + barrier, the stage's 4 LDS-DMA and the 8 ds_read_b128 of the
  // fragments the next phase needs first woven in by the slot tables of schedule S.
  constexpr int kVm = LAG == 1 ? 24 : 20;
  auto phase = [&](const v8i (&af)[4], const v8i (&bf)[4], v4f (&c)[4][4], auto stage, auto read) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = k >> 2, j = k & 3;
      mfma_acc<FA, FB>(af[i], bf[j], c[i][j], unit);
      if (k == w4::sched_kb(S)) {
        __builtin_amdgcn_sched_barrier(0);
        // RAW: the half this phase reads has landed (own pieces; the barrier covers the other waves').  WAR: the half this phase
        // restages is no longer being read -- LAG 1: it was read in the previous phase (all LDS reads retired); LAG 2: two
        // phases ago (LDS reads retire in order: the 8 of the previous phase may stay in flight)
        if (LAG == 1) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(20) lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (k == w4::sched_dm(S, p)) stage(p);
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (k == w4::sched_rd(S, n)) read(n);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto ktile = [&](auto par_c, int t) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    uint8_t* const cur = par ? buf1 : buf0;
    uint8_t* const oth = par ? buf0 : buf1;
    if (LAG == 1) {
      phase(aA, b0[par], acc[0][0], [&](int p) __attribute__((always_inline)) { stage_b(0, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(b1, fb_lo[par], fb_hi[par], 1, n); });
      phase(aA, b1, acc[0][1], [&](int p) __attribute__((always_inline)) { stage_b(1, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(aB, fa_lo[par], fa_hi[par], 1, n); });
      phase(aB, b1, acc[1][1], [&](int p) __attribute__((always_inline)) { stage_a(1, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(aA, fa_lo[par ^ 1], fa_hi[par ^ 1], 0, n); });
      phase(aB, b0[par], acc[1][0], [&](int p) __attribute__((always_inline)) { stage_a(0, t + 3, oth, p); },
            [&](int n) __attribute__((always_inline)) { read_part(b0[par ^ 1], fb_lo[par ^ 1], fb_hi[par ^ 1], 0, n); });
    } else {  // every stage of K-tile t goes to step t + 2, i.e. into this K-tile's own buffer
      phase(aA, b0[par], acc[0][0], [&](int p) __attribute__((always_inline)) { stage_a(0, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(b1, fb_lo[par], fb_hi[par], 1, n); });
      phase(aA, b1, acc[0][1], [&](int p) __attribute__((always_inline)) { stage_b(0, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(aB, fa_lo[par], fa_hi[par], 1, n); });
      phase(aB, b1, acc[1][1], [&](int p) __attribute__((always_inline)) { stage_b(1, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(aA, fa_lo[par ^ 1], fa_hi[par ^ 1], 0, n); });
      phase(aB, b0[par], acc[1][0], [&](int p) __attribute__((always_inline)) { stage_a(1, t + 2, cur, p); },
            [&](int n) __attribute__((always_inline)) { read_part(b0[par ^ 1], fb_lo[par ^ 1], fb_hi[par ^ 1], 0, n); });
    }
  };

  // prologue: the stage sequence of the virtual phases before p0(0), then the two read-only phases p2(-1), p3(-1)
  auto stage4a = [&](int h, int kt, uint8_t* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) stage_a(h, kt, buf, p);
  };
  auto stage4b = [&](int h, int kt, uint8_t* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) stage_b(h, kt, buf, p);
  };
  stage4a(0, 0, buf0);
  stage4b(0, 0, buf0);
  stage4b(1, 0, buf0);
  stage4a(1, 0, buf0);
  stage4a(0, 1, buf1);
  stage4b(0, 1, buf1);
  if (LAG == 1) {
    stage4b(1, 1, buf1);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage4a(1, 1, buf1);
#pragma unroll
    for (int n = 0; n < 8; ++n) read_part(aA, fa_lo[0], fa_hi[0], 0, n);
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage4a(0, 2, buf0);
  } else {
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage4b(1, 1, buf1);
#pragma unroll
    for (int n = 0; n < 8; ++n) read_part(aA, fa_lo[0], fa_hi[0], 0, n);
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage4a(1, 1, buf1);
  }
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

// ------------------------------------------------------------------------------------------------
// gemm_w4p: the persistent form.  One workgroup per CU walks its tiles; the (tile, K-tile) steps form ONE pipeline (stages run
// two steps ahead across tile boundaries; phase schedule = S3 of the sweep above: stage lag 2, reads on the odd slots, one
// LDS-DMA every fourth slot), and a tile's epilogue is spread over the TEN phases around the boundary, in the shadow of the
// wave's own MFMAs.  Frame of a tile boundary: MFMA slot g = 16 x phase + k, phase 0 = p0 of the tile's LAST K-tile.
//   * a wave's 128x128 output = 4 quadrants (A half, B half) x 4 fragment rows (16 rows x 64 columns = 4 accumulator tiles);
//     quadrant order Q00, Q01, Q11, Q10 = the order in which the LAST K-tile finishes them (after p0, p1, p2, p3) and in which
//     the next tile's FIRST K-tile (zero-C MFMAs) overwrites them (in p0', p1', p2', p3');
//   * what fits in the shadow of a wave's own MFMA was MEASURED (tools/probe_mfma_shadow.hip, profiles/r03_mfma_shadow_probe.txt):
//     the MFMA holds the issue port ~16 of its 32 cycles, so ~4 plain VALU per slot are free and every further one costs ~4.3
//     cycles; v_pk_mul_f32 costs 12 (never: scalar v_mul_f32 through inline asm, which the SLP vectoriser cannot fuse); a
//     1-KiB store blocks its wave ~47 cycles; and ANY branch ~31 cycles (one wave per SIMD: nobody covers the instruction-
//     buffer refill) -- so the K-tiles are branch-free and all four waves store in the same slots (a per-wave stagger behind
//     wave-uniform branches was built and measured: the 128 branches per tile cost more than the store collisions they avoid,
//     profiles/r03_w4p_v2_staggered_stores_ab.txt);
//   * fragment row r = 5 chunks in consecutive slots from conv_start(r): 4 x {read 4 accumulators, x alpha, 2 cvt_pk_bf16}
//     (+ the permlane16 swap that widens two tiles to 8 columns per lane after every second), then the DPP row exchange that
//     makes whole 128-byte lines; the packed row (8 registers) is PARKED in one of 5 register slots until its stores come up.
//     Rows 0-7 are converted at the pace of the stores, rows 8-15 back to back so that every accumulator is read before the
//     zero-C MFMA that overwrites it (w4::schedule_ok checks every deadline at compile time);
//   * stores: four waves storing in lockstep collide on the CU's one store path, so the workgroup's 128 stores of a tile are
//     dealt ONE PER MFMA SLOT: store event e belongs to slot 24 + 4 e + w for wave w, behind a wave-uniform branch (the only
//     branches inside the K-tiles).  Last store: slot 151 = p1 of the next tile's SECOND K-tile.
//   Four epilogue schedules were built and measured against each other and against the eight-wave kernel on the twelve 3B decoder
//   shapes (profiles/r03_w4p_*): this one (kernel time 0.976 x the eight-wave kernel's, 0.95 on the shapes where both run 256 x 256
//   tiles), lockstep stores without any branch (1.005), and two finer micro-op schedules (1.005 / 1.007) -- per-K-tile stamps
//   (profiles/r03_w4p_v3_ktile_stamps.txt) show what all of them share: the four boundary K-tiles cost ~7 k cycles more than four
//   plain ones, because the wave's issue port is the bottleneck at one wave per SIMD and the epilogue is ~830 VALU instructions and
//   32 stores per wave and tile whatever their order.
// vmcnt retires in order, so a phase's wait also allows the stores issued during the five phases before it (wait_x).
// The FIRST / SECOND K-tiles of a workgroup's first tile run the same code on zero accumulators and store through a buffer
// descriptor of zero records (the address check drops the stores; they still count in vmcnt, so the waits stay uniform): a
// branch with MFMAs in both arms makes hipcc merge the 256 accumulator registers through scratch, and a branch around every
// chunk spills fragments (reloaded behind `s_waitcnt vmcnt(0)`).  K >= 512 (at least FIRST, SECOND, THIRD, LAST).
namespace w4 {
constexpr int kPark = 5;
constexpr int conv_start(int r) { return r < 8 ? 19 + 8 * r : 80 + 5 * (r - 8); }  // 5 chunk slots: T0, T1 + swap, T2, T3 + swap, DPP
constexpr int store_slot(int e) { return 24 + 4 * e; }    // store event e = (row e >> 1, line half e & 1): slot store_slot(e) + wave
constexpr int kFrameSlots = store_slot(31) + 4;           // first slot past the last store
constexpr int stores_in_phase(int gp) {                   // per wave
  int n = 0;
  for (int e = 0; e < 32; ++e)
    if (store_slot(e) / 16 == gp) ++n;
  return n;
}
// stores a wave issued in the phases G-5 .. G-1 (the stage awaited at phase G was issued in phase G-6)
constexpr int wait_x(int G) {
  int n = 0;
  for (int q = G - 5; q < G; ++q)
    if (q >= 0) n += stores_in_phase(q);
  return n;
}
enum { MID = -1, LAST = 0, FIRST = 4, SECOND = 8, THIRD = 12 };  // value = frame phase of the K-tile's p0
constexpr bool schedule_ok() {
  for (int r = 0; r < 16; ++r) {
    const int q = r / 4, i = r % 4;
    if (conv_start(r) < 16 + 16 * q) return false;                   // the quadrant is final after phase q of the LAST K-tile
    if (conv_start(r) + 3 >= 64 + 16 * q + 4 * i + 3) return false;  // tile j is read in slot conv_start + j, before the zero-C MFMA of slot 64 + 16 q + 4 i + j
    if (store_slot(2 * r) < conv_start(r) + 5) return false;         // the row is complete before its first store
    if (r + kPark < 16 && conv_start(r + kPark) + 1 <= store_slot(2 * r + 1) + 3) return false;  // parking slot free again
  }
  return kFrameSlots <= 160;                                         // the frame ends inside the SECOND K-tile
}
static_assert(schedule_ok(), "epilogue schedule violates a deadline");
}  // namespace w4

// zero-C form of mfma_acc (first K-tile of a tile): the accumulator stays tied ("+a") so the allocator keeps it in place
template <int FA, int FB>
__device__ __forceinline__ void mfma_acc_zero(const v8i& a, const v8i& b, v4f& acc, int unit) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0] cbsz:%4 blgp:%5"
               : "+a"(acc)
               : "v"(b), "v"(a), "v"(unit), "n"(FB), "n"(FA));
}

// Kernel arguments of the one-problem forms
struct W4Single {
  const uint8_t* A;
  const uint8_t* B;
  uint16_t* D;
  const float* sa_inv;
  const float* sb_inv;
  const uint16_t* bias;  // MODE 1: bf16 [n_cols]
  unsigned long long* dbg;
  int K, lda, ldb, ldd, tiles_m, tiles_n, a_bytes, b_bytes, d_bytes, n_cols;
};

// MODE: 0 = one problem; 1 = one problem + bias (bf16 [N], added after the alpha multiply as the eight-wave kernel does: two
// roundings); 2 = grouped: up to 4 problems in one launch, tiles dealt by the host's longest-processing-time schedule
// (GroupArgs, mi_gemm_grouped.hip) -- every per-problem quantity (descriptors, leading dimensions, K-tile count, alpha, output) is
// carried per stage cursor / per epilogue frame and switched at tile boundaries.
// ABL: 0 = product, 1 = no stores, 2 = clock stamps (dbg u64[4 * grid]), 3 = no epilogue at all (timing: wrong results),
// 4 / 5 / 6 = plain (write-back) / nt / sc1+nt epilogue stores instead of sc1 (timing A/B of the store policy),
// 7 = workgroup 0 stamps s_memtime at the start of every K-tile (dbg u64[1024]: the per-K-tile timeline around tile boundaries)
template <int FA, int FB, int ABL, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_w4p(const std::conditional_t<MODE == 2, GroupArgs, W4Single> ka) {
  constexpr bool GROUPED = MODE == 2, BIAS = MODE == 1;
  constexpr int S = 3;  // phase schedule (w4::sched_*)
  static_assert(w4::sched_lag(S) == 2, "the cursor logic below assumes every stage of step s goes to step s + 2");
  static_assert(!(GROUPED && ABL != 0), "the grouped form has no timing builds");
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes + (ABL == 7 ? 8192 : 0)];
  int stamp_idx = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int G = gridDim.x, bid = blockIdx.x;

  // ---- this workgroup's tiles: lane i of `tab` keeps tile row | tile column << 14 | problem << 28 of its i-th tile (<= 64)
  int my_tiles = 0, tab = 0;
  float alpha_of[kMaxGroup] = {1.f, 1.f, 1.f, 1.f};  // read ONCE: a memory read inside the tile walk would drain the LDS-DMA pipeline
  if constexpr (GROUPED) {
    // virtual index: XCD-major order of the workgroups (round-robin dispatch over the 8 XCDs: bid & 7 names the L2)
    const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int my_cnt[kMaxGroup];
#pragma unroll
    for (int q = 0; q < kMaxGroup; ++q) {
      my_cnt[q] = q < ka.n ? (int)ka.cnt[q][v] : 0;
      my_tiles += my_cnt[q];
      float a = 1.0f;
      if (q < ka.n) a = (*ka.p[q].sa_inv) * (*ka.p[q].sb_inv);
      alpha_of[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
    }
    // pass 1 (uniform loop over my tiles): the tile's id within its problem = R[p][j] + rank of v among the workgroups that have a
    // j-th tile of p (ballots over the 256-entry count table); pass 2 (one vector evaluation): id -> (tm, tn) in the grouped order
    int my_id = 0, my_p = 0, ti = 0;
#pragma unroll
    for (int q = 0; q < kMaxGroup; ++q) {
      if (q < ka.n) {
        const unsigned c4 = reinterpret_cast<const unsigned*>(ka.cnt[q])[lane];  // counts of virtual workgroups 4 l .. 4 l + 3
        for (int j = 0; j < my_cnt[q]; ++j, ++ti) {
          int rank = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool f = (int)((c4 >> (8 * k)) & 255u) > j && (4 * lane + k) < v;
            rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
          }
          if (lane == ti) {
            my_id = (int)ka.R[q][j] + rank;
            my_p = q;
          }
        }
      }
    }
    int tmn = ka.p[0].tiles_m, tnn = ka.p[0].tiles_n;
#pragma unroll
    for (int q = 1; q < kMaxGroup; ++q) {
      const bool is = my_p == q;
      tmn = is ? ka.p[q].tiles_m : tmn;
      tnn = is ? ka.p[q].tiles_n : tnn;
    }
    int tm, tn;
    tile_of_flat(my_id, max(tmn, 1), max(tnn, 1), tm, tn);
    tab = tm | (tn << 14) | (my_p << 28);
  } else {
    const int ntiles = ka.tiles_m * ka.tiles_n;
    my_tiles = (ntiles - bid + G - 1) / G;  // tiles bid, bid + G, ... (host: <= 64 per workgroup)
    int tm, tn;
    tile_of_block(bid + min(lane, max(my_tiles - 1, 0)) * G, ntiles, ka.tiles_m, ka.tiles_n, tm, tn);
    tab = tm | (tn << 14);
    alpha_of[0] = (*ka.sa_inv) * (*ka.sb_inv);
  }
  auto tile_of = [&](int ti, int& p, int& ra, int& rb) __attribute__((always_inline)) {
    const int t = __builtin_amdgcn_readlane(tab, ti);
    p = (t >> 28) & 3;
    ra = (t & 0x3FFF) * 256;
    rb = ((t >> 14) & 0x3FFF) * 256;
  };

  uint8_t* const buf0 = lds;
  uint8_t* const buf1 = lds + kBufBytes;
  // The four pieces p = 0..3 of a stage group land 1 KiB apart in LDS: ONE M0 value (the group's base) + the instruction's immediate
  // offset p * 1024 instead of an s_mov m0 (and its hazard s_nop) in front of every DMA -- 12 scalar instructions of the MFMA wave's own
  // issue port per K-tile.  The immediate shifts the GLOBAL address as well, so the scalar offset carries - p * 1024 (never below zero:
  // it holds at least p * 8 * ld with ld >= 512); the descriptors are 4 KiB longer than the operands because the range check adds
  // the immediate to the per-lane offset (the scalar offset is outside the check anyway).
  auto dma = [&](rsrc_t rs, uint8_t* dst_base, int voff, int soff, auto p_c) __attribute__((always_inline)) {
    constexpr int IMM = decltype(p_c)::value * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst_base), 16, voff, soff - IMM, IMM, 0);
  };
  // stage cursor: wave-uniform (problem, tile origin, K offset) of step s + 2, clamped to the last step.  a_v / b_v: per-lane
  // byte offset row * ld + chunk of the wave's first piece (they depend on the problem's leading dimensions)
  struct cursor_t {
    int oa, ob, kb, ti, kt, nk, lda, ldb, a_v, b_v, step;
    rsrc_t rsA, rsB;
  };
  auto enter_tile = [&](cursor_t& c, int ti) __attribute__((always_inline)) {
    int p, ra, rb;
    tile_of(ti, p, ra, rb);
    if constexpr (GROUPED) {
      const GroupProblem& P = ka.p[p];
      c.nk = P.nk;
      c.lda = P.lda;
      c.ldb = P.ldb;
      c.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)((unsigned)P.a_bytes + 4096u), 0x00020000);
      c.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)((unsigned)P.b_bytes + 4096u), 0x00020000);
      int ln;
      asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));  // (not hoisted: nothing of this stays live across the K loop)
      const int lr = ln >> 3, chunk = ((ln & 7) ^ swz_f(lr)) * 16;
      const int row = (wave >> 1) * 128 + (wave & 1) * 32 + lr;
      c.a_v = row * P.lda + chunk;
      c.b_v = row * P.ldb + chunk;
    }
    c.oa = ra * c.lda;
    c.ob = rb * c.ldb;
  };
  int total = 0;  // steps of this workgroup (one-problem forms)
  auto advance = [&](cursor_t& c) __attribute__((always_inline)) {
    if constexpr (GROUPED) {
      // the tile switch reloads the problem's descriptors: a wave-uniform branch, taken once per tile (no MFMA inside)
      const bool wrap = c.kt + 1 == c.nk;
      if (wrap && c.ti + 1 < my_tiles) {
        c.ti += 1;
        c.kt = 0;
        c.kb = 0;
        enter_tile(c, c.ti);
      } else {
        const int inc = wrap ? 0 : 1;
        c.kt += inc;
        c.kb += inc * BK;
      }
    } else {
      // branch-free (a branch costs ~31 cycles at one wave per SIMD, and the cursor moves once per K-tile): selects only
      const int more = (c.step + 1 < total) ? 1 : 0;
      c.step += more;
      const int kt1 = c.kt + more;
      const int wrap = (kt1 == c.nk) ? 1 : 0;
      c.kt = wrap ? 0 : kt1;
      c.kb = wrap ? 0 : c.kb + more * BK;
      c.ti += wrap;
      enter_tile(c, c.ti);
    }
  };
  auto stage_a = [&](int h, const cursor_t& c, uint8_t* buf, auto p_c) __attribute__((always_inline)) {
    constexpr int p = decltype(p_c)::value;
    dma(c.rsA, buf + (h ? kOffA1 : kOffA0) + wave * 4096, c.a_v, c.oa + c.kb + (h * 64 + p * 8) * c.lda, p_c);
  };
  auto stage_b = [&](int h, const cursor_t& c, uint8_t* buf, auto p_c) __attribute__((always_inline)) {
    constexpr int p = decltype(p_c)::value;
    dma(c.rsB, buf + (h ? kOffB1 : kOffB0) + wave * 4096, c.b_v, c.ob + c.kb + (h * 64 + p * 8) * c.ldb, p_c);
  };

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
  auto read_part = [&](v8i (&dst)[4], int lo, int hi, int half, int n) __attribute__((always_inline)) {
    const int f = n >> 1;
    const v4i v = *reinterpret_cast<lds_v4i_p>((size_t)(unsigned)(((n & 1) ? hi : lo) + half * kHalfBytes + f * 2048));
    if (n & 1) dst[f].hi = v;
    else dst[f].lo = v;
  };

  v4f acc[2][2][4][4];
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

  // ---- epilogue (see the header).  An epilogue frame belongs to ONE tile: its output descriptor, offset, leading dimension and
  // alpha travel in `epi_t` (the LAST K-tile runs the tile's own frame, the next tile's FIRST / SECOND K-tiles finish it)
  struct epi_t {
    rsrc_t rs;
    int d_off, ldd, d_voff;
    float alpha;
  };
  auto epi_of = [&](int ti, bool real) __attribute__((always_inline)) -> epi_t {
    int p, ra, rb;
    tile_of(ti, p, ra, rb);
    epi_t e;
    uint16_t* Dp;
    int d_bytes;
    if constexpr (GROUPED) {
      const GroupProblem& P = ka.p[p];
      Dp = P.D;
      d_bytes = P.d_bytes;
      e.ldd = P.ldd;
      e.alpha = p == 0 ? alpha_of[0] : p == 1 ? alpha_of[1] : p == 2 ? alpha_of[2] : alpha_of[3];
    } else {
      Dp = ka.D;
      d_bytes = ka.d_bytes;
      e.ldd = ka.ldd;
      e.alpha = alpha_of[0];
    }
    // no previous tile: a descriptor of zero records (the address check drops the stores)
    e.rs = __builtin_amdgcn_make_buffer_rsrc((void*)Dp, 0, real ? d_bytes : 0, 0x00020000);
    e.d_off = (ra * e.ldd + rb) * 2;
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
    const int fr = ln & 15, fq = ln >> 4;
    const int ecol = (fq & 1) * 16 + (fq >> 1) * 8;  // column of this lane's 8-wide piece inside a 32-column block after the swap
    // whole-line stores: lane (m = fr, q) of line half 0 covers row m & 7, of half 1 row 8 + (m & 7); lanes m >= 8 carry block 1
    e.d_voff = ((wr * 128 + (fr & 7)) * e.ldd + wc * 128 + (fr >> 3) * 32 + ecol) * 2;
    return e;
  };
  v4i park[w4::kPark][2];
#pragma unroll
  for (int i = 0; i < w4::kPark; ++i) park[i][0] = park[i][1] = (v4i){0, 0, 0, 0};
  // BIAS: this lane's 4 bias values (packed bf16) for each of the 8 accumulator-tile columns (B half b, fragment j) of the CURRENT
  // tile, loaded in the tile's SECOND K-tile (after the previous tile's conversions, >= 9 phases before the first use, so the
  // counted waits of the phases in between retire them: vmcnt retires in order)
  typedef unsigned int v2u_ __attribute__((ext_vector_type(2)));
  v2u_ bias_w[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_w[b][j] = (v2u_){0u, 0u};
  u32 px = 0, py = 0;
  auto conv_chunk = [&](auto r_c, auto sub_c, const epi_t& e) __attribute__((always_inline)) {
    constexpr int r = decltype(r_c)::value, sub = decltype(sub_c)::value;
    constexpr int q = r >> 2, a = q >> 1, b = (q == 1 || q == 2) ? 1 : 0, i = r & 3;  // Q00, Q01, Q11, Q10
    v4i& e0 = park[r % w4::kPark][0];
    v4i& e1 = park[r % w4::kPark][1];
    if constexpr (sub < 4) {
      // the AGPR -> VGPR copy of an asm output is placed at its DEFINITION (right behind the tile's last MFMA), i.e. a whole
      // quadrant would sit in 64 VGPRs until its chunks come up; passing the tile through an empty asm here pins the copy here
      asm volatile("" : "+a"(acc[a][b][i][sub]));
      // four SCALAR multiplies: as a <4 x float> product the back end emits two v_pk_mul_f32, 12 cycles each on this wave's issue port
      // against 4.3 for a v_mul_f32 (tools/probe_mfma_shadow.hip); the empty asm keeps the legaliser from re-pairing them
      const v4f t = acc[a][b][i][sub];
      float m0 = t[0] * e.alpha, m1 = t[1] * e.alpha, m2 = t[2] * e.alpha, m3 = t[3] * e.alpha;
      asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
      v4f v = {m0, m1, m2, m3};
      if constexpr (BIAS) {
        v[0] += __uint_as_float(bias_w[b][sub].x << 16);
        v[1] += __uint_as_float(bias_w[b][sub].x & 0xFFFF0000u);
        v[2] += __uint_as_float(bias_w[b][sub].y << 16);
        v[3] += __uint_as_float(bias_w[b][sub].y & 0xFFFF0000u);
      }
      const u32 x = pack_bf16x2(v[0], v[1]), y = pack_bf16x2(v[2], v[3]);
      if constexpr ((sub & 1) == 0) {
        px = x;
        py = y;
      } else {
        const auto sx = __builtin_amdgcn_permlane16_swap(px, x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(py, y, false, false);
        const v4i rr = {(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
        if constexpr (sub == 1) e0 = rr;
        else e1 = rr;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int keep = e1[k];
        e1[k] = __builtin_amdgcn_update_dpp(e1[k], e0[k], 0x128, 0xF, 0x3, false);  // rows m < 8  <- block 0 of row m + 8
        e0[k] = __builtin_amdgcn_update_dpp(e0[k], keep, 0x128, 0xF, 0xC, false);   // rows m >= 8 <- block 1 of row m - 8
      }
    }
  };
  auto store_event = [&](auto e_c, auto w_c, const epi_t& e) __attribute__((always_inline)) {
    constexpr int ev = decltype(e_c)::value, r = ev >> 1, half = ev & 1, w = decltype(w_c)::value;
    constexpr int q = r >> 2, a = q >> 1, b = (q == 1 || q == 2) ? 1 : 0, i = r & 3;
    const int rowoff = e.d_off + ((a * 64 + i * 16 + half * 8) * e.ldd + b * 64) * 2;
    const v4i data = park[r % w4::kPark][half];
    if (ABL == 1) {
      asm volatile("" ::"v"(data));
    } else {
      // sc1 = write-through (the tile's 128 KiB would otherwise evict the operand panels from this XCD's L2); `s_nop 1` keeps the
      // data registers live across the store's read (hipcc schedules the next VALU write too early: mi_gemm.hip)
      constexpr int kAux = ABL == 4 ? 0 : ABL == 5 ? 2 : ABL == 6 ? 18 : 16;
      // the store alone sits behind a wave-uniform branch (never an MFMA or a conversion micro-op: branches around those make
      // hipcc spill accumulators / fragments)
      if (wave == w) {
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)data, e.rs, e.d_voff, rowoff, kAux);
        asm volatile("s_nop 1" ::"v"(data) : "memory");
      }
    }
  };
  // everything the epilogue does in frame slot g (after that slot's MFMA)
  auto epi_slot = [&](auto g_c, const epi_t& e) __attribute__((always_inline)) {
    constexpr int g = decltype(g_c)::value;
    if constexpr (ABL != 3) {
      static_for<16>([&](auto r_c) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value, cs = w4::conv_start(r);
        if constexpr (g >= cs && g < cs + 5) conv_chunk(r_c, std::integral_constant<int, g - cs>{}, e);
      });
      if constexpr (g >= w4::store_slot(0) && g < w4::kFrameSlots)
        store_event(std::integral_constant<int, (g - w4::store_slot(0)) / 4>{}, std::integral_constant<int, (g - w4::store_slot(0)) % 4>{}, e);
    }
  };
  rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)lds, 0, 0, 0x00020000);
  if constexpr (BIAS) rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)ka.bias, 0, ka.n_cols * 2, 0x00020000);
  auto load_bias = [&](int rb) __attribute__((always_inline)) {  // 8 loads of 8 bytes per lane; reads past N return 0 (range check)
    if constexpr (BIAS) {
      const int voff = (wc * 128 + (lane >> 4) * 4) * 2;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bias_w[b][j] = __builtin_bit_cast(v2u_, __builtin_amdgcn_raw_buffer_load_b64(rsBias, voff, rb * 2 + (b * 64 + j * 16) * 2, 0));
    }
  };

  cursor_t c2;
  // e: the frame this K-tile works on (LAST: this tile's; FIRST, SECOND: the previous tile's); rb_cur: first column of the current tile
  auto phase = [&](auto mode_c, auto p_c, const v8i (&af)[4], const v8i (&bf)[4], v4f (&c)[4][4], auto stage, auto read,
                   const epi_t& e, int rb_cur) __attribute__((always_inline)) {
    constexpr int MODE_ = decltype(mode_c)::value, P = decltype(p_c)::value;
    constexpr bool ZC = MODE_ == w4::FIRST;
    constexpr bool NOST = ABL == 1 || ABL == 3;
    // younger ops a wait must allow besides the 20 LDS-DMA of the five later stages: the stores, and (BIAS) the 8 bias loads
    // issued in p0 of the SECOND K-tile (frame phase 8), for the five phases that follow it
    constexpr int G_ = MODE_ + P;
    constexpr int XB = (BIAS && MODE_ != w4::MID && G_ >= 9 && G_ <= 13) ? 8 : 0;
    constexpr int X = ((NOST || MODE_ == w4::MID) ? 0 : w4::wait_x(G_)) + XB;
    static_assert(20 + X <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>([&](auto k_c) __attribute__((always_inline)) {
      constexpr int k = decltype(k_c)::value, i = k >> 2, j = k & 3;
      if constexpr (ZC) mfma_acc_zero<FA, FB>(af[i], bf[j], c[i][j], unit);
      else mfma_acc<FA, FB>(af[i], bf[j], c[i][j], unit);
      if constexpr (k == w4::sched_kb(S)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(8)" ::"n"(20 + X) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (BIAS && MODE_ == w4::SECOND && P == 0) load_bias(rb_cur);
      }
      static_for<4>([&](auto p_c) __attribute__((always_inline)) {
        if constexpr (k == w4::sched_dm(S, decltype(p_c)::value)) stage(p_c);
      });
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (k == w4::sched_rd(S, n)) read(n);
      if constexpr (MODE_ != w4::MID && MODE_ != w4::THIRD) epi_slot(std::integral_constant<int, 16 * (MODE_ + P) + k>{}, e);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using p0_t = std::integral_constant<int, 0>;
  using p1_t = std::integral_constant<int, 1>;
  using p2_t = std::integral_constant<int, 2>;
  using p3_t = std::integral_constant<int, 3>;
  // simple_c: this K-tile's stage cursor stays inside its tile (every MID K-tile but the last pair's): three scalar adds instead of
  // the ~20 selects + the tile-table lookup of the general advance -- issue slots of the one wave that also issues the MFMAs
  auto ktile = [&](auto mode_c, auto par_c, const epi_t& e, int rb_cur, auto simple_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    constexpr bool SIMPLE = decltype(simple_c)::value;  // (grouped form too: its general advance is a branch per K-tile)
    uint8_t* const cur = par ? buf1 : buf0;
    if (ABL == 7) {
      if (bid == 0 && wave == 0 && stamp_idx < 1024) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) reinterpret_cast<unsigned long long*>(lds + kLdsBytes)[stamp_idx] = t;
      }
      ++stamp_idx;
    }
    phase(mode_c, p0_t{}, aA, b0[par], acc[0][0], [&](auto p) __attribute__((always_inline)) { stage_a(0, c2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(b1, fb_lo[par], fb_hi[par], 1, n); }, e, rb_cur);
    phase(mode_c, p1_t{}, aA, b1, acc[0][1], [&](auto p) __attribute__((always_inline)) { stage_b(0, c2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(aB, fa_lo[par], fa_hi[par], 1, n); }, e, rb_cur);
    phase(mode_c, p2_t{}, aB, b1, acc[1][1], [&](auto p) __attribute__((always_inline)) { stage_b(1, c2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(aA, fa_lo[par ^ 1], fa_hi[par ^ 1], 0, n); }, e, rb_cur);
    phase(mode_c, p3_t{}, aB, b0[par], acc[1][0], [&](auto p) __attribute__((always_inline)) { stage_a(1, c2, cur, p); },
          [&](int n) __attribute__((always_inline)) { read_part(b0[par ^ 1], fb_lo[par ^ 1], fb_hi[par ^ 1], 0, n); }, e, rb_cur);
    if constexpr (SIMPLE) {
      c2.step += 1;
      c2.kt += 1;
      c2.kb += BK;
    } else {
      advance(c2);
    }
  };

  if (my_tiles <= 0) return;
  // prologue: steps 0 and 1 in the steady-state stage order, with the two read-only phases p2(-1), p3(-1) woven in
  c2.ti = 0; c2.kt = 0; c2.kb = 0; c2.step = 0;
  if constexpr (!GROUPED) {
    c2.nk = ka.K / BK;
    c2.lda = ka.lda;
    c2.ldb = ka.ldb;
    c2.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)ka.A, 0, (int)((unsigned)ka.a_bytes + 4096u), 0x00020000);
    c2.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)ka.B, 0, (int)((unsigned)ka.b_bytes + 4096u), 0x00020000);
    const int lr = lane >> 3, chunk = ((lane & 7) ^ swz_f(lr)) * 16;
    const int row = (wave >> 1) * 128 + (wave & 1) * 32 + lr;
    c2.a_v = row * ka.lda + chunk;
    c2.b_v = row * ka.ldb + chunk;
    total = my_tiles * c2.nk;
  }
  enter_tile(c2, 0);
  auto stage4a = [&](int h, uint8_t* buf) __attribute__((always_inline)) {
    static_for<4>([&](auto p_c) __attribute__((always_inline)) { stage_a(h, c2, buf, p_c); });
  };
  auto stage4b = [&](int h, uint8_t* buf) __attribute__((always_inline)) {
    static_for<4>([&](auto p_c) __attribute__((always_inline)) { stage_b(h, c2, buf, p_c); });
  };
  stage4a(0, buf0);
  stage4b(0, buf0);
  stage4b(1, buf0);
  stage4a(1, buf0);
  advance(c2);  // step 1
  stage4a(0, buf1);
  stage4b(0, buf1);
  asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stage4b(1, buf1);
#pragma unroll
  for (int n = 0; n < 8; ++n) read_part(aA, fa_lo[0], fa_hi[0], 0, n);
  asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stage4a(1, buf1);
#pragma unroll
  for (int n = 0; n < 8; ++n) read_part(b0[0], fb_lo[0], fb_hi[0], 0, n);
  advance(c2);  // step 2
  __builtin_amdgcn_sched_barrier(0);

  unsigned long long ck0 = 0, rt0 = 0;
  if (ABL == 2) {
    ck0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  using par0_t = std::integral_constant<int, 0>;
  using par1_t = std::integral_constant<int, 1>;
  using mid_t = std::integral_constant<int, w4::MID>;
  using first_t = std::integral_constant<int, w4::FIRST>;
  using second_t = std::integral_constant<int, w4::SECOND>;
  using third_t = std::integral_constant<int, w4::THIRD>;
  using last_t = std::integral_constant<int, w4::LAST>;
  epi_t ep_prev = epi_of(0, false);
  int steps_done = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    const epi_t ep_cur = epi_of(ti, true);
    int p_, ra_, rb_cur;
    tile_of(ti, p_, ra_, rb_cur);
    int nk_t;
    if constexpr (GROUPED) nk_t = ka.p[p_].nk;
    else nk_t = ka.K / BK;
    using full_t = std::false_type;
    using simple_t = std::true_type;
    ktile(first_t{}, par0_t{}, ep_prev, rb_cur, full_t{});
    ktile(second_t{}, par1_t{}, ep_prev, rb_cur, full_t{});
    ktile(third_t{}, par0_t{}, ep_cur, rb_cur, full_t{});
    // MID K-tiles j = 3 .. nk - 2; the stage cursor (two K-tiles ahead, advanced at the end of a K-tile) leaves the tile in j = nk - 3
    const int npair = nk_t / 2;
    for (int pair = 2; pair < npair - 1; ++pair) {
      ktile(mid_t{}, par1_t{}, ep_cur, rb_cur, simple_t{});
      ktile(mid_t{}, par0_t{}, ep_cur, rb_cur, simple_t{});
    }
    if (npair > 2) {
      ktile(mid_t{}, par1_t{}, ep_cur, rb_cur, full_t{});
      ktile(mid_t{}, par0_t{}, ep_cur, rb_cur, full_t{});
    }
    ktile(last_t{}, par1_t{}, ep_cur, rb_cur, full_t{});
    ep_prev = ep_cur;
    steps_done += nk_t;
  }
  if (ABL == 2) {
    const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      if constexpr (!GROUPED) {
        unsigned long long* o = ka.dbg + (size_t)bid * 4;
        o[0] = ck1 - ck0;
        o[1] = rt1 - rt0;
        o[2] = (unsigned long long)steps_done;
        o[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
      }
    }
  }
  if (ABL == 7) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if constexpr (!GROUPED) {
      if (bid == 0 && wave == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 1024; i += 64)
          ka.dbg[i] = i < stamp_idx ? reinterpret_cast<unsigned long long*>(lds + kLdsBytes)[i] : (i == stamp_idx ? t_end : 0ull);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // tail prefetches retired; last MFMAs written back
  // the rest of the last tile's frame
  static_for<w4::kFrameSlots - 64>([&](auto g_c) __attribute__((always_inline)) {
    epi_slot(std::integral_constant<int, 64 + decltype(g_c)::value>{}, ep_prev);
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int FA, int FB>
static int launch_w4_fmt(const uint8_t* a, const uint8_t* b, uint16_t* D, const float* sa_inv, const float* sb_inv, int64_t M,
                         int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int variant, void* dbg, hipStream_t st,
                         const void* bias) {
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
  const dim3 grid(tiles_m * tiles_n), block(256);
#define MI_W4(ABLv)                                                                                                      \
  hipLaunchKernelGGL((gemm_w4<FA, FB, ABLv>), grid, block, 0, st, a, b, D, sa_inv, sb_inv, (int)K, (int)lda, (int)ldb,    \
                     (int)ldd, tiles_m, tiles_n, (int)(M * lda), (int)(N * ldb), (int)(M * ldd * 2), (unsigned long long*)dbg)
  W4Single ws;
  ws.A = a; ws.B = b; ws.D = D; ws.sa_inv = sa_inv; ws.sb_inv = sb_inv; ws.bias = (const uint16_t*)bias; ws.dbg = (unsigned long long*)dbg;
  ws.K = (int)K; ws.lda = (int)lda; ws.ldb = (int)ldb; ws.ldd = (int)ldd; ws.tiles_m = tiles_m; ws.tiles_n = tiles_n;
  ws.a_bytes = (int)(M * lda); ws.b_bytes = (int)(N * ldb); ws.d_bytes = (int)(M * ldd * 2); ws.n_cols = (int)N;
#define MI_W4P(ABLv) hipLaunchKernelGGL((gemm_w4p<FA, FB, ABLv, 0>), pgrid, block, 0, st, ws)
  const int ntiles = tiles_m * tiles_n;
  const dim3 pgrid(ntiles < num_cus() ? ntiles : num_cus());
  if (variant >= 10 && variant < 20 && ((ntiles + (int)pgrid.x - 1) / (int)pgrid.x > 64 || K < 512 || tiles_m >= 16384 || tiles_n >= 16384)) {
    set_error("mi_gemm (w4 persistent): needs K >= 512, at most 64 tiles per workgroup and fewer than 16384 tiles per dimension");
    return MI_ERR_SHAPE;
  }
  if (bias != nullptr && variant != 10) {
    set_error("mi_gemm (w4): only the persistent product kernel takes a bias");
    return MI_ERR_ARG;
  }
  if (variant == 0) MI_W4(0);
  else if (variant == 10 && bias != nullptr) hipLaunchKernelGGL((gemm_w4p<FA, FB, 0, 1>), pgrid, block, 0, st, ws);
  else if (variant == 10) MI_W4P(0);
  else if (variant == 12) {
    if constexpr (FA == 0 && FB == 0) MI_W4P(2);
    else {
      set_error("mi_gemm (w4): the clock-stamp build is E4M3 x E4M3 only");
      return MI_ERR_ARG;
    }
  }
#ifdef MI_DIAG
  else if (variant == 1) MI_W4(1);
  else if (variant == 2) MI_W4(2);
#define MI_W4S(ABLv, Sv)                                                                                                 \
  hipLaunchKernelGGL((gemm_w4<FA, FB, ABLv, Sv>), grid, block, 0, st, a, b, D, sa_inv, sb_inv, (int)K, (int)lda, (int)ldb, \
                     (int)ldd, tiles_m, tiles_n, (int)(M * lda), (int)(N * ldb), (int)(M * ldd * 2), (unsigned long long*)dbg)
  else if (variant >= 20 && variant < 40 && FA == 0 && FB == 0) {  // schedule sweep (E4M3 x E4M3): 20 + 4 S + {0: product, 1: no stores, 2: stamps}
    const int S = (variant - 20) / 4, kind = (variant - 20) % 4;
    if constexpr (FA == 0 && FB == 0) {
      switch (S * 4 + kind) {
        case 4: MI_W4S(0, 1); break;
        case 5: MI_W4S(1, 1); break;
        case 6: MI_W4S(2, 1); break;
        case 12: MI_W4S(0, 3); break;
        case 13: MI_W4S(1, 3); break;
        case 14: MI_W4S(2, 3); break;
        case 16: MI_W4S(0, 4); break;
        case 17: MI_W4S(1, 4); break;
        case 18: MI_W4S(2, 4); break;
        default: set_error("mi_gemm (w4): no such schedule variant %d", variant); return MI_ERR_ARG;
      }
    }
  }
#undef MI_W4S
  else if ((variant == 11 || (variant >= 13 && variant <= 17)) && !(FA == 0 && FB == 0)) {
    set_error("mi_gemm (w4): timing variant %d is built for E4M3 x E4M3 only", variant);
    return MI_ERR_ARG;
  }
#define MI_W4P_LAB(ABLv) do { if constexpr (FA == 0 && FB == 0) MI_W4P(ABLv); } while (0)
  else if (variant == 11) MI_W4P_LAB(1);
  else if (variant == 13) MI_W4P_LAB(3);
  else if (variant == 14) MI_W4P_LAB(4);
  else if (variant == 15) MI_W4P_LAB(5);
  else if (variant == 16) MI_W4P_LAB(6);
  else if (variant == 17) MI_W4P_LAB(7);
#undef MI_W4P_LAB
#endif
  else {
    set_error("mi_gemm (w4): variant %d is a timing build of the lab library", variant);
    return MI_ERR_ARG;
  }
#undef MI_W4P
#undef MI_W4
  MI_CHECK_LAUNCH("mi_gemm (w4) launch");
  return MI_OK;
}

// variant: one tile per workgroup: 0 = product, 1 = no stores, 2 = clock stamps (dbg = u64[4 * tiles]); persistent (gemm_w4p):
// 10 = product, 11 = no stores, 12 = clock stamps (dbg = u64[4 * grid]), 13 = no epilogue.  Shapes: M, N % 256 == 0, K % 256 == 0,
// operands below 2 GiB (the dispatcher in mi_gemm.hip checks).
int launch_w4(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldd, int fa, int fb, int variant, void* dbg, hipStream_t st, const void* bias) {
  const uint8_t *a = (const uint8_t*)A, *b = (const uint8_t*)B;
  uint16_t* d = (uint16_t*)D;
  if (fa == 0 && fb == 0) return launch_w4_fmt<0, 0>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st, bias);
  if (fa == 0 && fb == 1) return launch_w4_fmt<0, 1>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st, bias);
  if (fa == 1 && fb == 0) return launch_w4_fmt<1, 0>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st, bias);
  return launch_w4_fmt<1, 1>(a, b, d, sa_inv, sb_inv, M, N, K, lda, ldb, ldd, variant, dbg, st, bias);
}

// grouped launch (mi_gemm_fp8_grouped, tile_cfg 4): the schedule in `ga` is the eight-wave kernel's (256 x 256 tiles)
int launch_w4_grouped(const GroupArgs& ga, int fa, int fb, int grid, hipStream_t st) {
  for (int q = 0; q < ga.n; ++q)
    if (ga.p[q].nk < 4 || (ga.p[q].nk & 1)) {
      set_error("mi_gemm_fp8_grouped (four-wave kernel): every problem needs K >= 512 and K %% 256 == 0");
      return MI_ERR_SHAPE;
    }
  if (fa == 0 && fb == 0) hipLaunchKernelGGL((gemm_w4p<0, 0, 0, 2>), dim3(grid), dim3(256), 0, st, ga);
  else if (fa == 0 && fb == 1) hipLaunchKernelGGL((gemm_w4p<0, 1, 0, 2>), dim3(grid), dim3(256), 0, st, ga);
  else if (fa == 1 && fb == 0) hipLaunchKernelGGL((gemm_w4p<1, 0, 0, 2>), dim3(grid), dim3(256), 0, st, ga);
  else hipLaunchKernelGGL((gemm_w4p<1, 1, 0, 2>), dim3(grid), dim3(256), 0, st, ga);
  MI_CHECK_LAUNCH("mi_gemm_fp8_grouped (four-wave) launch");
  return MI_OK;
}

}  // namespace mi
