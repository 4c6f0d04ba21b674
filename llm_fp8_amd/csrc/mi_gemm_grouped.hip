// Grouped FP8 GEMM: up to 4 independent "TN" problems D_p = (A_p . B_p^T) * alpha_p in ONE persistent launch.
//
// Why: a Linear's backward is two GEMMs on the same grad_output -- dgrad dX[M,K] = G8[M,N] . W8T[K,N]^T and wgrad
// dW[N,K] = G8T[N,M] . X8T[K,M]^T (te_llama.py:77,80 call sites).  Launched one after the other each pays its
// own ramp (first operand panels from beyond L2), its own exposed last epilogue (128 KiB of output per CU that nothing
// overlaps) and, worst, its own wave quantisation: 3072x8192 is 384 tiles = 1.5 rounds on 256 CUs, 3072x3072 is 0.56 round.
// Here the tiles of all problems form one list, longest tiles first, dealt round-robin to the persistent workgroups, so the
// short problem's tiles fill the holes of the long one's last round and the pipeline (LDS-DMA prefetch, distributed epilogue)
// runs straight across the problem boundary.
//
// Kernel = the persistent 8-phase kernel of mi_gemm.hip (same LDS image, same two-wave-group phase schedule, same distributed
// whole-line epilogue) with every per-problem quantity -- buffer descriptors, leading dimensions, K-tile count, alpha, output
// -- carried per prefetch cursor and switched at tile boundaries.  Per-tensor scaling only (no MX, no bias): what dgrad / wgrad
// need.  All problems share the operand formats (FA, FB) and the tile shape.
#include "mi_gemm_dev.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace mi {

// prefetch cursor: the (tile, K-tile) of a future step with everything needed to stage from it (wave-uniform)
struct Cursor {
  int ti, kt, nk, oa, ob, lda, ldb;
  rsrc_t rsA, rsB;
  int va0, va1, vb0, vb1;  // per-lane byte offsets (row * ld + chunk) of the wave's first piece of each half-tile: recomputed
                           // only when the cursor enters a tile (the leading dimensions belong to the tile's problem)
};
// what the epilogue of a finished tile needs of its problem
struct Epi {
  rsrc_t rsD;
  int ldd, d_off;
  float alpha;
};

template <int FA, int FB, int MA1, int NB1>
__global__ __launch_bounds__(512, 2) void gemm_256_grp(GroupArgs ga) {
  constexpr int RA0 = 64, RA1 = 16 * MA1, RB0 = 32, RB1 = 16 * NB1;
  constexpr int TBM = 2 * (RA0 + RA1), TBN = 4 * (RB0 + RB1);
  constexpr int nA1 = MA1 / 2, nB1 = NB1;
  constexpr int W = nA1 + nB1 + 4;
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int G = gridDim.x, bid = blockIdx.x;
  // virtual index: XCD-major order of the workgroups (round-robin dispatch over the 8 XCDs: bid & 7 names the L2)
  const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
  const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int my_cnt[kMaxGroup], my_tiles = 0;
#pragma unroll
  for (int q = 0; q < kMaxGroup; ++q) {
    my_cnt[q] = q < ga.n ? (int)ga.cnt[q][v] : 0;
    my_tiles += my_cnt[q];
  }

  // alpha of every problem, read ONCE (inside the tile walk a memory read of it put `s_waitcnt vmcnt(0)` -- a drain of the whole
  // LDS-DMA pipeline -- at every tile boundary), kept as wave-uniform values
  float alpha_of[kMaxGroup];
#pragma unroll
  for (int q = 0; q < kMaxGroup; ++q) {
    float a = 1.0f;
    if (q < ga.n) a = (*ga.p[q].sa_inv) * (*ga.p[q].sb_inv);
    alpha_of[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
  }

  // ---- tile table: lane i holds (problem, tile row, tile column) of this workgroup's i-th tile, packed into one register:
  // tm in bits 0-13, tn in bits 14-27, problem in bits 28-29 (host: my_tiles <= 64, tiles_m, tiles_n < 16384).
  // Pass 1 (uniform loop over my tiles): the tile's id within its problem = R[p][j] + rank of v among the workgroups that have a
  // j-th tile of p (a 256-entry count by ballots); pass 2 (one vector evaluation): id -> (tm, tn) in the grouped tile order.
  int tab;
  {
    int my_id = 0, my_p = 0;
    int ti = 0;
#pragma unroll
    for (int q = 0; q < kMaxGroup; ++q) {
      if (q < ga.n) {
        // lane l looks at the counts of virtual workgroups 4 l .. 4 l + 3
        const unsigned c4 = reinterpret_cast<const unsigned*>(ga.cnt[q])[lane];
        for (int j = 0; j < my_cnt[q]; ++j, ++ti) {
          int rank = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool f = (int)((c4 >> (8 * k)) & 255u) > j && (4 * lane + k) < v;
            rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
          }
          if (lane == ti) {
            my_id = (int)ga.R[q][j] + rank;
            my_p = q;
          }
        }
      }
    }
    int tmn = ga.p[0].tiles_m, tnn = ga.p[0].tiles_n;
#pragma unroll
    for (int q = 1; q < kMaxGroup; ++q) {
      const bool is = my_p == q;
      tmn = is ? ga.p[q].tiles_m : tmn;
      tnn = is ? ga.p[q].tiles_n : tnn;
    }
    int tm, tn;
    tile_of_flat(my_id, max(tmn, 1), max(tnn, 1), tm, tn);
    tab = tm | (tn << 14) | (my_p << 28);
  }
  auto tile_of = [&](int ti, int& p, int& ra, int& rb) {
    const int t = __builtin_amdgcn_readlane(tab, ti);
    p = (t >> 28) & 3;
    ra = (t & 0x3FFF) * TBM;
    rb = ((t >> 14) & 0x3FFF) * TBN;
  };

  // ---- staging addresses.  A wave feeds 8-row x 128-byte pieces of each half-tile; lane (lr = lane >> 3, lc = lane & 7) reads
  // the swizzled 16-byte chunk lc ^ swz(lr) of a tile row; the wave's second piece of a half (where there is one) is the 8 rows
  // below (wave w feeds local rows [16 w, 16 w + 16) of a 128-row half, which never straddle a 32- or 64-row block), so the
  // scalar offset carries the 8 * ld.  The byte offsets row * ld + chunk are per cursor (the leading dimensions change with the problem) and are recomputed when a
  // cursor enters a tile -- never inside a K-tile: a dozen VALU per load segment cost the kernel 12 %.
  auto stage_rows = [&](int& a0, int& a1, int& b0, int& b1, int& chunk) {  // evaluated when a cursor enters a tile, from the lane id
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));  // (not hoisted: nothing of this stays live across the K loop)
    const int lr = ln >> 3;
    chunk = ((ln & 7) ^ swz_f(lr)) * 16;
    const int l0 = wave * 16 + lr;
    const int la = wave * nA1 * 8 + lr, lb = wave * nB1 * 8 + lr;
    a0 = (l0 / RA0) * (RA0 + RA1) + l0 % RA0;
    b0 = (l0 / RB0) * (RB0 + RB1) + l0 % RB0;
    a1 = (la / RA1) * (RA0 + RA1) + RA0 + la % RA1;
    b1 = (lb / RB1) * (RB0 + RB1) + RB0 + lb % RB1;
  };

  // ---- prefetch cursors: the (tile, K-tile) of step s + 1 and s + 2 with everything needed to stage from them
  auto load_tile = [&](Cursor& c, int ti) {
    int p, ra, rb;
    tile_of(ti, p, ra, rb);
    const GroupProblem& P = ga.p[p];
    c.ti = ti;
    c.kt = 0;
    c.nk = P.nk;
    c.lda = P.lda;
    c.ldb = P.ldb;
    c.oa = ra * P.lda;
    c.ob = rb * P.ldb;
    c.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, P.a_bytes, 0x00020000);
    c.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, P.b_bytes, 0x00020000);
    int a0, a1, b0, b1, ch;
    stage_rows(a0, a1, b0, b1, ch);
    c.va0 = a0 * P.lda + ch;
    c.va1 = a1 * P.lda + ch;
    c.vb0 = b0 * P.ldb + ch;
    c.vb1 = b1 * P.ldb + ch;
  };
  auto advance = [&](Cursor& c) {  // next step; past the last one the cursor stays on the final K-tile (dead re-fetch)
    if (c.kt + 1 < c.nk) {
      ++c.kt;
    } else if (c.ti + 1 < my_tiles) {
      load_tile(c, c.ti + 1);
    }
  };
  enum { kA0 = 0, kA1 = 1, kB0 = 2, kB1 = 3 };
  auto stage_a = [&](const Cursor& c, int kind, int n, uint8_t* lds_half) {
    const int soff = c.oa + c.kt * BK, voff = kind == kA0 ? c.va0 : c.va1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < n)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsA, LDS_PTR(lds_half + (wave * n + i) * 1024), 16, voff, soff + i * 8 * c.lda, 0, 0);
  };
  auto stage_b = [&](const Cursor& c, int kind, int n, uint8_t* lds_half) {
    const int soff = c.ob + c.kt * BK, voff = kind == kB0 ? c.vb0 : c.vb1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < n)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsB, LDS_PTR(lds_half + (wave * n + i) * 1024), 16, voff, soff + i * 8 * c.ldb, 0, 0);
  };

  Cursor c1, c2;
  load_tile(c1, 0);
  Cursor c0 = c1;  // step 0
  advance(c1);
  c2 = c1;
  advance(c2);

  v4f acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};

  uint8_t* const buf0 = lds;
  uint8_t* const buf1 = lds + kBufBytes;
  // prologue: step 0 complete, (step 1: A0, B0) in flight
  stage_a(c0, kA0, 2, buf0 + kOffA0);
  stage_b(c0, kB0, 2, buf0 + kOffB0);
  stage_b(c0, kB1, nB1, buf0 + kOffB1);
  stage_a(c0, kA1, nA1, buf0 + kOffA1);
  stage_a(c1, kA0, 2, buf1 + kOffA0);
  stage_b(c1, kB0, 2, buf1 + kOffB0);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();

  // ---- epilogue of the PREVIOUS tile (its problem's output, leading dimension and alpha), distributed over the four phases of
  // the next tile's first K-tile exactly as in mi_gemm.hip (epi_part): whole 128-byte lines where NB1 == 2
  auto epi_of = [&](int ti) -> Epi {
    int p, ra, rb;
    tile_of(ti, p, ra, rb);
    const GroupProblem& P = ga.p[p];
    Epi e;
    e.rsD = __builtin_amdgcn_make_buffer_rsrc((void*)P.D, 0, P.d_bytes, 0x00020000);
    e.ldd = P.ldd;
    e.d_off = (ra * P.ldd + rb) * 2;
    e.alpha = p == 0 ? alpha_of[0] : p == 1 ? alpha_of[1] : p == 2 ? alpha_of[2] : alpha_of[3];
    return e;
  };
  v4i held[2];
  auto epi_part = [&](auto a_c, auto part_c, const Epi& e, bool zero) __attribute__((always_inline)) {
    constexpr int a = decltype(a_c)::value, PART = decltype(part_c)::value;
    constexpr int F = a == 0 ? 4 : MA1, H = F / 2;
    constexpr int bf = a == 0 ? 0 : 1;
    int ln;
    asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));  // lane-derived store offsets: recomputed here, not kept live
    const int fr = ln & 15, fq = ln >> 4;
    const int ecol = (fq & 1) * 16 + (fq >> 1) * 8;
    auto zero_blk = [&](int i, int b) __attribute__((always_inline)) {
      if (zero) {
#pragma unroll
        for (int j = 0; j < (b == 0 ? 2 : NB1); ++j) acc[a][i][b][j] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto pack_blk = [&](int i, int b) __attribute__((always_inline)) -> v4i {
      if (b == 1 && NB1 == 1) {
        const v4f v0 = acc[a][i][1][0] * e.alpha;
        return (v4i){(int)pack_bf16x2(v0[0], v0[1]), (int)pack_bf16x2(v0[2], v0[3]), 0, 0};
      }
      const v4f v0 = acc[a][i][b][0] * e.alpha, v1 = acc[a][i][b][1] * e.alpha;
      u32 p0x = pack_bf16x2(v0[0], v0[1]), p0y = pack_bf16x2(v0[2], v0[3]);
      u32 p1x = pack_bf16x2(v1[0], v1[1]), p1y = pack_bf16x2(v1[2], v1[3]);
      auto sx = __builtin_amdgcn_permlane16_swap(p0x, p1x, false, false);
      auto sy = __builtin_amdgcn_permlane16_swap(p0y, p1y, false, false);
      return (v4i){(int)sx[0], (int)sy[0], (int)sx[1], (int)sy[1]};
    };
    auto store_row = [&](int i, const v4i& o0, const v4i& o1) __attribute__((always_inline)) {
      const int rowoff = e.d_off + ((a * RA0 + i * 16) * e.ldd) * 2;
      if (NB1 == 2) {
        const bool lo = fr < 8;
        v4i x, y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = lo ? o1[k] : o0[k];
          const int r = __builtin_amdgcn_update_dpp(0, t, 0x128, 0xF, 0xF, false);  // row_ror:8 = lane m ^ 8
          x[k] = lo ? o0[k] : r;
          y[k] = lo ? r : o1[k];
        }
        const int dvo = ((wr * (RA0 + RA1) + (fr & 7)) * e.ldd + wc * (RB0 + RB1) + (fr >> 3) * RB0 + ecol) * 2;
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)x, e.rsD, dvo, rowoff, 16);
        asm volatile("s_nop 1" ::"v"(x) : "memory");
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)y, e.rsD, dvo, rowoff + 8 * e.ldd * 2, 16);
        asm volatile("s_nop 1" ::"v"(y) : "memory");
      } else {
        const int dvo = ((wr * (RA0 + RA1) + fr) * e.ldd + wc * (RB0 + RB1)) * 2;
        __builtin_amdgcn_raw_buffer_store_b128((mi::v4u)o0, e.rsD, dvo + ecol * 2, rowoff, 16);
        asm volatile("s_nop 1" ::"v"(o0) : "memory");
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        const v2u o2 = {(unsigned)o1[0], (unsigned)o1[1]};
        __builtin_amdgcn_raw_buffer_store_b64(o2, e.rsD, dvo + fq * 8, rowoff + RB0 * 2, 16);
        asm volatile("s_nop 1" ::"v"(o1) : "memory");
      }
    };
#pragma unroll
    for (int i = 0; i < F; ++i) {
      const bool first_rows = i < H;
      if (PART == 2 || (PART == 0 && first_rows)) {
        const v4i o0 = pack_blk(i, 0), o1 = pack_blk(i, 1);
        store_row(i, o0, o1);
        zero_blk(i, 0);
        zero_blk(i, 1);
      } else if (PART == 0) {
        held[i - H] = pack_blk(i, bf);
        zero_blk(i, bf);
      } else if (!first_rows) {
        const v4i oo = pack_blk(i, 1 - bf);
        if (bf == 0) store_row(i, held[i - H], oo);
        else store_row(i, oo, held[i - H]);
        zero_blk(i, 1 - bf);
      }
    }
  };
  using c0_t = std::integral_constant<int, 0>;
  using c1_t = std::integral_constant<int, 1>;
  using c2_t = std::integral_constant<int, 2>;

  // ---- one K-tile = 4 phases (mi_gemm.hip: ktile).  Waits: see the table there; stores per phase 4 | 4 | MA1 | MA1.
#define MI_WAIT_SYNC(MODE_, flag, PH)                                                                              \
  {                                                                                                                \
    constexpr int kAllow = (PH) == 0 ? W - nA1 : (PH) == 1 ? W : (PH) == 2 ? W - 2 : W - nB1;                      \
    constexpr int kX = (MODE_) == 0 ? 0                                                                            \
                     : (MODE_) == 1 ? ((PH) == 0 ? 4 : (PH) == 1 ? 8 : (PH) == 2 ? 8 + MA1 : 8 + 2 * MA1)          \
                                    : ((PH) <= 1 ? 4 + 2 * MA1 : (PH) == 2 ? MA1 : 0);                             \
    if ((flag) && kX != 0) wait_vmcnt<kAllow + kX>();                                                              \
    else wait_vmcnt<kAllow>();                                                                                     \
  }                                                                                                                \
  __builtin_amdgcn_s_barrier();                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_setprio(1);
#define MI_PIN(NI, NJ, EXPR)                                           \
  _Pragma("unroll") for (int i = 0; i < NI; ++i)                       \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(EXPR));

  v8i af[4], b0f[2], b1f[2];
  auto ktile = [&](auto mode_c, uint8_t* cur, uint8_t* oth, bool flag, const Epi& pe) {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool ZC = MODE == 1;
    // ---- phase 0: C[0][*][0][*]
    if (MODE == 1) {
      stage_b(c1, kB1, nB1, oth + kOffB1);
      if (flag) epi_part(c0_t{}, c0_t{}, pe, false);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) b0f[j] = read_frag(cur + kOffB0, wc * 2 + j, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = read_frag(cur + kOffA0, wr * 4 + i, lane);
    if (MODE != 1) stage_b(c1, kB1, nB1, oth + kOffB1);
    MI_WAIT_SYNC(MODE, flag, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ZC) mfma_ba_zero<FA, FB>(af[i], b0f[j], acc[0][i][0][j], kUnitScale);
        else acc[0][i][0][j] = mfma_ba<FA, FB>(af[i], b0f[j], acc[0][i][0][j], kUnitScale, kUnitScale);
      }
    MI_PIN(4, 2, acc[0][i][0][j])
    MI_PHASE_END();
    // ---- phase 1: C[0][*][1][*]
    if (MODE == 1) {
      stage_a(c1, kA1, nA1, oth + kOffA1);
      if (flag) epi_part(c0_t{}, c1_t{}, pe, false);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) b1f[j] = read_frag(cur + kOffB1, wc * NB1 + j, lane);
    if (MODE != 1) stage_a(c1, kA1, nA1, oth + kOffA1);
    MI_WAIT_SYNC(MODE, flag, 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NB1; ++j) {
        if (ZC) mfma_ba_zero<FA, FB>(af[i], b1f[j], acc[0][i][1][j], kUnitScale);
        else acc[0][i][1][j] = mfma_ba<FA, FB>(af[i], b1f[j], acc[0][i][1][j], kUnitScale, kUnitScale);
      }
    MI_PIN(4, NB1, acc[0][i][1][j])
    MI_PHASE_END();
    // ---- phase 2: C[1][*][1][*]
    if (MODE == 1) {
      stage_a(c2, kA0, 2, cur + kOffA0);
      if (flag) epi_part(c1_t{}, c0_t{}, pe, false);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MA1; ++i) af[i] = read_frag(cur + kOffA1, wr * MA1 + i, lane);
    if (MODE != 1) stage_a(c2, kA0, 2, cur + kOffA0);
    MI_WAIT_SYNC(MODE, flag, 2)
#pragma unroll
    for (int i = 0; i < MA1; ++i)
#pragma unroll
      for (int j = 0; j < NB1; ++j) {
        if (ZC) mfma_ba_zero<FA, FB>(af[i], b1f[j], acc[1][i][1][j], kUnitScale);
        else acc[1][i][1][j] = mfma_ba<FA, FB>(af[i], b1f[j], acc[1][i][1][j], kUnitScale, kUnitScale);
      }
    MI_PIN(MA1, NB1, acc[1][i][1][j])
    MI_PHASE_END();
    // ---- phase 3: C[1][*][0][*]
    stage_b(c2, kB0, 2, cur + kOffB0);
    if (MODE == 1) {
      if (flag) epi_part(c1_t{}, c1_t{}, pe, false);
      __builtin_amdgcn_sched_barrier(0);
    }
    MI_WAIT_SYNC(MODE, flag, 3)
#pragma unroll
    for (int i = 0; i < MA1; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ZC) mfma_ba_zero<FA, FB>(af[i], b0f[j], acc[1][i][0][j], kUnitScale);
        else acc[1][i][0][j] = mfma_ba<FA, FB>(af[i], b0f[j], acc[1][i][0][j], kUnitScale, kUnitScale);
      }
    MI_PIN(MA1, 2, acc[1][i][0][j])
    MI_PHASE_END();
    advance(c1);
    advance(c2);
  };

  Epi prev = epi_of(0);
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool have_prev = ti > 0;
    const int nk = ga.p[(__builtin_amdgcn_readlane(tab, ti) >> 28) & 3].nk;
    ktile(c1_t{}, buf0, buf1, have_prev, prev);
    ktile(c2_t{}, buf1, buf0, have_prev, prev);
    for (int pair = 1; pair < nk / 2; ++pair) {
      ktile(c0_t{}, buf0, buf1, false, prev);
      ktile(c0_t{}, buf1, buf0, false, prev);
    }
    prev = epi_of(ti);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (my_tiles > 0) {
    epi_part(c0_t{}, c2_t{}, prev, false);
    epi_part(c1_t{}, c2_t{}, prev, false);
  }
#undef MI_WAIT_SYNC
#undef MI_PIN
}

// ---- host: longest-processing-time schedule of the group's tiles over the persistent workgroups, cached per shape
struct Sched {
  uint8_t cnt[kMaxGroup][kMaxWg];
  uint16_t R[kMaxGroup][kMaxPerWg];
  int max_per_wg;
  bool ok;
};

static Sched make_sched(int G, int n, const int* ntiles, const int* nk) {
  Sched sc;
  std::memset(&sc, 0, sizeof(sc));
  // problems arrive sorted by decreasing nk.  Greedy: every tile goes to the workgroup with the least work so far (ties: lowest
  // virtual index) -- the classic LPT rule; with tiles of few distinct lengths it is within one short tile of the optimum.
  std::vector<std::pair<long long, int>> heap;  // (load, v), min-heap
  heap.reserve(G);
  for (int v = 0; v < G; ++v) heap.emplace_back(0LL, v);
  auto cmp = [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a > b; };
  std::make_heap(heap.begin(), heap.end(), cmp);
  std::vector<int> per_wg(G, 0);
  sc.ok = true;
  for (int p = 0; p < n; ++p) {
    for (int t = 0; t < ntiles[p]; ++t) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      auto& top = heap.back();
      const int v = top.second;
      if (sc.cnt[p][v] == 255 || per_wg[v] == kMaxPerWg) sc.ok = false;
      else {
        ++sc.cnt[p][v];
        ++per_wg[v];
      }
      top.first += nk[p];
      std::push_heap(heap.begin(), heap.end(), cmp);
    }
    int run = 0;
    for (int j = 0; j < kMaxPerWg; ++j) {
      sc.R[p][j] = (uint16_t)run;
      int have = 0;
      for (int v = 0; v < G; ++v) have += sc.cnt[p][v] > j ? 1 : 0;
      run += have;
    }
    if (run != ntiles[p] || ntiles[p] > 65535) sc.ok = false;
  }
  sc.max_per_wg = 0;
  for (int v = 0; v < G; ++v) sc.max_per_wg = per_wg[v] > sc.max_per_wg ? per_wg[v] : sc.max_per_wg;
  return sc;
}

static const Sched& cached_sched(int G, int n, const int* ntiles, const int* nk) {
  static std::mutex mu;
  static std::map<std::vector<int>, Sched> cache;
  std::vector<int> key{G, n};
  for (int p = 0; p < n; ++p) {
    key.push_back(ntiles[p]);
    key.push_back(nk[p]);
  }
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    if (cache.size() > 256) cache.clear();
    it = cache.emplace(key, make_sched(G, n, ntiles, nk)).first;
  }
  return it->second;
}

template <int FA, int FB>
static int launch_grouped(const GroupArgs& ga, int cfg, int grid, hipStream_t st) {
  switch (cfg) {
    case 0: hipLaunchKernelGGL((gemm_256_grp<FA, FB, 4, 2>), dim3(grid), dim3(512), 0, st, ga); break;
    case 1: hipLaunchKernelGGL((gemm_256_grp<FA, FB, 4, 1>), dim3(grid), dim3(512), 0, st, ga); break;
    case 2: hipLaunchKernelGGL((gemm_256_grp<FA, FB, 2, 2>), dim3(grid), dim3(512), 0, st, ga); break;
    default: hipLaunchKernelGGL((gemm_256_grp<FA, FB, 2, 1>), dim3(grid), dim3(512), 0, st, ga); break;
  }
  MI_CHECK_LAUNCH("mi_gemm_fp8_grouped launch");
  return MI_OK;
}

}  // namespace mi

extern "C" int mi_gemm_fp8_grouped(const mi_gemm_problem* problems, int n, int fmt_a, int fmt_b, int tile_cfg, void* stream) {
  using namespace mi;
  MI_CHECK_ARG(problems && n >= 1 && n <= kMaxGroup, "mi_gemm_fp8_grouped: 1 to %d problems", kMaxGroup);
  MI_CHECK_ARG((fmt_a == 0 || fmt_a == 1) && (fmt_b == 0 || fmt_b == 1), "mi_gemm_fp8_grouped: bad fmt");
  static const int bm[4] = {256, 256, 192, 192}, bn[4] = {256, 192, 256, 192};
  const int ncu = num_cus();
  // tile shape: the caller's, or the one that minimises rounds x tile area / efficiency over the WHOLE group (mi_gemm.hip pick_tile_cfg)
  // tile_cfg 4 = 256 x 256 tiles on the four-wave kernel (mi_gemm_w4.hip; every problem K >= 512): same schedule, other kernel
  const bool four_wave = tile_cfg == 4;
  int cfg = four_wave ? 0 : tile_cfg;
  if (cfg < 0) {
    static const double eff[4] = {1.0, 0.90, 0.90, 0.80};
    double best = 0;
    for (int c = 0; c < 4; ++c) {
      bool ok = true;
      double steps = 0;
      for (int i = 0; i < n; ++i) {
        if (problems[i].M % bm[c] || problems[i].N % bn[c]) ok = false;
        else steps += (double)(problems[i].M / bm[c]) * (problems[i].N / bn[c]) * (problems[i].K / 128);
      }
      if (!ok) continue;
      // persistent walk over a shared tile list: no per-problem round quantisation, only the tail of the whole list
      const double per_wg = steps / ncu;
      const double cost = per_wg * bm[c] * bn[c] / eff[c];
      if (cfg < 0 || cost < best * 0.97) {
        cfg = c;
        best = cost;
      }
    }
  }
  MI_CHECK_ARG(cfg >= 0 && cfg < 4, "mi_gemm_fp8_grouped: no tile shape divides every problem");
  // longest tiles first: a round-robin deal of that order over the persistent workgroups is a longest-processing-time schedule
  int order[kMaxGroup];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (problems[order[j]].K > problems[order[i]].K) {
        const int t = order[i];
        order[i] = order[j];
        order[j] = t;
      }
  GroupArgs ga;
  std::memset(&ga, 0, sizeof(ga));
  int total = 0, ntiles[kMaxGroup], nks[kMaxGroup];
  for (int k = 0; k < n; ++k) {
    const mi_gemm_problem& q = problems[order[k]];
    MI_CHECK_ARG(q.A && q.B && q.D && q.sa_inv && q.sb_inv, "mi_gemm_fp8_grouped: null pointer in problem %d", order[k]);
    MI_CHECK_ARG(q.M > 0 && q.N > 0 && q.K > 0 && q.M % bm[cfg] == 0 && q.N % bn[cfg] == 0 && q.K % 256 == 0,
                 "mi_gemm_fp8_grouped: problem %d (%lld x %lld x %lld) does not fit tile shape %d x %d / K %% 256", order[k],
                 (long long)q.M, (long long)q.N, (long long)q.K, bm[cfg], bn[cfg]);
    MI_CHECK_ARG(q.lda >= q.K && q.ldb >= q.K && q.ldd >= q.N && q.lda % 16 == 0 && q.ldb % 16 == 0 && q.ldd % 4 == 0,
                 "mi_gemm_fp8_grouped: bad leading dimensions in problem %d", order[k]);
    MI_CHECK_ARG(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0 && ((uintptr_t)q.D % 16) == 0,
                 "mi_gemm_fp8_grouped: operands must be 16-byte aligned");
    MI_CHECK_ARG(q.M * q.lda < (1LL << 31) && q.N * q.ldb < (1LL << 31) && q.M * q.ldd * 2 < (1LL << 31),
                 "mi_gemm_fp8_grouped: operands of problem %d exceed 2 GiB (32-bit buffer offsets)", order[k]);
    GroupProblem& P = ga.p[k];
    P.A = (const uint8_t*)q.A;
    P.B = (const uint8_t*)q.B;
    P.D = (uint16_t*)q.D;
    P.sa_inv = q.sa_inv;
    P.sb_inv = q.sb_inv;
    P.lda = (int)q.lda;
    P.ldb = (int)q.ldb;
    P.ldd = (int)q.ldd;
    P.nk = (int)(q.K / 128);
    P.tiles_m = (int)(q.M / bm[cfg]);
    P.tiles_n = (int)(q.N / bn[cfg]);
    MI_CHECK_ARG(P.tiles_m < 16384 && P.tiles_n < 16384, "mi_gemm_fp8_grouped: problem %d has too many tiles per dimension", order[k]);
    P.ntiles = P.tiles_m * P.tiles_n;
    P.tile_base = total;
    P.a_bytes = (int)(q.M * q.lda);
    P.b_bytes = (int)(q.N * q.ldb);
    P.d_bytes = (int)(q.M * q.ldd * 2);
    ntiles[k] = P.ntiles;
    nks[k] = P.nk;
    total += P.ntiles;
  }
  ga.n = n;
  ga.total_tiles = total;
  const int grid = total < ncu ? total : (ncu > kMaxWg ? kMaxWg : ncu);
  const Sched& sc = cached_sched(grid, n, ntiles, nks);
  MI_CHECK_ARG(sc.ok && sc.max_per_wg <= kMaxPerWg, "mi_gemm_fp8_grouped: more than %d tiles per workgroup (%d tiles in all)", kMaxPerWg, total);
  std::memcpy(ga.cnt, sc.cnt, sizeof(ga.cnt));
  std::memcpy(ga.R, sc.R, sizeof(ga.R));
  hipStream_t st = (hipStream_t)stream;
  if (four_wave) return launch_w4_grouped(ga, fmt_a, fmt_b, grid, st);
  if (fmt_a == 0 && fmt_b == 0) return launch_grouped<0, 0>(ga, cfg, grid, st);
  if (fmt_a == 1 && fmt_b == 0) return launch_grouped<1, 0>(ga, cfg, grid, st);
  if (fmt_a == 0 && fmt_b == 1) return launch_grouped<0, 1>(ga, cfg, grid, st);
  return launch_grouped<1, 1>(ga, cfg, grid, st);
}
