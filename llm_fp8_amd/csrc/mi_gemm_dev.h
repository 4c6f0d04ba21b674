// Device helpers shared by the FP8 GEMM kernels (mi_gemm.hip, mi_gemm_grouped.hip): MFMA wrappers, the swizzled LDS image,
// LDS-DMA staging through buffer descriptors, the XCD-aware tile map.  See mi_gemm.hip for the design notes.
#pragma once
#include "mi_common.h"

namespace mi {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int kUnitScale = 0x7F7F7F7F;

template <int FA, int FB>
__device__ __forceinline__ v4f mfma_ba(const v8i& a, const v8i& b, v4f acc, int sa, int sb) {
  // first operand = B fragment (rows -> n), second = A fragment (cols -> m): acc[j] = D[m = lane&15][n = 4*(lane>>4)+j]
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, acc, FB, FA, 0, sb, 0, sa);
}

// same, the A-side scale taken from byte OPS of `sa` (one dword holds the scales of a lane's 4 row-interleaved fragments)
template <int FA, int FB, int OPS>
__device__ __forceinline__ v4f mfma_ba_sel(const v8i& a, const v8i& b, v4f acc, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, acc, FB, FA, 0, sb, OPS, sa);
}

// D = A.B + 0 written as inline asm with the accumulator TIED ("+v"): the hardware ignores the old value (src C is the inline
// constant 0), but the register allocator sees the same read-modify-write as an accumulating MFMA and keeps the tile's
// accumulators in place.  With the builtin and a zero C operand it stops accumulating in place around the tile loop, spills
// the fresh accumulators and reloads them behind `s_waitcnt vmcnt(0)`.  Unit scales only (per-tensor path).
template <int FA, int FB>
__device__ __forceinline__ void mfma_ba_zero(const v8i& a, const v8i& b, v4f& acc, int unit) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0] cbsz:%4 blgp:%5"
               : "+v"(acc)
               : "v"(b), "v"(a), "v"(unit), "n"(FB), "n"(FA));
}

__device__ __forceinline__ int swz_f(int row) { return ((row >> 1) & 3) << 1; }  // depends on row & 7 only

// ------------------------------------------------------------------------------------------------
// Fast path geometry
constexpr int BM = 256, BN = 256, BK = 128;
constexpr int kTileBytes = BM * BK;          // 32 KiB per operand tile
constexpr int kBufBytes = 2 * kTileBytes;    // A + B
constexpr int kLdsBytes = 2 * kBufBytes;     // double buffer = 128 KiB

// XCD-aware, L2-friendly workgroup -> tile map.  Workgroups are dealt round-robin over the 8
// XCDs; give every XCD a contiguous run of the (grouped) tile order so the tiles resident on one
// L2 share A/B panels.  Bijective for any grid size.  Speed only -- never correctness.
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GM = 4;  // super-rows of 4 tile-rows, column-major inside
  int group = id / (GM * tiles_n);
  int first_m = group * GM;
  int gsz = min(tiles_m - first_m, GM);
  int in_g = id - group * GM * tiles_n;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

// grouped order only (stream-K: consecutive flat ids belong to consecutive workgroups of one XCD)
__device__ __forceinline__ void tile_of_flat(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int GM = 4;
  int group = id / (GM * tiles_n);
  int first_m = group * GM;
  int gsz = min(tiles_m - first_m, GM);
  int in_g = id - group * GM * tiles_n;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

// Read the fragment of 16-row group `g` (rows 16g..16g+15 of the tile) for this lane.
__device__ __forceinline__ v8i read_frag(const uint8_t* lds_tile, int g, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const int f = swz_f(r);
  const uint8_t* base = lds_tile + g * 2048 + (r >> 3) * 1024 + (r & 7) * 128;
  v4i lo = *reinterpret_cast<const v4i*>(base + ((q ^ f) << 4));
  v4i hi = *reinterpret_cast<const v4i*>(base + (((4 + q) ^ f) << 4));
  return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int kHalfBytes = 128 * BK;  // 16 KiB
constexpr int kOffA0 = 0, kOffA1 = kHalfBytes, kOffB0 = 2 * kHalfBytes, kOffB1 = 3 * kHalfBytes;

// LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): SGPR resource + per-lane 32-bit
// voffset (constant for the whole kernel) + uniform soffset per stage -> no per-stage VALU address math.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
template <int NP>
__device__ __forceinline__ void stage_n(rsrc_t rs, const int* voff, int soff, uint8_t* lds_half, int wave) {
#pragma unroll
  for (int i = 0; i < NP; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_half + (wave * NP + i) * 1024), 16, voff[i], soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define MI_PHASE_SYNC_BEFORE_MFMA()                  \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);                 \
  __builtin_amdgcn_s_setprio(1);
#define MI_PHASE_END()            \
  __builtin_amdgcn_s_setprio(0);  \
  __builtin_amdgcn_s_barrier();   \
  __builtin_amdgcn_sched_barrier(0);

// ------------------------------------------------------------------------------------------------
// Grouped launches (mi_gemm_grouped.hip: eight-wave kernel; mi_gemm_w4.hip: four-wave kernel, 256 x 256 tiles)
constexpr int kMaxGroup = 4;

struct GroupProblem {
  const uint8_t* A;
  const uint8_t* B;
  uint16_t* D;
  const float* sa_inv;
  const float* sb_inv;
  int lda, ldb, ldd, nk;
  int tiles_m, tiles_n, tile_base, ntiles;
  int a_bytes, b_bytes, d_bytes, pad;
};

constexpr int kMaxWg = 256;      // workgroups of the persistent grid (one per CU)
constexpr int kMaxPerWg = 64;    // tiles per workgroup (one lane of the tile table each)

// The schedule (host, longest-processing-time greedy, cached per shape): cnt[p][v] = tiles of problem p that the workgroup with
// VIRTUAL index v walks (v = XCD-major order of the workgroups: neighbours in v share an L2).  Tile ids of a problem are bound
// round-major: the j-th tile of workgroup v is id R[p][j] + #{v' < v : cnt[p][v'] > j}, so the workgroups of one XCD hold
// consecutive ids -- adjacent tiles of the grouped order, sharing A/B panels -- at the same time.
struct GroupArgs {
  GroupProblem p[kMaxGroup];
  int n, total_tiles;
  uint8_t cnt[kMaxGroup][kMaxWg];
  uint16_t R[kMaxGroup][kMaxPerWg];
};

// the grouped launch of the four-wave kernel (tile shape 256 x 256, every problem K >= 512)
int launch_w4_grouped(const GroupArgs& ga, int fa, int fb, int grid, hipStream_t st);

// mi_gemm_w4.hip: the four-wave (128x128 wave tile) kernel.  variant 0 = product, 1 = no stores, 2 = clock stamps (dbg)
int launch_w4(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldd, int fa, int fb, int variant, void* dbg, hipStream_t st, const void* bias = nullptr);

static inline int num_cus() {
  static int n = 0;  // benign race: every thread computes the same value
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}


}  // namespace mi
