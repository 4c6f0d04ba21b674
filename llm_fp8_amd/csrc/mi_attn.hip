// Attention core of te.pytorch.MultiheadAttention (te_llama.py:45-56,77: causal, GQA, bshd, no dropout) as flash-style
// bf16 kernels for gfx950: S = QK^T never leaves the register file.
//
//   attn_fwd_kernel   one workgroup = 4 waves = 128 query rows of one (batch, head); each wave keeps its 32 query rows as
//                     the B operand of  S^T = K . Q^T  (v_mfma_f32_32x32x16_bf16), so a lane holds 2 x 16 scores of ONE
//                     query row: the online-softmax state (m, l) and the rescale of O^T are lane-local.  The S^T
//                     accumulators, converted pairwise to bf16, are directly the B operand of  O^T += V^T . P^T ; the
//                     V^T fragments come from the row-major V tile with ds_read_b64_tr_b16.  K/V tiles of 64 keys go
//                     global -> LDS by DMA (buffer_load ... lds; the XOR swizzle of the image is applied on the global
//                     side) into a double buffer; one barrier per tile; operand fragments are read one k-step ahead.
//   attn_bwd_dq_kernel / attn_bwd_dkdv_kernel   recompute P from the stored log-sum-exp (no S x S tensor); dQ and
//                     dK/dV are separate passes so that no sum crosses workgroups (bitwise reproducible, no atomics).
//
// Layouts: q/o [B, S, H, D], k/v [B, S, G, D] bf16 with D contiguous and a caller-given token stride; lse [B, H, S] f32 in
// the log2 domain (m * c + log2 l, c = scale * log2 e).
#include "mi_common.h"

namespace mi {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s4v lds_s4v;

// byte offset of 16-byte chunk `ch` of row `row` in a [rows][D] bf16 tile image; the XOR keeps both the ds_read_b128 row
// reads and the ds_read_b64_tr_b16 transposed reads of the 32x32x16 operands spread over the banks
template <int D>
__device__ __forceinline__ int tile_off(int row, int ch) {
  if (D == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  return 128 * row + 16 * (ch ^ (((row & 1) << 2) | ((row >> 1) & 3)));
}

__device__ __forceinline__ bf8 as_bf8(v4i v) { return __builtin_bit_cast(bf8, v); }

// row fragment of the 32x32x16 A/B operand: lane (r, h) <- row r0 + r, elements [16 t + 8 h, +8)
template <int D>
__device__ __forceinline__ bf8 row_frag(const char* tile, int r0, int t, int lane) {
  const int r = lane & 31, h = lane >> 5;
  return as_bf8(*reinterpret_cast<const v4i*>(tile + tile_off<D>(r0 + r, 2 * t + h)));
}

// transposed fragment: lane (c = lane & 31, h) <- column c0 + c, rows {rb, rb+1, rb+2, rb+3, rb+8, .., rb+11}, rb = r0 + 4 h:
// exactly the row set a 32x32 accumulator keeps in registers 8 s .. 8 s + 7 (r0 = 16 s)
template <int D>
__device__ __forceinline__ bf8 tr_frag(const char* tile, int r0, int c0, int lane) {
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3, half = (lane >> 4) & 1, h = lane >> 5;
  const int ch = (c0 >> 3) + 2 * half + (pp >> 1);
  const int rb = r0 + 4 * h + qq;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tile + tile_off<D>(rb, ch) + 8 * (pp & 1)));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(tile + tile_off<D>(rb + 8, ch) + 8 * (pp & 1)));
  s8v v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf8, v);
}

// accumulator registers 8 s .. 8 s + 7 -> bf16 fragment (B operand of a product that sums over the accumulator's rows)
__device__ __forceinline__ bf8 acc_frag(const f16v& a, int s) {
  v4i v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (int)pack_bf16x2(a[8 * s + 2 * j], a[8 * s + 2 * j + 1]);
  return as_bf8(v);
}

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Make a prologue load complete HERE (a use the compiler cannot move): left alone, such loads sink towards their first use
// inside the tile loop, and the s_waitcnt vmcnt placed there also waits -- the counter is in-order -- for the NEXT tile's
// just-issued prefetch loads, every tile.
__device__ __forceinline__ void retire(bf8& f) {
  v4i tmp = __builtin_bit_cast(v4i, f);
  asm volatile("" : "+v"(tmp));
  f = as_bf8(tmp);
}
__device__ __forceinline__ void retire(float& f) { asm volatile("" : "+v"(f)); }

constexpr int ATT_QB = 128;  // query rows per workgroup (4 waves x 32)
constexpr int ATT_KB = 64;   // keys per tile

// LDS-DMA staging (buffer_load_dwordx4 ... lds): a wave-wide instruction deposits 64 x 16 bytes at consecutive LDS addresses,
// so the tile image's XOR swizzle is applied on the GLOBAL side: lane l of this wave's i-th instruction fetches the chunk
// (row, ch) whose image slot is byte (4 i + w) 1024 + 16 l.  No staging registers, no ds_write, no VALU address math per tile
// (per-lane voffset fixed for the kernel, tile advance in the scalar offset).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MI_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int D>
__device__ __forceinline__ void dma_voffsets(int* voff, int64_t tok_stride, int tid) {
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    const int p = (4 * i + w) * 1024 + 16 * lane;
    const int row = p / (2 * D), slot = (p % (2 * D)) / 16;
    const int ch = D == 128 ? slot ^ (((row & 3) << 2) | ((row >> 2) & 3)) : slot ^ (((row & 1) << 2) | ((row >> 1) & 3));
    voff[i] = (int)((int64_t)row * tok_stride * 2) + 16 * ch;
  }
}
template <int D>
__device__ __forceinline__ void stage_dma(rsrc_t rs, const int* voff, int soff, char* tile, int w) {
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MI_LDS_PTR(tile + (4 * i + w) * 1024), 16, voff[i], soff, 0, 0);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D, bool CAUSAL, bool DIAG = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                          const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                          float* __restrict__ lse, int S, int H, int G, int64_t q_ts,
                                                          int64_t k_ts, int64_t v_ts, int64_t o_ts, float c,
                                                          unsigned long long* __restrict__ dbg = nullptr) {
  constexpr int TILE = ATT_KB * D * 2;  // bytes of one K (or V) tile
  constexpr int KT = D / 16, DB = D / 32;
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int qb = gridDim.x - 1 - blockIdx.x, head = blockIdx.y, b = blockIdx.z, g = head / (H / G);
  const int q_first = qb * ATT_QB + w * 32;  // first query row of this wave
  const int qrow = q_first + r;
  const int ntiles = CAUSAL ? (qb * ATT_QB + ATT_QB) / ATT_KB : S / ATT_KB;
  const uint16_t* kbase = k + (int64_t)b * S * k_ts + (int64_t)g * D;
  const uint16_t* vbase = v + (int64_t)b * S * v_ts + (int64_t)g * D;

  bf8 qf[KT];
  {
    const uint16_t* qp = q + ((int64_t)b * S + qrow) * q_ts + (int64_t)head * D + 8 * h;
#pragma unroll
    for (int t = 0; t < KT; ++t) qf[t] = as_bf8(*reinterpret_cast<const v4i*>(qp + 16 * t));
  }
  // retire the Q loads HERE: left to the compiler they sink below the prologue and their vmcnt waits end up in front of the
  // loop's first MFMAs, where the in-order counter makes every tile wait for the NEXT tile's just-issued loads as well
#pragma unroll
  for (int t = 0; t < KT; ++t) retire(qf[t]);
  f16v oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[d][i] = 0.0f;
  float m = -1.0e30f, l = 0.0f;

  // K / V tiles of 64 keys go global -> LDS by DMA, double-buffered: tile j + 1 is requested at the top of tile j
  const rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)((((int64_t)S - 1) * k_ts + D) * 2), 0x00020000);
  const rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)((((int64_t)S - 1) * v_ts + D) * 2), 0x00020000);
  int kvo[D / 32], vvo[D / 32];
  dma_voffsets<D>(kvo, k_ts, tid);
  dma_voffsets<D>(vvo, v_ts, tid);
  stage_dma<D>(rsK, kvo, 0, lds, w);
  stage_dma<D>(rsV, vvo, 0, lds + 2 * TILE, w);
  dma_wait_all();
  __syncthreads();

  unsigned long long tsum[5] = {0, 0, 0, 0, 0}, t_prev = 0;
#define MI_STAMP(k)                                                                       \
  if (DIAG) {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long tn_;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");          \
    if ((k) >= 0) tsum[(k) < 0 ? 0 : (k)] += tn_ - t_prev;                                  \
    t_prev = tn_;                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  }
  MI_STAMP(-1)
  for (int j = 0; j < ntiles; ++j) {
    const char* kt = lds + (j & 1) * TILE;
    const char* vt = lds + (2 + (j & 1)) * TILE;
    const bool more = j + 1 < ntiles;
    if (more) {
      stage_dma<D>(rsK, kvo, (int)((int64_t)(j + 1) * ATT_KB * k_ts * 2), lds + ((j + 1) & 1) * TILE, w);
      stage_dma<D>(rsV, vvo, (int)((int64_t)(j + 1) * ATT_KB * v_ts * 2), lds + (2 + ((j + 1) & 1)) * TILE, w);
    }
    const int key0 = j * ATT_KB;
    if (!CAUSAL || key0 <= q_first + 31) {  // wave-uniform: tiles entirely above the diagonal contribute nothing
      f16v sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.0f;
        // K fragments one k-step ahead of their MFMAs, pinned by scheduling barriers (left alone the scheduler emits
        // ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per MFMA and pays the LDS latency every time)
        bf8 ka = row_frag<D>(kt, 32 * kb, 0, lane);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          bf8 kn = ka;
          if (t + 1 < KT) kn = row_frag<D>(kt, 32 * kb, t + 1, lane);
          __builtin_amdgcn_sched_barrier(0);
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[t], sacc[kb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          ka = kn;
        }
      }
      if (DIAG) {
        asm volatile("" : "+v"(sacc[0][0]), "+v"(sacc[1][15]));
      }
      MI_STAMP(0)
      if (CAUSAL && key0 + ATT_KB - 1 > q_first) {  // tile touches the diagonal
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (key0 + 32 * kb + acc_row(i, h) > qrow) sacc[kb][i] = -INFINITY;
      }
      float mx = sacc[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
      const float mc = m_new * c;
      m = m_new;
      float psum = 0.0f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][i], c, -mc));
          sacc[kb][i] = p;
          psum += p;
        }
      l = l * alpha + psum;
#pragma unroll
      for (int d = 0; d < DB; ++d) oacc[d] = oacc[d] * alpha;  // whole-vector form: lowered to v_pk_mul_f32 (2 floats per issue slot)
      if (DIAG) {
        asm volatile("" : "+v"(oacc[0][0]), "+v"(sacc[1][15]), "+v"(l));
      }
      MI_STAMP(1)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf8 pf = acc_frag(sacc[kb], s);
          bf8 fa = tr_frag<D>(vt, 32 * kb + 16 * s, 0, lane);
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            bf8 fn = fa;
            if (d + 1 < DB) fn = tr_frag<D>(vt, 32 * kb + 16 * s, 32 * (d + 1), lane);
            __builtin_amdgcn_sched_barrier(0);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf, oacc[d], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            fa = fn;
          }
        }
    }
    if (DIAG) {
      asm volatile("" : "+v"(oacc[0][0]), "+v"(oacc[DB - 1][15]));
    }
    MI_STAMP(2)
    dma_wait_all();  // the next tile has landed (this wave's share; the barrier covers the others')
    MI_STAMP(3)
    __syncthreads();
    MI_STAMP(4)
  }
  if (DIAG && dbg != nullptr && lane == 0) {
    unsigned long long* dp = dbg + ((((int64_t)b * H + head) * gridDim.x + qb) * 4 + w) * 8;
    for (int i2 = 0; i2 < 5; ++i2) dp[i2] = tsum[i2];
    dp[5] = (unsigned long long)ntiles;
  }
#undef MI_STAMP

  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  if (h == 0) lse[((int64_t)b * H + head) * S + qrow] = m * c + __builtin_amdgcn_logf(l);  // v_log_f32 = log2
  // O^T (d in registers, query on the lane) -> whole rows through this wave's own LDS slice
  char* ot = lds + w * (32 * D * 2);
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = 4 * d + i;
      uint2 pk;
      pk.x = pack_bf16x2(oacc[d][4 * i] * inv, oacc[d][4 * i + 1] * inv);
      pk.y = pack_bf16x2(oacc[d][4 * i + 2] * inv, oacc[d][4 * i + 3] * inv);
      *reinterpret_cast<uint2*>(ot + r * (D * 2) + 16 * (ch ^ (r & (D / 8 - 1))) + 8 * h) = pk;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slice is private to the wave, no barrier needed
  __builtin_amdgcn_wave_barrier();
  constexpr int CPR = D / 8;  // 16-byte chunks per row
#pragma unroll
  for (int it = 0; it < 32 * CPR / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / CPR, ch = idx % CPR;
    const v4i val = *reinterpret_cast<const v4i*>(ot + row * (D * 2) + 16 * (ch ^ (row & (CPR - 1))));
    *reinterpret_cast<v4i*>(o + ((int64_t)b * S + q_first + row) * o_ts + (int64_t)head * D + ch * 8) = val;
  }
}


// O^T-style accumulators (d in registers, row index on the lane) -> bf16 rows of `out` through a wave-private LDS slice
template <int D>
__device__ __forceinline__ void store_rows_from_accT(const f16v (&acc)[D / 32], float mul, char* slice, uint16_t* out,
                                                     int64_t row_stride, int lane) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int CPR = D / 8;
#pragma unroll
  for (int d = 0; d < D / 32; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = 4 * d + i;
      uint2 pk;
      pk.x = pack_bf16x2(acc[d][4 * i] * mul, acc[d][4 * i + 1] * mul);
      pk.y = pack_bf16x2(acc[d][4 * i + 2] * mul, acc[d][4 * i + 3] * mul);
      *reinterpret_cast<uint2*>(slice + r * (D * 2) + 16 * (ch ^ (r & (CPR - 1))) + 8 * h) = pk;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 32 * CPR / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / CPR, ch = idx % CPR;
    const v4i val = *reinterpret_cast<const v4i*>(slice + row * (D * 2) + 16 * (ch ^ (row & (CPR - 1))));
    *reinterpret_cast<v4i*>(out + (int64_t)row * row_stride + ch * 8) = val;
  }
}

// dQ pass: same walk as the forward (128 query rows per workgroup, 64-key tiles).  S^T and dP^T are recomputed with the
// query on the lane (LSE and delta are lane constants), dS^T feeds  dQ^T += K^T . dS^T  straight from the accumulators.
// Also writes delta[b, h, q] = sum_d dO * O for the dK/dV pass.
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                             const uint16_t* __restrict__ v, const uint16_t* __restrict__ o,
                                                             const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ delta, uint16_t* __restrict__ dq, int S,
                                                             int H, int G, int64_t q_ts, int64_t k_ts, int64_t v_ts,
                                                             int64_t o_ts, int64_t do_ts, int64_t dq_ts, float c, float scale) {
  constexpr int TILE = ATT_KB * D * 2;
  constexpr int KT = D / 16, DB = D / 32;
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int qb = gridDim.x - 1 - blockIdx.x, head = blockIdx.y, b = blockIdx.z, g = head / (H / G);
  const int q_first = qb * ATT_QB + w * 32;
  const int qrow = q_first + r;
  const int ntiles = CAUSAL ? (qb * ATT_QB + ATT_QB) / ATT_KB : S / ATT_KB;
  const uint16_t* kbase = k + (int64_t)b * S * k_ts + (int64_t)g * D;
  const uint16_t* vbase = v + (int64_t)b * S * v_ts + (int64_t)g * D;

  bf8 qf[KT], dof[KT];
  float dl = 0.0f;
  {
    const uint16_t* qp = q + ((int64_t)b * S + qrow) * q_ts + (int64_t)head * D + 8 * h;
    const uint16_t* dp = dout + ((int64_t)b * S + qrow) * do_ts + (int64_t)head * D + 8 * h;
    const uint16_t* op = o + ((int64_t)b * S + qrow) * o_ts + (int64_t)head * D + 8 * h;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      qf[t] = as_bf8(*reinterpret_cast<const v4i*>(qp + 16 * t));
      const v4i dv4 = *reinterpret_cast<const v4i*>(dp + 16 * t);
      const v4i ov4 = *reinterpret_cast<const v4i*>(op + 16 * t);
      dof[t] = as_bf8(dv4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 a = (u32)dv4[j], bb = (u32)ov4[j];
        dl += __uint_as_float(a << 16) * __uint_as_float(bb << 16) + __uint_as_float(a & 0xFFFF0000u) * __uint_as_float(bb & 0xFFFF0000u);
      }
    }
  }
  dl += __shfl_xor(dl, 32);
  const int64_t stat = ((int64_t)b * H + head) * S + qrow;
  if (h == 0) delta[stat] = dl;
  float my_lse = lse[stat];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    retire(qf[t]);
    retire(dof[t]);
  }
  retire(my_lse);
  retire(dl);

  f16v acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[d][i] = 0.0f;

  // K / V tiles of 64 keys go global -> LDS by DMA, double-buffered: tile j + 1 is requested at the top of tile j
  const rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)((((int64_t)S - 1) * k_ts + D) * 2), 0x00020000);
  const rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)((((int64_t)S - 1) * v_ts + D) * 2), 0x00020000);
  int kvo[D / 32], vvo[D / 32];
  dma_voffsets<D>(kvo, k_ts, tid);
  dma_voffsets<D>(vvo, v_ts, tid);
  stage_dma<D>(rsK, kvo, 0, lds, w);
  stage_dma<D>(rsV, vvo, 0, lds + 2 * TILE, w);
  dma_wait_all();
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const char* kt = lds + (j & 1) * TILE;
    const char* vt = lds + (2 + (j & 1)) * TILE;
    const bool more = j + 1 < ntiles;
    if (more) {
      stage_dma<D>(rsK, kvo, (int)((int64_t)(j + 1) * ATT_KB * k_ts * 2), lds + ((j + 1) & 1) * TILE, w);
      stage_dma<D>(rsV, vvo, (int)((int64_t)(j + 1) * ATT_KB * v_ts * 2), lds + (2 + ((j + 1) & 1)) * TILE, w);
    }
    const int key0 = j * ATT_KB;
    auto block = [&](const int kb) {  // one 32-key block of the tile against this wave's 32 query rows
      f16v sacc, pacc;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sacc[i] = 0.0f; pacc[i] = 0.0f; }
      // fragments one k-step ahead of their MFMAs, pinned by scheduling barriers: left alone the scheduler emits
      // ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma for every single MFMA and the LDS latency is paid 48 times per tile
      constexpr int PF = 1;  // k-steps of look-ahead (2 measured the same; 1 leaves 10 VGPRs of slack under the 256 of 2 waves per SIMD)
      bf8 kr[KT], vr[KT];    // (fully unrolled: only PF + 1 of each are live at a time)
#pragma unroll
      for (int t = 0; t < PF; ++t) {
        kr[t] = row_frag<D>(kt, 32 * kb, t, lane);
        vr[t] = row_frag<D>(vt, 32 * kb, t, lane);
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (t + PF < KT) {
          kr[t + PF] = row_frag<D>(kt, 32 * kb, t + PF, lane);
          vr[t + PF] = row_frag<D>(vt, 32 * kb, t + PF, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[t], qf[t], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[t], dof[t], pacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool diag = CAUSAL && key0 + 32 * kb + 31 > q_first;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], c, -my_lse));
        if (diag && key0 + 32 * kb + acc_row(i, h) > qrow) p = 0.0f;
        sacc[i] = p * (pacc[i] - dl);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf8 dsf = acc_frag(sacc, s);
        bf8 fa = tr_frag<D>(kt, 32 * kb + 16 * s, 0, lane);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          bf8 fn = fa;
          if (d + 1 < DB) fn = tr_frag<D>(kt, 32 * kb + 16 * s, 32 * (d + 1), lane);
          __builtin_amdgcn_sched_barrier(0);
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, dsf, acc[d], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          fa = fn;
        }
      }
    };
    // (wave-uniform tests: blocks entirely above the diagonal contribute nothing)
    if (!CAUSAL || key0 <= q_first + 31) block(0);
    if (!CAUSAL || key0 + 32 <= q_first + 31) block(1);
    dma_wait_all();  // the next tile has landed (this wave's share; the barrier covers the others')
    __syncthreads();
  }
  store_rows_from_accT<D>(acc, scale, lds + w * (32 * D * 2), dq + ((int64_t)b * S + q_first) * dq_ts + (int64_t)head * D, dq_ts, lane);
}

// dK / dV pass: one workgroup = 4 waves = 128 keys of one (batch, kv head); each wave keeps K and V rows of its 32 keys as
// B operands (key on the lane) and dK^T, dV^T of those keys in accumulators while the workgroup sweeps the group's query
// heads x 32-row query slices (Q and dO slices staged once for all four waves; row reads for S and dP, transposed reads
// for the dV^T and dK^T products).
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, D == 128 ? 1 : 2) void attn_bwd_dkdv_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                               const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, int S, int H,
                                                               int G, int B, int64_t q_ts, int64_t k_ts, int64_t v_ts,
                                                               int64_t do_ts, int64_t dk_ts, int64_t dv_ts, float c, float scale) {
  constexpr int SL = 32 * D * 2;  // bytes of one 32-row slice image
  constexpr int KT = D / 16, DB = D / 32;
  constexpr int BUF = 2 * SL + 256;  // Q slice, dO slice, lse[32], delta[32]
  __shared__ __attribute__((aligned(16))) char lds[(2 * BUF > 4 * SL) ? 2 * BUF : 4 * SL];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  // heaviest key blocks first: the key-block index is the slowest-varying part of the linear workgroup id
  const int lin = blockIdx.x, nkb = S / 128;
  const int kbi = CAUSAL ? lin / (G * B) : nkb - 1 - lin / (G * B);
  const int g = (lin / B) % G, b = lin % B;
  const int rep = H / G;
  const int key_first = kbi * 128 + w * 32;  // first key of this wave
  const int key = key_first + r;

  bf8 kf[KT], vf[KT];
  {
    const uint16_t* kp = k + ((int64_t)b * S + key) * k_ts + (int64_t)g * D + 8 * h;
    const uint16_t* vp = v + ((int64_t)b * S + key) * v_ts + (int64_t)g * D + 8 * h;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      kf[t] = as_bf8(*reinterpret_cast<const v4i*>(kp + 16 * t));
      vf[t] = as_bf8(*reinterpret_cast<const v4i*>(vp + 16 * t));
    }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    retire(kf[t]);
    retire(vf[t]);
  }
  f16v dka[DB], dva[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dka[d][i] = 0.0f; dva[d][i] = 0.0f; }

  const int sl0 = CAUSAL ? (kbi * 128) / 32 : 0;  // first query slice that can see this key block
  const int nsl = S / 32 - sl0;
  const int nsteps = rep * nsl;
  // staging: thread -> 2 chunks of the Q slice and 2 of the dO slice; threads 0..31 lse, 32..63 delta
  v4i qreg[2], dreg[2];
  float sreg = 0.0f;
  auto load_step = [&](int step) {
    const int head = g * rep + step / nsl, q0 = (sl0 + step % nsl) * 32;
    const uint16_t* qp = q + ((int64_t)b * S + q0) * q_ts + (int64_t)head * D;
    const uint16_t* dp = dout + ((int64_t)b * S + q0) * do_ts + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cidx = tid + 256 * i, row = cidx / (D / 8), ch = cidx % (D / 8);
      if (D == 128 || cidx < 32 * (D / 8)) {
        qreg[i] = *reinterpret_cast<const v4i*>(qp + (int64_t)row * q_ts + ch * 8);
        dreg[i] = *reinterpret_cast<const v4i*>(dp + (int64_t)row * do_ts + ch * 8);
      }
    }
    if (tid < 64) sreg = (tid < 32 ? lse : delta)[((int64_t)b * H + head) * S + q0 + (tid & 31)];
  };
  auto store_step = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cidx = tid + 256 * i, row = cidx / (D / 8), ch = cidx % (D / 8);
      if (D == 128 || cidx < 32 * (D / 8)) {
        *reinterpret_cast<v4i*>(buf + tile_off<D>(row, ch)) = qreg[i];
        *reinterpret_cast<v4i*>(buf + SL + tile_off<D>(row, ch)) = dreg[i];
      }
    }
    if (tid < 64) reinterpret_cast<float*>(buf + 2 * SL)[tid] = sreg;
  };
  load_step(0);
  store_step(lds);
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const char* buf = lds + (step & 1) * BUF;
    const bool more = step + 1 < nsteps;
    if (more) load_step(step + 1);
    const int q0 = (sl0 + step % nsl) * 32;
    if (!CAUSAL || q0 + 31 >= key_first) {  // wave-uniform: slices entirely before this wave's keys see none of them
      const char* qt = buf;
      const char* dt = buf + SL;
      const float* st = reinterpret_cast<const float*>(buf + 2 * SL);
      f16v sacc, pacc;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sacc[i] = 0.0f; pacc[i] = 0.0f; }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(qt, 0, t, lane), kf[t], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(dt, 0, t, lane), vf[t], pacc, 0, 0, 0);
      }
      const bool diag = CAUSAL && q0 < key_first + 31;
      f16v dsacc;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const v4f ls = *reinterpret_cast<const v4f*>(st + 8 * a + 4 * h);
        const v4f dl = *reinterpret_cast<const v4f*>(st + 32 + 8 * a + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * a + e;
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], c, -ls[e]));
          if (diag && key > q0 + acc_row(i, h)) p = 0.0f;
          sacc[i] = p;
          dsacc[i] = p * (pacc[i] - dl[e]);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf8 pf = acc_frag(sacc, s), dsf = acc_frag(dsacc, s);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          dva[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(dt, 16 * s, 32 * d, lane), pf, dva[d], 0, 0, 0);
          dka[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<D>(qt, 16 * s, 32 * d, lane), dsf, dka[d], 0, 0, 0);
        }
      }
    }
    // keep the loop-carried dK / dV accumulators in AGPRs: left to itself the allocator carries them in VGPRs and moves all
    // 128 registers to AGPRs and back around the MFMAs of every step (302 v_accvgpr moves per step against 32 MFMAs)
    // (head_dim 64 is bounded to 2 waves per SIMD = 256 registers, where the compiler uses the VGPR form of the MFMAs and
    // needs no AGPRs at all; with a bound of 1 it assumes AGPRs may be needed, selects the AGPR form and copies around it)
    if (D == 128) {
#pragma unroll
      for (int d = 0; d < DB; ++d) asm volatile("" : "+a"(dka[d]), "+a"(dva[d]));
    }
    if (more) store_step(lds + ((step + 1) & 1) * BUF);
    __syncthreads();
  }
  char* slice = lds + w * SL;
  store_rows_from_accT<D>(dka, scale, slice, dk + ((int64_t)b * S + key_first) * dk_ts + (int64_t)g * D, dk_ts, lane);
  __builtin_amdgcn_wave_barrier();
  store_rows_from_accT<D>(dva, 1.0f, slice, dv + ((int64_t)b * S + key_first) * dv_ts + (int64_t)g * D, dv_ts, lane);
}

}  // namespace mi

extern "C" int mi_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H, int G,
                           int D, int64_t q_ts, int64_t k_ts, int64_t v_ts, int64_t o_ts, float scale, int causal,
                           void* stream) {
  MI_CHECK_ARG(q && k && v && o && lse, "mi_attn_fwd: null pointer");
  MI_CHECK_ARG(B >= 1 && H >= 1 && G >= 1 && H % G == 0, "mi_attn_fwd: bad B/H/G (%d, %d, %d)", B, H, G);
  MI_CHECK_ARG(D == 128 || D == 64, "mi_attn_fwd: head_dim %d not supported (64, 128)", D);
  MI_CHECK_ARG(S >= 128 && S % 128 == 0, "mi_attn_fwd: seq %d must be a multiple of 128", S);
  MI_CHECK_ARG(q_ts % 8 == 0 && k_ts % 8 == 0 && v_ts % 8 == 0 && o_ts % 8 == 0, "mi_attn_fwd: token strides must be multiples of 8");
  MI_CHECK_ARG((int64_t)S * k_ts * 2 < (1LL << 31) && (int64_t)S * v_ts * 2 < (1LL << 31) && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0,
               "mi_attn_fwd: one batch of K / V must span < 2 GiB and be 16-byte aligned (32-bit buffer offsets of the LDS DMA)");
  MI_CHECK_ARG(H <= 65535 && B <= 65535, "mi_attn_fwd: grid too large");
  const float c = scale * 1.4426950408889634f;
  dim3 grid(S / mi::ATT_QB, H, B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MI_ATTN_FWD(DD, CC)                                                                                            \
  hipLaunchKernelGGL((mi::attn_fwd_kernel<DD, CC>), grid, block, 0, st, (const uint16_t*)q, (const uint16_t*)k,          \
                     (const uint16_t*)v, (uint16_t*)o, lse, S, H, G, q_ts, k_ts, v_ts, o_ts, c, (unsigned long long*)nullptr)
  if (D == 128) {
    if (causal) MI_ATTN_FWD(128, true); else MI_ATTN_FWD(128, false);
  } else {
    if (causal) MI_ATTN_FWD(64, true); else MI_ATTN_FWD(64, false);
  }
#undef MI_ATTN_FWD
  MI_CHECK_LAUNCH("mi_attn_fwd launch");
  return MI_OK;
}

#ifdef MI_DIAG
// timing-only diagnostic build of the forward kernel: per wave, cycles spent in {S^T MFMAs, softmax, P.V MFMAs, stage stores
// (incl. the wait for the next tile's global loads), barrier}; dbg [B, H, S/128, 4 waves, 8] u64
extern "C" int mi_attn_fwd_diag(const void* q, const void* k, const void* v, void* o, float* lse, unsigned long long* dbg, int B,
                                int S, int H, int G, int D, int64_t q_ts, int64_t k_ts, int64_t v_ts, int64_t o_ts, float scale,
                                void* stream) {
  MI_CHECK_ARG(q && k && v && o && lse && dbg, "mi_attn_fwd_diag: null pointer");
  MI_CHECK_ARG(D == 128 && S % 128 == 0 && H % G == 0, "mi_attn_fwd_diag: unsupported shape");
  dim3 grid(S / mi::ATT_QB, H, B), block(256);
  hipLaunchKernelGGL((mi::attn_fwd_kernel<128, true, true>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)q,
                     (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, lse, S, H, G, q_ts, k_ts, v_ts, o_ts,
                     scale * 1.4426950408889634f, dbg);
  MI_CHECK_LAUNCH("mi_attn_fwd_diag launch");
  return MI_OK;
}
#endif  // MI_DIAG

extern "C" int mi_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                           float* delta, void* dq, void* dk, void* dv, int B, int S, int H, int G, int D, int64_t q_ts,
                           int64_t k_ts, int64_t v_ts, int64_t o_ts, int64_t do_ts, int64_t dq_ts, int64_t dk_ts,
                           int64_t dv_ts, float scale, int causal, void* stream) {
  MI_CHECK_ARG(q && k && v && o && dout && lse && delta && dq && dk && dv, "mi_attn_bwd: null pointer");
  MI_CHECK_ARG(B >= 1 && H >= 1 && G >= 1 && H % G == 0, "mi_attn_bwd: bad B/H/G (%d, %d, %d)", B, H, G);
  MI_CHECK_ARG(D == 128 || D == 64, "mi_attn_bwd: head_dim %d not supported (64, 128)", D);
  MI_CHECK_ARG(S >= 128 && S % 128 == 0, "mi_attn_bwd: seq %d must be a multiple of 128", S);
  MI_CHECK_ARG((q_ts | k_ts | v_ts | o_ts | do_ts | dq_ts | dk_ts | dv_ts) % 8 == 0, "mi_attn_bwd: token strides must be multiples of 8");
  MI_CHECK_ARG((int64_t)S * k_ts * 2 < (1LL << 31) && (int64_t)S * v_ts * 2 < (1LL << 31) && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0,
               "mi_attn_bwd: one batch of K / V must span < 2 GiB and be 16-byte aligned (32-bit buffer offsets of the LDS DMA)");
  MI_CHECK_ARG(H <= 65535 && B <= 65535, "mi_attn_bwd: grid too large");
  const float c = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid_q(S / mi::ATT_QB, H, B), grid_kv((unsigned)((int64_t)(S / 128) * G * B)), block(256);
#define MI_ATTN_BWD(DD, CC)                                                                                            \
  do {                                                                                                                 \
    hipLaunchKernelGGL((mi::attn_bwd_dq_kernel<DD, CC>), grid_q, block, 0, st, (const uint16_t*)q, (const uint16_t*)k,   \
                       (const uint16_t*)v, (const uint16_t*)o, (const uint16_t*)dout, lse, delta, (uint16_t*)dq, S, H,  \
                       G, q_ts, k_ts, v_ts, o_ts, do_ts, dq_ts, c, scale);                                               \
    hipLaunchKernelGGL((mi::attn_bwd_dkdv_kernel<DD, CC>), grid_kv, block, 0, st, (const uint16_t*)q,                    \
                       (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)dout, lse, delta, (uint16_t*)dk,        \
                       (uint16_t*)dv, S, H, G, B, q_ts, k_ts, v_ts, do_ts, dk_ts, dv_ts, c, scale);                      \
  } while (0)
  if (D == 128) {
    if (causal) MI_ATTN_BWD(128, true); else MI_ATTN_BWD(128, false);
  } else {
    if (causal) MI_ATTN_BWD(64, true); else MI_ATTN_BWD(64, false);
  }
#undef MI_ATTN_BWD
  MI_CHECK_LAUNCH("mi_attn_bwd launch");
  return MI_OK;
}
