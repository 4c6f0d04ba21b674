/*
 * mi_fp8.h -- C ABI of libmi_fp8.so: the MI355X (gfx950) FP8 Linear hot path.
 *
 * This is the drop-in boundary of the build.  The reference
 * (xuanvinh1997/llm-fp8) has no FFI of its own: its FP8 arithmetic is reached
 * through Python calls into transformer_engine.pytorch modules
 *   te_llama.py:45-63,76-80      MultiheadAttention / LayerNormMLP under fp8_autocast
 *   te_llama_hybrid.py:39,75,78  single HYBRID recipe
 *   te_llama_mxfp8.py:28-29,86,93  MXFP8BlockScaling recipe
 *   accelerate utils/transformer_engine.py:52-59  nn.Linear -> te.Linear
 * Each entry point below names the TE-internal kernel
 * it replaces on that path.  INTEGRATION.md shows the ctypes stub that binds
 * them (llm_fp8_amd/_lib.py is that stub).
 *
 * Conventions
 *   - plain C, device pointers are `void*` / `float*` into HBM owned by the caller
 *     (PyTorch caching allocator); the library allocates nothing persistent.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no
 *     internal synchronisation, no global mutable state besides the thread-local
 *     error string -- with ONE documented exception: mi_gemm_set_workspace binds a
 *     caller-owned stream-K workspace to the current device (per-device table, used
 *     only by the explicit algo 44; never touched by the automatic choice).
 *   - return 0 on success, <0 on error: -1 invalid argument, -2 unsupported shape,
 *     -3 HIP runtime error.  mi_last_error() gives the message (thread-local).
 *   - matrices are row-major; `fmt`: 0 = OCP E4M3FN, 1 = OCP E5M2.
 *   - all GEMMs are "TN": D[M,N] = A[M,K] . B[N,K]^T with K contiguous in both
 *     operands; dgrad / wgrad use the transposed fp8 copies the cast kernels emit.
 */
#ifndef MI_FP8_H
#define MI_FP8_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_FMT_E4M3 0
#define MI_FMT_E5M2 1

#define MI_OK 0
#define MI_ERR_ARG (-1)
#define MI_ERR_SHAPE (-2)
#define MI_ERR_HIP (-3)

#define MI_OUT_BF16 0
#define MI_OUT_F32 1

#define MI_AMAX_ALGO_MAX 0
#define MI_AMAX_ALGO_MOST_RECENT 1

/* ABI version: bumped whenever an entry point is added, removed or changes its signature or the meaning of an argument.
 * mi_abi_version() returns the value the library was built with; a binding compares it with the header it was written
 * against (llm_fp8_amd/_lib.py does at load time).
 *   1  round 1 (the surface of SURVEY.md 8b + fused neighbours)
 *   2  round 2: mi_gemm_fp8 algo values 20-30, 46 (diagnostic builds), mi_adamw_cast_bf16_multi, mi_transpose_u8,
 *      mi_gemm_fp8_grouped
 *   3  round 2: mi_adamw_mxcast_bf16_multi
 *   4  round 3: diagnostic algo values and mi_attn_fwd_diag moved out (lab build, -DMI_DIAG); mi_gemm_fp8 algos 6, 9, 47 and
 *      mi_gemm_fp8_grouped tile_cfg 4 (four-wave kernel); mi_gemm_fp8_clock; mi_swiglu_cast_bias, mi_dswiglu_cast_bias,
 *      mi_add_bias_rmsnorm_stats (bias fused into the consumer of the GEMM output)
 */
#define MI_ABI_VERSION 4
int mi_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* mi_last_error(void);
/* 1 if the current HIP device is gfx950 (MI355X), else 0; <0 on HIP error. */
int mi_device_supported(void);

/*
 * K1/K2  cast (+transpose) + amax, delayed scaling  [replaces TE quantize / cast_transpose].
 *   y[r*ld_y + c]   = sat_cast_fmt(float(x[r*cols+c]) * *scale)          (RNE, NaN -> 0x7F)
 *   yT[c*ld_yT + r] = same byte                                           (if yT != NULL)
 *   *amax           = max(*amax, max|x|)   atomically (fmaxf semantics: NaN ignored)
 * x: bf16 [rows, cols] contiguous.  rows, cols multiples of 8.  ld_y >= cols, ld_yT >= rows.
 * y may be NULL when only the transposed copy is wanted.  amax may be NULL.
 */
int mi_cast_amax(const void* x_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                 int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, int fmt, void* stream);
/* yT[c * ld_yT + r] = y[r * ld_y + c] on FP8 bytes (rows, cols multiples of 8): the transposed copy of an operand that was
 * quantised elsewhere -- after an FP8 all-gather of row shards (llm_fp8_amd.distributed.ShardedFP8DP). */
int mi_transpose_u8(const void* y, void* yT, int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, void* stream);
/*
 * mi_cast_amax that also returns the column sums of x (the bias gradient when x = grad_output of a Linear,
 * `db = sum_M dy`): colsum_partial [ceil(rows/64), cols] fp32, one row per 64-row tile, to be reduced by
 * mi_colsum_finish.  Fixed summation order: reproducible.
 */
int mi_cast_amax_colsum(const void* x_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                        float* colsum_partial, int64_t rows, int64_t cols, int64_t ld_y, int64_t ld_yT, int fmt, void* stream);
/* out[c] = sum_p partial[p, c]; out bf16 (MI_OUT_BF16) or fp32 (MI_OUT_F32).  Second stage of every partial column sum
 * the library emits (mi_cast_amax_colsum, mi_dswiglu_cast, mi_mxfp8_dswiglu_quantize, mi_rmsnorm_bwd). */
int mi_colsum_finish(const float* partial, int64_t P, int64_t C, void* out, int out_dtype, void* stream);
/* Up to 4 mi_colsum_finish in one launch (host arrays of length n; same arithmetic and summation order per output). */
int mi_colsum_finish_multi(const void* const* partials, const int64_t* P, const int64_t* C, void* const* outs,
                           const int* out_dtypes, int n, void* stream);

/*
 * K3  amax-history roll + scale update for S slots in ONE launch
 *     [replaces TE fused_amax_and_scale_update_after_reduction].
 *   amax = max_h hist[h][s] (algo MAX) | hist[0][s] (MOST_RECENT)
 *   hist = roll(hist, -1, dim 0); hist[0][s] = 0
 *   sf = (fp8_max[s] / amax) / 2^margin ; keep old scale if !(amax > 0) or !isfinite(amax);
 *   sf = FLT_MAX if isinf(sf) ; scale[s] = sf ; scale_inv[s] = 1 / sf
 * amax_history: [H, S] fp32, row stride `ld` >= S floats (a used prefix of a larger arena).  H <= 4096.
 */
int mi_scale_update(float* amax_history, float* scale, float* scale_inv, const float* fp8_max,
                    int H, int S, int64_t ld, int margin, int algo, void* stream);

/*
 * K4/K5/K6  FP8 x FP8 -> bf16 GEMM on gfx950 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit scales)
 *           [replaces TE's cuBLASLt FP8 GEMM: fprop, dgrad, wgrad].
 *   D[m*ldd + n] = bf16( (sum_k A[m*lda+k] * B[n*ldb+k]) * (*sa_inv * *sb_inv) + bias[n] )
 * A: fp8 [M,K] fmt_a, B: fp8 [N,K] fmt_b, D: bf16 [M,N] (out_dtype 0) or fp32 (out_dtype 1).
 * bias: bf16 [N] or NULL.  M, N multiples of 16 (8 for the generic path), K multiple of 16.
 * algo: 0 = auto, 1 = generic 64x64 tile, 2 = 256x256 two-phase, 3 = 256x256 eight-phase ping-pong (eight waves),
 *       4 = persistent eight-phase (one workgroup per CU, epilogue overlapped with the next tile),
 *       5 = the kernel of 4 launched with one workgroup per tile (hardware-balanced: for use while collectives hold part of
 *           the chip), 6 = four-wave kernel (128x128 wave tiles, one wave per SIMD; mi_gemm_w4.hip), one tile per workgroup,
 *       9 = persistent four-wave kernel, 40-43 force one tile shape of 4, 44 = stream-K form of 4 (see mi_gemm_set_workspace),
 *       45 = 4, 47 = auto that takes the persistent four-wave kernel where it is the faster one (no bias, 256-multiples,
 *       K >= 512, and 256 x 256 is the eight-wave kernel's own tile choice), else as 0.
 * 2/3/6 need M,N % 256 == 0 and K % 128 == 0 (6, 9: K % 256 == 0); 4 additionally K % 256 == 0, bf16 output and operands
 * below 2 GiB (M, N may also be multiples of 192: workgroup tiles of 256/192 rows x 256/192 columns are picked per shape).
 * auto picks a persistent kernel when the shape allows, else 3, else 1.  Every algo listed here gives the SAME bits for the
 * same inputs (one fp32 summation order per output element), tests/test_kernels_gpu.py.
 * Not part of this library: the timing / ablation / stamp builds (algos 7, 8, 10-30, 46) live in the lab build
 * (tools/bin/libmi_fp8_lab.so, `make -C llm_fp8_amd/csrc lab`, compiled with -DMI_DIAG; see tools/README.md); the product
 * library returns MI_ERR_ARG for them.
 */
int mi_gemm_fp8(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv,
                const void* bias_bf16, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                int64_t ldd, int fmt_a, int fmt_b, int out_dtype, int algo, void* stream);

/*
 * Measurement aid (bench.py `roofline.clock_ghz`): the persistent production kernel (algo 0 / 4 / 9 as mi_gemm_fp8 would run
 * it, E4M3 x E4M3, no bias) with two clock reads around each workgroup's whole tile walk.  stamps: u64 [4 * workgroups] =
 * {shader cycles (s_memtime), 100 MHz ticks (s_memrealtime), K-tile steps walked, XCC id} per workgroup; the output D is
 * written as usual.  In-kernel clock = cycles / ticks x 100 MHz.
 */
int mi_gemm_fp8_clock(const void* A, const void* B, void* D, const float* sa_inv, const float* sb_inv, int64_t M, int64_t N,
                      int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int algo, unsigned long long* stamps, void* stream);

/*
 * Grouped form of mi_gemm_fp8: up to 4 independent problems D_p = (A_p . B_p^T) * (*sa_inv_p * *sb_inv_p), bf16 outputs, no
 * bias, in ONE persistent launch whose workgroups walk a shared tile list (longest tiles first) -- a Linear's dgrad + wgrad
 * (both consume the same grad_output) pay one ramp and one exposed epilogue, and the short problem's tiles fill
 * the idle part of the long one's last round of tiles.  All problems share fmt_a / fmt_b and the tile shape: tile_cfg
 * 0 = 256x256, 1 = 256x192, 2 = 192x256, 3 = 192x192, 4 = 256x256 on the four-wave kernel (every K_p >= 512), -1 = choose among 0-3 (M_p, N_p must be multiples of the tile, K_p of 256;
 * operands below 2 GiB; at most 64 tiles per workgroup).  Results are bitwise those of mi_gemm_fp8(algo 4) per problem.
 */
typedef struct mi_gemm_problem {
  const void* A;        /* fp8 [M, K], row stride lda */
  const void* B;        /* fp8 [N, K], row stride ldb */
  void* D;              /* bf16 [M, N], row stride ldd */
  const float* sa_inv;  /* device scalars */
  const float* sb_inv;
  int64_t M, N, K, lda, ldb, ldd;
} mi_gemm_problem;
int mi_gemm_fp8_grouped(const mi_gemm_problem* problems, int n, int fmt_a, int fmt_b, int tile_cfg, void* stream);

/*
 * Stream-K workspace for the persistent GEMM, used only by the explicit algo 44: when the 256x256 tile count is not a
 * multiple of the CU count, the (tile, K-tile) steps are cut into equal ranges and a tile split between two workgroups is
 * summed through this buffer (fp32 partial accumulators + one flag per workgroup).  The caller owns the memory: at least
 * mi_gemm_workspace_bytes(), 256-byte aligned, its first 4 KiB zeroed once; it is bound to the CURRENT device until replaced
 * (NULL unregisters) and must only be used by GEMMs of one stream at a time.
 */
int64_t mi_gemm_workspace_bytes(void);
int mi_gemm_set_workspace(void* workspace, int64_t bytes);

/*
 * K7  MXFP8 block quantise  [replaces TE's MXFP8 quantize, row-wise and column-wise].
 *   Row-wise: one E8M0 scale per 32 consecutive elements of a row:
 *     e = roundup_e8m0(amax_blk * (1/fp8_max)); y = sat_cast(x * 2^(127-e))
 *     y_row [rows, cols] fp8, s_row [cols/32, rows] u8 (BLOCK-MAJOR: the scales of one 32-block of every row are contiguous).
 *   Column-wise (blocks of 32 along rows), emitted TRANSPOSED so it is a TN GEMM operand:
 *     y_colT [cols, rows] fp8, s_colT [rows/32, cols] u8 (block-major w.r.t. the transposed operand).
 * Either pair may be NULL.  rows, cols multiples of 32.
 */
int mi_mxfp8_quantize(const void* x_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT,
                      int64_t rows, int64_t cols, int fmt, void* stream);

/*
 * mi_mxfp8_quantize for a tensor that is a ROW-BLOCK of a larger operand (query | key | value weights forming one GEMM
 * operand, te_llama.py:200-217): y_row / s_colT point at the block's first row inside the larger [R_total, cols] buffers
 * (contiguous); s_row / y_colT point at the block's first column and use ld_rows (= R_total, >= rows) as their leading
 * dimension.  colsum (optional): fp32 [ceil(rows/128), cols] partial column sums of x, one row per 128-row tile, for
 * mi_colsum_finish (bias gradient riding on the quantisation of grad_output).
 */
int mi_mxfp8_quantize_ex(const void* x_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT, float* colsum,
                         int64_t rows, int64_t cols, int64_t ld_rows, int fmt, void* stream);
/*
 * K7 fused with the GEMM neighbours (MXFP8 counterparts of K9 / K10; same output layout as mi_mxfp8_quantize):
 *   mi_mxfp8_norm_quantize     quantises (x * rstd[r]) * gamma[c]                      (RMSNorm -> MXFP8)
 *   mi_mxfp8_swiglu_quantize   quantises silu(h[:, :F]) * h[:, F:]                     (outputs have F columns)
 *   mi_mxfp8_dswiglu_quantize  quantises [dact*dsilu(g)*u | dact*silu(g)] ([rows, 2F]) and writes the per-128-row
 *                              column sums (fp32 [ceil(rows/128), 2F], nullable) for the fc1 bias gradient
 */
int mi_mxfp8_norm_quantize(const void* x_bf16, const float* rstd, const void* gamma_bf16, void* y_row, void* s_row,
                           void* y_colT, void* s_colT, int64_t rows, int64_t cols, int fmt, void* stream);
int mi_mxfp8_swiglu_quantize(const void* h_bf16, void* y_row, void* s_row, void* y_colT, void* s_colT, int64_t rows,
                             int64_t F, int fmt, void* stream);
int mi_mxfp8_dswiglu_quantize(const void* h_bf16, const void* dact_bf16, void* y_row, void* s_row, void* y_colT,
                              void* s_colT, float* colsum, int64_t rows, int64_t F, int fmt, void* stream);
/*
 * RoPE backward (mi_rope_qkv, backward = 1) fused with mi_mxfp8_quantize of its result: the MXFP8 grad_output of the q|k|v
 * projection under MXFP8BlockScaling (te_llama_mxfp8.py:28-29,86), straight from the attention gradients dq / dk / dv; the
 * bf16 d(qkv) [rows, W] is never written.  Outputs as mi_mxfp8_quantize for a [rows, W] tensor, W = (n_q + 2 n_kv) * 128;
 * equal to the two-kernel sequence bit for bit.
 */
int mi_mxfp8_rope_bwd_quantize(const void* dq_bf16, const void* dk_bf16, const void* dv_bf16, const float* cos_tab,
                               const float* sin_tab, void* y_row, void* s_row, void* y_colT, void* s_colT, int64_t rows,
                               int64_t seq, int n_q_heads, int n_kv_heads, int head_dim, int fmt, void* stream);

/*
 * K8  block-scaled MXFP8 GEMM (v_mfma_scale_f32_16x16x128_f8f6f4 with per-32 E8M0 scales)
 *   D[m,n] = bf16( sum_blk 2^(sa[m,blk]+sb[n,blk]-254) * sum_{k in blk} A[m,k] B[n,k] + bias[n] )
 * A [M,K] fp8 + SA [K/32, M] u8; B [N,K] fp8 + SB [K/32, N] u8 (block-major, as mi_mxfp8_quantize emits); K multiple of 32.
 * algo: 0 = auto (persistent 256x256 kernel when M,N,K % 256 == 0 and bf16 output; else generic), 1 = generic.
 */
int mi_gemm_mxfp8(const void* A, const void* SA, const void* B, const void* SB, void* D,
                  const void* bias_bf16, int64_t M, int64_t N, int64_t K, int fmt_a, int fmt_b,
                  int out_dtype, int algo, void* stream);

/*
 * RoPE fused with the q/k/v split of the QKV projection output  [TE applies a fused RoPE kernel between
 * layernorm_qkv and the attention core on the reference path, te_llama.py:65-66,77].
 *   forward  (backward = 0): q, k, v <- fused [rows, (nq + 2 nkv) * D]; q, k heads rotated by +theta(row % seq)
 *   backward (backward = 1): fused gradient <- dq, dk, dv with the conjugate rotation
 * out1 = x1 cos - x2 sin, out2 = x2 cos + x1 sin over the two halves of a head (non-interleaved);
 * cos_tab / sin_tab: fp32 [>= seq, D/2]; math in fp32, one bf16 rounding.
 */
int mi_rope_qkv(void* fused_bf16, void* q_bf16, void* k_bf16, void* v_bf16, const float* cos_tab,
                const float* sin_tab, int64_t rows, int64_t seq, int n_q_heads, int n_kv_heads, int head_dim,
                int backward, void* stream);
/*
 * RoPE backward (mi_rope_qkv with backward = 1) fused with mi_cast_amax of its result: dq [rows, n_q*D], dk, dv
 * [rows, n_kv*D] bf16 -> the FP8 copies y [rows, W] / yT [W, rows] (W = (n_q + 2 n_kv) * D) of the fused gradient d(qkv),
 * which is grad_output of the q|k|v projection (te_llama.py:45-56: `layernorm_qkv` feeding the rotary attention core), plus
 * its amax.  The bf16 gradient is never written.  Bytes and amax equal the two-kernel sequence bit for bit.  head_dim 128.
 */
int mi_rope_qkv_bwd_cast(const void* dq_bf16, const void* dk_bf16, const void* dv_bf16, const float* cos_tab, const float* sin_tab,
                         void* y_fp8, void* yT_fp8, const float* scale, float* amax, int64_t rows, int64_t seq, int n_q_heads,
                         int n_kv_heads, int head_dim, int fmt, void* stream);

/*
 * K10  SwiGLU fused with the FP8 cast of fc2's input  [TE LayerNormMLP activation="swiglu", te_llama.py:58-63].
 *   act[r,c] = silu(h[r,c]) * h[r,F+c] in fp32, c in [0,F);  y = sat_cast(act * *scale), yT its transpose,
 *   *amax = max(*amax, max|act|).  h: bf16 [rows, 2F] (gate | up).  rows, F multiples of 8.
 */
int mi_swiglu_cast(const void* h_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                   int64_t rows, int64_t F, int fmt, void* stream);

/*
 * K10 backward: dh = [dact * dsilu(g) * u | dact * silu(g)] in fp32 -> FP8 [rows, 2F] (+ transpose) + amax.
 * colsum (nullable): fp32 [ceil(rows/128), 2F] per-row-block column sums of dh in a fixed order (the caller adds
 * the row blocks to get the fc1 bias gradient bitwise reproducibly).
 */
int mi_dswiglu_cast(const void* h_bf16, const void* dact_bf16, void* y_fp8, void* yT_fp8, const float* scale,
                    float* amax, float* colsum, int64_t rows, int64_t F, int fmt, void* stream);
/* The two SwiGLU kernels with the fc1 bias (bf16 [2F], 16-byte aligned) added to `h` INSIDE the kernel, in fp32, before the
 * activation: `h` is then the fc1 GEMM's output WITHOUT its bias.  TE's bias + activation fusion (LayerNormMLP keeps TE's default
 * bias=True on the reference path, te_llama.py:58-63): the add rides in an HBM-bound kernel instead of in the GEMM epilogue. */
int mi_swiglu_cast_bias(const void* h_bf16, const void* bias_bf16, void* y_fp8, void* yT_fp8, const float* scale, float* amax,
                        int64_t rows, int64_t F, int fmt, void* stream);
int mi_dswiglu_cast_bias(const void* h_bf16, const void* bias_bf16, const void* dact_bf16, void* y_fp8, void* yT_fp8,
                         const float* scale, float* amax, float* colsum, int64_t rows, int64_t F, int fmt, void* stream);

/*
 * K9  RMSNorm fused with the FP8 cast of the following GEMM's input  [TE LayerNormLinear / LayerNormMLP with
 *     normalization="RMSNorm", te_llama.py:45-63: the normalised bf16 activation is never materialised].
 *   mi_rmsnorm_stats: rstd[r] = rsqrt(mean_c x[r,c]^2 + eps)                                   (fp32, one wave per row)
 *   mi_norm_cast:     v = (x[r,c] * rstd[r]) * gamma[c] in fp32; y = sat_cast(v * *scale), yT, *amax = max(*amax, max|v|)
 *   mi_rmsnorm_bwd:   dx = rstd * (dy*gamma - xhat * mean_c(dy*gamma*xhat)) (+ dres), xhat = x*rstd;
 *                     dgamma_partial[b, c] = sum over block b's rows of dy*xhat (fp32, fixed order; caller adds the blocks).
 *                     cols % 512 == 0, cols <= 8192.
 */
int mi_rmsnorm_stats(const void* x_bf16, float* rstd, int64_t rows, int64_t cols, float eps, void* stream);
/* out = a + b (bf16, one rounding of the fp32 sum) and rstd[r] of the rounded sum in one pass: the residual add of a decoder
 * layer (te_llama.py:78,81) fused with the statistics pass of the RMSNorm that consumes it. */
int mi_add_rmsnorm_stats(const void* a_bf16, const void* b_bf16, void* out_bf16, float* rstd, int64_t rows, int64_t cols,
                         float eps, void* stream);
/* mi_add_rmsnorm_stats with a bias (bf16 [cols]) on the second addend: out = a + (b + bias) -- `b` is the fc2 GEMM's output without
 * its bias, which joins the residual add it feeds anyway. */
int mi_add_bias_rmsnorm_stats(const void* a_bf16, const void* b_bf16, const void* bias_bf16, void* out_bf16, float* rstd,
                              int64_t rows, int64_t cols, float eps, void* stream);
int mi_norm_cast(const void* x_bf16, const float* rstd, const void* gamma_bf16, void* y_fp8, void* yT_fp8,
                 const float* scale, float* amax, int64_t rows, int64_t cols, int fmt, void* stream);
int mi_rmsnorm_bwd(const void* dy_bf16, const void* x_bf16, const float* rstd, const void* gamma_bf16,
                   const void* dres_bf16, void* dx_bf16, float* dgamma_partial, int n_partials, int64_t rows,
                   int64_t cols, void* stream);

/*
 * Token cross-entropy on the bf16 logits of the FP8 lm_head (the loss of the reference loop, HF ForCausalLMLoss:
 * mean over tokens whose label != -100; train_fp8.py:278-279 `outputs.loss`), without an fp32 copy of the logits.
 *   mi_ce_forward : lse[r] = log sum_c exp(x[r,c]); loss_rows[r] = lse[r] - x[r, labels[r]] (0 where the label is ignored)
 *   mi_ce_backward: dlogits[r,c] = (exp(x[r,c] - lse[r]) - [c == labels[r]]) * *gscale (0 on ignored rows);
 *                   *gscale = upstream gradient / number of valid tokens (device scalar).  cols % 8 == 0.
 */
int mi_ce_forward(const void* logits_bf16, const int64_t* labels, float* lse, float* loss_rows, int64_t rows,
                  int64_t cols, void* stream);
int mi_ce_backward(const void* logits_bf16, const int64_t* labels, const float* lse, const float* gscale,
                   void* dlogits_bf16, int64_t rows, int64_t cols, void* stream);
/*
 * mi_ce_backward fused with mi_cast_amax of its result: d(logits) leaves as the FP8 copies y [rows, cols] / yT [cols, rows]
 * (+ amax) that the lm_head Linear's backward quantises its grad_output into (train_fp8.py:276-280: `outputs.loss` of the
 * FP8 lm_head); the bf16 d(logits) is never written.  Bytes and amax equal the two-kernel sequence bit for bit.
 */
int mi_ce_backward_cast(const void* logits_bf16, const int64_t* labels, const float* lse, const float* gscale, void* y_fp8,
                        void* yT_fp8, const float* scale, float* amax, int64_t rows, int64_t cols, int fmt, void* stream);

/*
 * Optimiser step of the reference loop (train_fp8.py:288-291: clip_grad_norm_(model.parameters(), 1.0) then
 * AdamW(fused=True).step()), used by llm_fp8_amd.train on the single-GPU path.
 *   mi_sumsq_bf16: partial[b] = sum of g^2 over block b's grid-stride share, b < n_partials (fixed order: reproducible).
 *   mi_adamw_bf16: torch.optim.AdamW update on bf16 p / exp_avg / exp_avg_sq with fp32 math; the gradient is
 *                  multiplied by *grad_scale (device scalar, NULL = 1) -- the clip coefficient, so the gradients are
 *                  not rescaled in a separate pass.  `step` is the 1-based step count for the bias corrections.
 */
int mi_sumsq_bf16(const void* g_bf16, int64_t n, float* partial, int n_partials, void* stream);
/*
 * Multi-tensor forms: one launch for every tensor of a parameter group.  table [5, n_tensors] int64 ON THE DEVICE = rows of
 * p / g / exp_avg / exp_avg_sq addresses and element counts; chunks [n_chunks, 2] int32 on the device = (tensor index, chunk
 * index inside the tensor); each workgroup handles one chunk of chunk_elems (multiple of 8) elements.
 * mi_sumsq_bf16_multi writes partial[n_chunks] in FLOAT64 (exact fp32 squares accumulated in fp64: the total is independent of the
 * chunking and of how a caller groups or shards the tensors, to 2^-52); mi_adamw_bf16_multi = mi_adamw_bf16 on every tensor.
 */
int mi_sumsq_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                        double* partial, void* stream);
int mi_adamw_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                        const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int64_t step, void* stream);
/*
 * mi_adamw_bf16_multi that ALSO emits the FP8 copies the next forward would cast (K1/K2 folded into the optimiser pass).
 * table: int64 [12, n_tensors]: rows 0-4 as above (p, g, exp_avg, exp_avg_sq, numel), then per tensor
 *   5 cols (0 = no FP8 sink: flat chunks of chunk_elems; > 0: the tensor is a [numel / cols, cols] weight and its chunks are
 *     128 x 128 tiles, chunk index = tile_row * ceil(cols / 128) + tile_col),
 *   6 y (fp8 [rows, ld_y] or 0), 7 yT (fp8 [cols, ld_yT] or 0), 8 ld_y, 9 ld_yT, 10 const float* scale, 11 float* amax (or 0).
 * For a sink tensor every element is updated, rounded to bf16 and stored, and that rounded value is quantised exactly as
 * mi_cast_amax would: y = sat_cast_e4m3(float(bf16) * *scale), *amax = max(*amax, |bf16|) -- bitwise the bytes of a
 * mi_cast_amax call on the updated weight.  rows and cols multiples of 8.
 */
int mi_adamw_cast_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                             const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t step, void* stream);
/*
 * The MXFP8 form of mi_adamw_cast_bf16_multi (ABI 3): block scaling has no state, so the copies the next forward would
 * quantise (K7 on every weight: row-wise blocks for fprop, column-wise blocks stored transposed for dgrad) leave with the update.
 * table: int64 [12, n_tensors]: rows 0-5 as above (cols > 0: a [numel / cols, cols] weight walked in 128 x 128 tiles), then
 *   6 y_row (fp8 e4m3 [rows, cols]), 7 s_row (E8M0 [cols/32, ldr]), 8 y_colT (fp8 [cols, ldr]), 9 s_colT (E8M0 [rows/32, cols]),
 *   10 ldr (row count of the operand the tensor is a row-block of: query | key | value form one operand), 11 unused.
 * Rows 6-9 point at the tensor's first row inside the operand's buffers.  Bitwise the bytes of mi_mxfp8_quantize(_ex) on the
 * updated, bf16-rounded weight.  rows and cols multiples of 32; the row offset of a part a multiple of 32.
 */
int mi_adamw_mxcast_bf16_multi(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                               const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int64_t step, void* stream);

/*
 * Embedding weight gradient added in place: grad[id, :] += alpha * sum_{tokens t with ids[t] == id} dY[t, :]   (bf16, fp32
 * sums).  The tied lm_head / embedding table of the reference's Llama models (te_llama.py: `tie_word_embeddings`) already
 * holds the lm_head wgrad when the embedding's backward runs; this replaces a dense [vocab, hidden] scatter + add by a pass
 * over the touched rows.  sorted_ids / perm: the token ids sorted ascending (stable) and the permutation that sorts them
 * (row perm[i] of dY belongs to sorted_ids[i]); ids outside [0, vocab) and padding_idx (-1: none) are skipped.
 * Deterministic (fixed summation order, one writer per row).
 */
int mi_embedding_grad_add(void* grad_bf16, const void* dy_bf16, const int64_t* sorted_ids, const int64_t* perm, int64_t tokens,
                          int64_t hidden, int64_t vocab, float alpha, int64_t padding_idx, void* stream);
int mi_adamw_bf16(void* p_bf16, const void* g_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16, int64_t n,
                  const float* grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, void* stream);

/*
 * Attention core of te.pytorch.MultiheadAttention as the reference configures it (te_llama.py:45-56: causal mask type,
 * num_gqa_groups, qkv_format="bshd", attention_dropout 0; TE dispatches to flash-attn, README.md:27-28).  bf16 in/out,
 * fp32 softmax statistics, scores never written to memory.
 *   q, o [B, S, H, D]; k, v [B, S, G, D] (H % G == 0), D contiguous, `*_ts` = token stride in elements (multiple of 8), so
 *   the operands may be column slices of one fused [tokens, (H + 2G) * D] buffer.  D in {64, 128}, S % 128 == 0.
 *   lse [B, H, S] fp32 = log2-domain log-sum-exp (max * c + log2(sum), c = scale * log2(e)); kept for the backward.
 */
int mi_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H, int G, int D,
                int64_t q_ts, int64_t k_ts, int64_t v_ts, int64_t o_ts, float scale, int causal, void* stream);
#ifdef MI_DIAG
/* LAB BUILD ONLY (-DMI_DIAG).  Timing-only diagnostic build of the forward kernel (causal, D = 128): per wave, shader cycles spent in {S^T MFMAs + K fragment
 * reads, softmax arithmetic, P.V MFMAs + V fragment reads, stage stores incl. the wait for the next tile's loads, barrier};
 * dbg [B, H, S/128, 4 waves, 8] u64 (slot 5 = tiles walked).  Shares only: the stamps forbid overlaps the real kernel has. */
int mi_attn_fwd_diag(const void* q, const void* k, const void* v, void* o, float* lse, unsigned long long* dbg, int B, int S,
                     int H, int G, int D, int64_t q_ts, int64_t k_ts, int64_t v_ts, int64_t o_ts, float scale, void* stream);
#endif
/*
 * Backward of mi_attn_fwd: P is recomputed from q, k and `lse`; two launches (dQ pass, which also writes
 * delta[B, H, S] = rowsum(dO * O), then the dK/dV pass), no sums across workgroups: results are bitwise reproducible.
 * dq [B, S, H, D], dk / dv [B, S, G, D] bf16 with their own token strides (they may be slices of one fused buffer).
 */
int mi_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                float* delta, void* dq, void* dk, void* dv, int B, int S, int H, int G, int D, int64_t q_ts,
                int64_t k_ts, int64_t v_ts, int64_t o_ts, int64_t do_ts, int64_t dq_ts, int64_t dk_ts, int64_t dv_ts,
                float scale, int causal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_FP8_H */
