/*
 * i2v_hip.h -- C ABI of libi2v_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * I2V-Adapter denoising path.
 *
 * The reference (xUhEngwAng/I2V-Adapter-Unofficial) has no FFI: the path sits behind a Python module
 * API and its arithmetic is executed by torch/aten kernels reached through `diffusers`.
 * Each entry point below replaces the aten kernel family named in its comment, at the reference call
 * site cited (file:line relative to /root/reference; i2v = src/modules/i2v_adapter.py,
 * unet = src/models/unet_motion_cross_frame_attn.py, pipe = src/pipelines/pipeline_i2v_adapter.py).
 *
 * Conventions
 *   - plain C: raw device pointers + explicit sizes/strides; no torch types.
 *   - activations are fp16, TOKEN-MAJOR CHANNELS-LAST: x[image][pixel][channel]; strides are in ELEMENTS.
 *   - every function is asynchronous on the caller's `stream` (a hipStream_t), never allocates, never
 *     synchronises, never touches the null stream => safe inside hipGraph capture.
 *   - return value: 0 ok, <0 error (I2V_ERR_*); i2v_last_error() returns a thread-local message.
 *   - the caller owns every buffer (including workspaces, whose sizes the *_workspace_bytes helpers give).
 */
#ifndef I2V_HIP_H
#define I2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2V_ABI_VERSION 9

#define I2V_OK 0
#define I2V_ERR_INVALID_ARG (-1)
#define I2V_ERR_UNSUPPORTED (-2)
#define I2V_ERR_LAUNCH (-3)

typedef void* i2v_stream_t; /* hipStream_t */

int i2v_abi_version(void);
const char* i2v_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution with fused epilogue (MFMA, fp16 in, fp32 accumulate, fp16 out)
 *
 *   C[m, n] = epi( sum_k A[m, k] * W[n, k]  + bias[n] + rowvec[m / rows_per_vec, n] + residual[m', n] )
 *             * out_scale
 *
 * replaces aten addmm / conv2d(+bias) at: attention projections to_q/to_k/to_v/to_out (diffusers
 * `Attention`, constructed i2v:32-37,409-418), GEGLU feed-forward (i2v:554), 1x1 proj_in/proj_out
 * (i2v:219-226,299-305), ResnetBlock2D conv1/conv2/conv_shortcut/time_emb_proj (unet:203-214,384-395,
 * 563-574,594-605), Downsample2D/Upsample2D convs (unet:250-259,431-432), conv_in/conv_out
 * (unet:757-759,879-881), motion-module proj_in/proj_out/q/k/v/out/ff (unet:232-244), time embedding MLP
 * (unet:766-770), ImageProjection (unet:1284-1287).
 * ------------------------------------------------------------------------------------------------ */
enum {
  I2V_EPI_NONE = 0,
  I2V_EPI_GELU = 1,  /* erf GELU on the result                                                     */
  I2V_EPI_GEGLU = 2  /* W rows interleaved (value_i, gate_i): C[m, n/2] = v * gelu_erf(g); N even   */
};

enum {
  I2V_STORE_ROWMAJOR = 0, /* C[m * ldc + n]                                                         */
  I2V_STORE_ROWPERM = 1,  /* rows arrive in (b, pixel, frame) order and are stored (and the residual
                             is read) in (b, frame, pixel) order: m = (b*hw + p)*frames + f ->
                             m' = (b*frames + f)*hw + p            (motion-module exit)   */
  I2V_STORE_VT = 2,       /* transposed, batched: element (m, n) -> C[((n / vt_len) * M + m) * vt_ld
                             + n % vt_len]; used with A = weight [C, K], W = tokens [T, K] to emit
                             V^T[batch][channel][key] for the attention kernels                     */
  I2V_STORE_VT_T = 3      /* the same V^T[batch][channel][key] output from the NATURAL operand order
                             (A = tokens [T, K], W = weight [C, K]): element (m, n) -> C[((m / vt_len) * N + n)
                             * vt_ld + m % vt_len]; lets large V projections run on the 256-row tile kernel  */
};

enum {
  I2V_A_PLAIN = 0,  /* A is [M, K] row-major (optionally two sources concatenated along K)          */
  I2V_A_CONV3X3 = 1 /* A is gathered on the fly from an NHWC image: K = 9 * cin, k = tap * cin + ci */
};

typedef struct i2v_gemm_params {
  const void* a;        /* fp16 */
  int64_t lda;
  const void* a2;       /* optional second K-range source (skip-connection concat, unet:478); NULL if unused */
  int64_t lda2;
  int32_t k_split;      /* columns [0, k_split) come from a, [k_split, K) from a2; multiple of 8     */
  int32_t a_mode;       /* I2V_A_*                                                                  */
  const void* w;        /* fp16 [N, K] row-major (torch Linear / repacked conv weight)               */
  int64_t ldw;
  const void* bias;     /* fp16 [N] or NULL                                                          */
  const void* residual; /* fp16 [M, N] (ld = ldr) or NULL                                            */
  int64_t ldr;
  const void* rowvec;   /* fp16 [M / rows_per_vec, N] (ld = ld_rowvec) or NULL: time-embedding add   */
  int64_t ld_rowvec;
  int32_t rows_per_vec;
  int32_t rowvec_period; /* > 0 (a power of two): the vector of row m is rowvec[m % rowvec_period] (positional-embedding
                            table of the motion modules: tokens are in (b, pixel, frame) order, period = frames);
                            with I2V_STORE_VT_T the table is passed TRANSPOSED, [N, ld_rowvec >= period]          */
  /* LayerNorm folded into the GEMM (i2v:444-445, 510, 539 and the motion modules' norm1/2/3).  With W' = W o gamma
     (the w passed here), ln_wsum fp32 [N] = sum_k W'[n][k] (summed from the fp16-ROUNDED W') and bias = W beta + b:
        C[m][n] = rstd_m (sum_k A[m][k] W'[n][k] - mean_m ln_wsum[n]) + bias[n]   = (LayerNorm(A) W^T + b)[m][n]
     where mean_m / rstd_m = 1 / sqrt(var_m + ln_eps) are the statistics of row m of A over its K columns, computed
     INSIDE the kernel from the A tiles its K loop streams anyway: the normalised activations are never written to /
     read back from HBM and there is no statistics pass.  NULL = plain GEMM.  Only the 8-wave LDS-DMA kernel
     implements it (single-source A, no residual): i2v_gemm_ln_supported() tells whether a problem qualifies. */
  const void* ln_wsum;
  float ln_eps;
  void* c;              /* fp16                                                                      */
  int64_t ldc;
  int32_t M, N, K;
  int32_t epilogue;     /* I2V_EPI_*                                                                 */
  int32_t store_mode;   /* I2V_STORE_*                                                               */
  int32_t frames, hw;   /* I2V_STORE_ROWPERM                                                         */
  int32_t vt_len, vt_ld;/* I2V_STORE_VT / I2V_STORE_VT_T                                             */
  float out_scale;
  /* I2V_A_CONV3X3 geometry: input image [n_img, in_h, in_w, cin] fp16 (pixel stride = lda elements),
     3x3 kernel, padding 1, `stride` 1 or 2; `upsample` = 1 applies nearest-2x to the input first
     (Upsample2D).  `asym_pad` = 1 (stride 2 only): no padding at the top / left and one zero row / column at the
     bottom / right = diffusers Downsample2D(padding=0) of the VAE encoder (F.pad(x, (0, 1, 0, 1)) + conv stride 2).
     M = n_img * out_h * out_w. */
  int32_t n_img, in_h, in_w, cin, out_h, out_w, stride, upsample, asym_pad;
  /* order of the 9 * cin contraction index of I2V_A_CONV3X3 (and of w's columns): 0 = tap-major, k = tap * cin + ci;
     64 = channel-block-major, k = ((ci / 64) * 9 + tap) * 64 + ci % 64 (cin % 64 == 0): the 9 taps of one 64-channel
     block are consecutive K tiles, so the 8 re-reads of an input pixel follow each other closely (measured: conv class
     12.40 -> 12.24 ms per step). */
  int32_t conv_kblock;
  /* Per-batch weights (GroupNorm folded into the proj_in GEMM of a transformer / motion-module entry, i2v:218-226, A9:
     the norm's per-image scale multiplies the weights, its shift becomes a per-image bias -- i2v_groupnorm_fold_f16):
     rows [i * rows_per_w, (i + 1) * rows_per_w) of A use the weight matrix w + i * w_batch_stride (elements).
     rows_per_w = 0: one weight matrix.  8-wave kernel only (N % 320 == 0, M and rows_per_w multiples of the row tile). */
  int64_t w_batch_stride;
  int32_t rows_per_w;
  /* A rows gathered through the (batch, frame, pixel) -> (batch, pixel, frame) permutation (the motion modules' entry,
     SURVEY A9): output row m = (b * a_perm_hw + p) * a_perm_frames + f reads A row (b * a_perm_frames + f) * a_perm_hw + p.
     a_perm_frames = 0: none; otherwise a power of two <= 64.  Same kernel restriction. */
  int32_t a_perm_frames, a_perm_hw;
  /* optional fp32 scratch for split-K (small M, long K: the 8 x 8 level's convolutions): when it holds at least
     i2v_gemm_workspace_bytes(p) bytes the K loop is split over several workgroups whose fp32 partial tiles are
     summed by a second kernel that applies the epilogue; NULL / too small => no split. */
  void* workspace;
  int64_t workspace_bytes;
  /* 1: c is FP32 [M, N] (ld = ldc elements) instead of fp16.  Row-major store, no GEGLU; narrow outputs only (the
     generic 4-wave kernel): the UNet's 4-channel conv_out (unet:879-881, 1443) hands the noise prediction to
     i2v_ddim_cfg_step without an fp16 rounding that the CFG combine (pipe:686-688) would amplify by up to
     2 * guidance_scale - 1. */
  int32_t c_is_f32;
  /* (ABI 8) optional: GroupNorm statistics of the result for the norm that consumes it (ResnetBlock2D conv1 -> norm2, unet:203-214;
     diffusers ResnetBlock2D), written by the epilogue so that i2v_groupnorm_f16 needs no statistics pass over the tensor:
     fp32 [n_img][out_h * out_w / R][gn_groups][2] = per (image, block of R consecutive rows, group) the mean and the sum of squared
     deviations over the block's rows x the group's channels, R = i2v_gemm_gn_partial_rows(p) (> 0 where implemented: un-split
     I2V_A_CONV3X3 problems of the 8-wave kernel, no residual, images of whole row blocks).  Hand the buffer and R to
     i2v_gn_params.gpartial_in / gpartial_rows.  NULL: none. */
  void* gn_partial;
  int32_t gn_groups;
  /* (ABI 9) the PRECISE residual stream: the stream between modules (outputs of conv_in, every ResnetBlock2D, spatial transformer,
     motion module, down sampler; pipe:666-697 runs it in fp32 on the reference's CPU path) as an fp16 PAIR hi + lo, hi = fp16(x),
     lo = fp16(x - hi): hi is the tensor every MFMA / LDS-DMA operand reads as before, lo carries the 11 bits the rounding dropped, so
     the identity path of a residual add no longer re-rounds the stream at every module (0.78 of the 0.89e-3 rms).
       residual_lo  fp16 [M, N] (ld = ldr), needs `residual`: the value added is (float)residual + (float)residual_lo
       c_lo         fp16 [M, N] (ld = ldc, row order of c): receives fp16(v - (float)fp16(v)) of the fp32 result v that c rounds
     Row-major / row-permuted fp16 stores without GEGLU / GELU, no LayerNorm fold, no GroupNorm partials.  NULL: none. */
  const void* residual_lo;
  void* c_lo;
} i2v_gemm_params;

int i2v_gemm_f16(const i2v_gemm_params* p, i2v_stream_t stream);
/* 1 if i2v_gemm_f16 accepts this problem with ln_wsum set (pointers are not dereferenced), else 0. */
int i2v_gemm_ln_supported(const i2v_gemm_params* p);
/* 1 if i2v_gemm_f16 accepts this problem with rows_per_w / a_perm_frames set (pointers are not dereferenced), else 0. */
int i2v_gemm_batch_supported(const i2v_gemm_params* p);
/* bytes of `workspace` with which i2v_gemm_f16 would split K for this problem (0: it would not split). */
int64_t i2v_gemm_workspace_bytes(const i2v_gemm_params* p);
/* rows per block of the GroupNorm partials i2v_gemm_f16 would write for this problem with gn_partial set (gn_groups must be set;
   pointers are not dereferenced); 0: not implemented for it. */
int32_t i2v_gemm_gn_partial_rows(const i2v_gemm_params* p);

/* ------------------------------------------------------------------------------------------------
 * Flash-style attention forward (MFMA QK^T / PV, wavefront-shuffle online softmax).
 *   O[bq, l, h, :] (+)= softmax_j( scale * Q[bq, l, h, :] . K[bq / kv_group, j, h, :] ) V[bq / kv_group, j, h, :]
 * replaces F.scaled_dot_product_attention at:
 *   K1 cross-frame adapter attention i2v:483-492  (kv_group = num_frames: every frame reads frame-0 K/V)
 *   K2 spatial self-attention        i2v:468-473  (kv_group = 1)
 *   K3 text cross-attention + IP-Adapter decoupled branch i2v:527-532, unet:1263-1279
 *      (second call with accumulate = 1, acc_scale = ip scale)
 * V is passed TRANSPOSED: vt[b][h*d + i][key] (see I2V_STORE_VT), row stride vt_row_stride >= lk rounded
 * up to 8; entries past lk may hold garbage (they are masked in-kernel).  head_dim % 8 == 0, <= 160.
 * One (batch, head) slice of K ((lk + 128) * k_row_stride elements) and of V^T (head_dim * vt_row_stride) must
 * span less than 1 GiB (the kernel addresses them with 32-bit buffer offsets); larger inputs are rejected with
 * I2V_ERR_INVALID_ARG.  Numerics: scale * log2(e) is folded into the fp16 Q fragments (one extra fp16
 * rounding of Q), softmax in base 2 with fp32 accumulation, P rounded to fp16 for the PV product.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_attn_params {
  const void* q;  int64_t q_row_stride, q_batch_stride;
  const void* k;  int64_t k_row_stride, k_batch_stride;
  const void* vt; int64_t vt_row_stride, vt_batch_stride;
  void* o;        int64_t o_row_stride, o_batch_stride;
  int32_t batch_q, kv_group, heads, head_dim, lq, lk;
  float scale;
  int32_t accumulate;
  float acc_scale;
  float* lse;     /* optional (ABI 6): fp32 [batch_q][heads][lq], the log2-sum-exp of the scaled logits of every query row,
                     written by the forward pass itself (what i2v_attention_lse_f32 recomputes; the training step keeps it
                     for i2v_attention_bwd_f16).  NULL: not written.  Not combined with accumulate.                       */
} i2v_attn_params;

int i2v_attention_f16(const i2v_attn_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Temporal (motion-module) self-attention: sequence = frames (<= 32) of one pixel; tokens are in
 * (b, pixel, frame) order so each pixel's q/k rows are `frames` consecutive rows; vt is
 * vt[pixel][h*d + i][frame] with row stride vt_ld (>= frames rounded up to 8).
 * Replaces SDPA inside diffusers TransformerTemporalModel (unet:323-326,520-523,688-691).
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_tattn_params {
  const void* q; int64_t q_row_stride;
  const void* k; int64_t k_row_stride;
  const void* vt; int32_t vt_ld;
  void* o; int64_t o_row_stride;
  int32_t n_pixels, frames, heads, head_dim;
  float scale;
} i2v_tattn_params;

int i2v_temporal_attention_f16(const i2v_tattn_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The whole attention sub-block of a motion module's temporal transformer block in one launch:
 *     n = LayerNorm(x) * gamma + beta + pe[row % frames];  q, k, v = n Wq^T, n Wk^T, n Wv^T (no bias);
 *     out[rows of pixel][head h] = softmax(q_h k_h^T * scale) v_h over the `frames` rows of each pixel
 * for rows in (batch, pixel, frame) order -- `norm1 -> attn1` / `norm2 -> attn2` of the BasicTransformerBlock inside
 * diffusers TransformerTemporalModel up to (not including) to_out (unet:232-244, 413-425, 607-619), which
 * the caller applies with i2v_gemm_f16 (+ residual).  LayerNorm output, q, k, v and P are rounded to fp16 where the
 * un-fused kernels store them; statistics, logits, softmax and accumulation in fp32.
 * gamma: fp32 [channels]; shift: fp32 [frames][channels] = beta[c] + pe[frame][c] (the LayerNorm shift and the sinusoidal
 * table of SURVEY A10 pre-added: both are constants of the module).
 * w_qkv: per head its rows of Wq, Wk, Wv, each zero-padded to P = pad16(head_dim) rows, stored in MFMA-fragment order:
 * fp16 [heads][3][channels / 32][P / 16][64][8] with element [h][part][s][t][l][j] = W_part[h * head_dim + 16 t + (l & 15)]
 * [32 s + 8 (l >> 4) + j]  (i2v_motion_attn_pack_rows(heads, head_dim) * channels elements in all).
 * Implemented for the SD-1.5 64^2 level: i2v_motion_attn_supported(...) != 0 (channels 320, 8 heads of 40, 16 frames,
 * rows a multiple of 128); any other shape returns I2V_ERR_INVALID_ARG and callers use the un-fused kernels.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_motion_attn_params {
  const void* x; int64_t ldx;           /* fp16 [rows, channels] */
  const void* gamma;                    /* fp32 [channels] */
  const void* shift; int64_t ld_shift;  /* fp32 [frames, channels] */
  const void* w_qkv;
  void* out; int64_t ldo;               /* fp16 [rows, channels] */
  int64_t rows;
  int32_t channels, heads, head_dim, frames;
  float eps, scale;
  /* (ABI 8) optional, both or neither: the sub-block's out-projection and residual in the same launch --
   * out = x + o to_out[0].weight^T + to_out[0].bias (what the caller otherwise does with i2v_gemm_f16 + residual).
   * w_o: to_out[0].weight [channels, channels] in the fragment order of i2v_cross_attn_fused_f16's w_q (rows grouped per head-sized
   * slice); b_o: fp32 [channels].  out may be x. */
  const void* w_o; const void* b_o;
} i2v_motion_attn_params;

int32_t i2v_motion_attn_supported(int64_t rows, int32_t channels, int32_t heads, int32_t head_dim, int32_t frames);
int32_t i2v_motion_attn_pack_rows(int32_t heads, int32_t head_dim);
int i2v_motion_attn_f16(const i2v_motion_attn_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * `norm2 -> attn2` of the spatial transformer block (i2v:510-533) up to (not including) to_out, for a context that fits
 * the registers (the 77 CLIP tokens), in one launch: n = LayerNorm(x) * gamma + beta; q = n Wq^T;
 *     out[row][head h] = softmax(q_h K_h^T * scale) V_h   with K = ctx_k rows, V^T = ctx_vt rows of the row's context
 * (rows [i * rows_per_ctx, (i + 1) * rows_per_ctx) attend to context i: the frames of one sample share its prompt).
 * Replaces native_layer_norm + to_q + SDPA(q, to_k(ctx), to_v(ctx)); the context projections are the caller's (they do
 * not depend on the step).  gamma, beta: fp32 [channels]; w_q: to_q's rows per head, zero-padded to pad16(head_dim), in the
 * fragment order of i2v_motion_attn_f16's w_qkv with one part ([heads][channels / 32][P / 16][64][8],
 * i2v_cross_attn_fused_pack_rows(heads, head_dim) * channels elements).
 * ctx_frag: the context's projected keys and values as the MFMA operand fragments the kernel keeps in registers, packed once
 * per prompt: fp16 [n_ctx][heads][30][64][4] (i2v_cross_attn_fused_ctx_elems(n_ctx, heads, head_dim) elements) with
 *   [c][h][3 kt + t][16 g + r][j]      = K[c][key = 16 kt + r][h * head_dim + 16 t + 4 g + j]       (kt < 5, t < 3)
 *   [c][h][15 + 5 t + kt][16 g + r][j] = V[c][key = 16 kt + 4 g + j][h * head_dim + 16 t + r]
 * and zero wherever key >= ctx_len or the channel offset >= head_dim.
 * Implemented where i2v_cross_attn_fused_supported(...) != 0: channels 320, 8 heads of 40, ctx_len <= 80, rows and
 * rows_per_ctx multiples of 128 (the SD-1.5 64^2 level).
 * ip_frag (optional, NULL = none): the IP-Adapter's image tokens of the same contexts (unet:1263-1279) packed the
 * same way from to_k_ip / to_v_ip (ip_len <= 16 tokens, i.e. key tile 0 of the layout): out += ip_scale * softmax(q K_ip^T * scale) V_ip.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_cross_attn_fused_params {
  const void* x; int64_t ldx;            /* fp16 [rows, channels] */
  const void* gamma; const void* beta;   /* fp32 [channels] */
  const void* w_q;
  const void* ctx_frag;
  void* out; int64_t ldo;                /* fp16 [rows, channels] */
  int64_t rows, rows_per_ctx;
  int32_t channels, heads, head_dim, ctx_len;
  float eps, scale;
  const void* ip_frag; int32_t ip_len; float ip_scale;
  /* (ABI 8) optional, both or neither: out = x + o to_out[0].weight^T + to_out[0].bias in the same launch (as i2v_motion_attn_params) */
  const void* w_o; const void* b_o;
} i2v_cross_attn_fused_params;

int32_t i2v_cross_attn_fused_supported(int64_t rows, int32_t channels, int32_t heads, int32_t head_dim, int32_t ctx_len,
                                       int64_t rows_per_ctx);
int32_t i2v_cross_attn_fused_pack_rows(int32_t heads, int32_t head_dim);
int64_t i2v_cross_attn_fused_ctx_elems(int32_t n_ctx, int32_t heads, int32_t head_dim);
int i2v_cross_attn_fused_f16(const i2v_cross_attn_fused_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * (ABI 8) LayerNorm 1 and the projections in front of the spatial block's self- / cross-frame attention in one launch
 * (i2v:444-445 `norm1`, 468-473 `attn1` to_q / to_k / to_v, 483-492 the adapter's to_q):
 *     n = LayerNorm(x) gamma + beta;   qk[r, :] = n[r] W_qk^T  (n_qk = 2 C: [q | k], 3 C: [q | k | q_adapter], or C: [k] alone);
 *     vt[r / rows_per_image][c][r % rows_per_image] = (n[r] W_v^T)[c]     (the V^T operand of i2v_attention_f16)
 * replaces native_layer_norm + three / four aten addmm (+ the transpose SDPA does internally).  LayerNorm output rounded to fp16
 * where the un-fused kernels store it; statistics and accumulation in fp32.  gamma, beta: fp32 [channels].
 * w: the rows of [W_qk ; W_v] ((n_qk + C) x C, diffusers Linear layout) per 16-row tile in MFMA-fragment order
 *    [(n_qk + C) / 16][C / 32][64][8]: element [T][s][l][j] = W[16 T + (l & 15)][32 s + 8 (l >> 4) + j].
 * Implemented for the SD-1.5 64^2 level (i2v_ln_qkv_supported: channels 320, rows and rows_per_image multiples of 128; 0 also when
 * the current device refuses the kernel's 160 KB of LDS).
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_ln_qkv_params {
  const void* x; int64_t ldx;            /* fp16 [rows, channels] */
  const void* gamma; const void* beta;   /* fp32 [channels] */
  const void* w;
  void* qk; int64_t ld_qk;               /* fp16 [rows, n_qk] */
  void* vt; int64_t vt_batch_stride, vt_row_stride;   /* fp16 [rows / rows_per_image][channels][>= rows_per_image] */
  int64_t rows, rows_per_image;
  int32_t channels, n_qk;
  float eps;
  /* > 0: image i's rows_per_image rows start at x + i * x_image_stride elements instead of following image i - 1 -- the frame-0
     rows of every clip read in place for the adapter's K0 | V0^T (n_qk = C: [k] only; i2v:484-492, no gathered copy) */
  int64_t x_image_stride;
} i2v_ln_qkv_params;

int32_t i2v_ln_qkv_supported(int64_t rows, int32_t channels, int32_t n_qk, int64_t rows_per_image);
int i2v_ln_qkv_f16(const i2v_ln_qkv_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The GEGLU feed-forward of a transformer block in one launch (i2v:539-561 `norm3 -> ff -> + residual`; the temporal block's
 * FeedForward):  out = x + W2 (value o gelu(gate)) + b2  with  [value | gate] = (LayerNorm(x) gamma + beta) W1^T + b1.
 * The inner activation (rows x inner) never leaves the CU.  LayerNorm output and the inner activation are rounded to fp16 where
 * the un-fused kernels store them; statistics, GELU (common.h gelu_erf) and accumulation in fp32.  out may alias x.
 * gamma, beta: fp32 [channels]; b2: fp32 [channels].  Packed operands (fp16 unless noted), with the inner dimension in chunks of 128:
 *   w1 [inner / 128][8][2][channels / 32][64][8]: element [ch][w][u][s][l][j] =
 *      W1[(m & 1) * inner + 128 ch + 16 w + 8 u + (m >> 1)][32 s + 8 (l >> 4) + j], m = l & 15  (diffusers GEGLU.proj: rows
 *      [0, inner) are the values, [inner, 2 inner) the gates: a 16-row tile holds 8 (value, gate) pairs);
 *   b1 fp32 [inner / 128][8][2][16]: the same rows' biases;
 *   w2 [8][inner / 128][4][3][64][8]: element [w][ch][ks][t][l][j] = W2[n = 40 w + 16 t + (l & 15)][128 ch + 32 ks + 8 (l >> 4) + j],
 *      zero where 16 t + (l & 15) >= 40.
 * Implemented for the SD-1.5 64^2 level (i2v_ff_fused_supported: channels 320, inner 1280, rows a multiple of 128; 0 also when
 * the current device refuses the kernel's 160 KB of LDS -- callers then take the un-fused GEMM pair).
 *
 * Optional tail (ABI 8; w3 != NULL): the Linear that follows the block in the same launch --
 *     out[perm(r)] = res2[perm(r)] + y[r] W3^T + b3,   y = x + FF(LayerNorm(x)) rounded to fp16 as the un-fused kernel stores it
 * i.e. the spatial transformer's proj_out with its residual (i2v:298-314; perm = identity) or the motion module's proj_out
 * (perm_frames = F > 0, a power of two: rows arrive in (batch, pixel, frame) order, and out / res2 are addressed in
 * (batch, frame, pixel) order with perm_hw pixels per image -- I2V_STORE_ROWPERM of i2v_gemm_f16).  w3: W3 [channels, channels]
 * per 40-row slice in fragment order, the layout of i2v_cross_attn_fused_params.w_q ([8][channels / 32][3][64][8]); b3 fp32
 * [channels]; res2 fp16 rows of ld_res2 elements.  With the tail, out must not alias x when perm_frames > 0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_ff_fused_params {
  const void* x; int64_t ldx;            /* fp16 [rows, channels] */
  const void* gamma; const void* beta;   /* fp32 [channels] */
  const void* w1; const void* b1;
  const void* w2; const void* b2;
  void* out; int64_t ldo;                /* fp16 [rows, channels] */
  int64_t rows;
  int32_t channels, inner;
  float eps;
  /* tail (all zero: none) */
  const void* w3; const void* b3;
  const void* res2; int64_t ld_res2;
  int32_t perm_frames, perm_hw;
  /* (ABI 9) the precise residual stream through the tail (see i2v_gemm_params.residual_lo / c_lo): res2_lo [rows of ld_res2] is
     added with res2, out_lo [rows of ldo] receives the low half of the result.  Tail only; both or neither. */
  const void* res2_lo;
  void* out_lo;
} i2v_ff_fused_params;

int32_t i2v_ff_fused_supported(int64_t rows, int32_t channels, int32_t inner);
/* ... with the tail: rows in (batch, pixel, frame) order of `perm_frames` frames and `perm_hw` pixels per image (0, 0: no permutation) */
int32_t i2v_ff_fused_tail_supported(int64_t rows, int32_t channels, int32_t inner, int32_t perm_frames, int32_t perm_hw);
int i2v_ff_fused_f16(const i2v_ff_fused_params* p, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) on token-major fp16: statistics in fp32.
 *   stat group = (frames_per_stat consecutive images) x (all pixels) x (C / groups channels)
 *   frames_per_stat = 1: ResnetBlock2D norm1/norm2, Transformer2D norm, conv_norm_out
 *                        (unet:203-214, i2v:218, unet:870-873)
 *   frames_per_stat = F: TransformerTemporalModel.norm, statistics over (C/G, F, H, W)
 * x may be two sources concatenated along channels (c1 from x, c2 from x2: skip concat, unet:478).
 * out_perm = 1 writes rows in (b, pixel, frame) order (motion-module entry).
 * workspace: i2v_groupnorm_workspace_bytes(...) bytes of scratch, fp32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_gn_params {
  const void* x;  int32_t c1;
  const void* x2; int32_t c2;
  const void* gamma; const void* beta; /* fp16 [c1 + c2] */
  void* y;                              /* fp16 [n_img, hw, c1 + c2] */
  int32_t n_img, hw, groups, frames_per_stat;
  float eps;
  int32_t silu;
  int32_t out_perm; int32_t frames; /* out_perm: images are (b, f); output row = (b*hw + p)*frames + f */
  void* workspace;
  /* (ABI 8) optional: the statistics as per-group partials written by the producing convolution (i2v_gemm_params.gn_partial) --
     fp32 [n_img][hw / gpartial_rows][groups][2]; the statistics pass over x is skipped.  frames_per_stat 1, no out_perm, no x2. */
  const void* gpartial_in; int32_t gpartial_rows;
} i2v_gn_params;

int64_t i2v_groupnorm_workspace_bytes(int32_t n_img, int32_t hw, int32_t channels);
int i2v_groupnorm_f16(const i2v_gn_params* p, i2v_stream_t stream);
/* GroupNorm (no activation) folded into the Linear / 1x1 conv that consumes it (the entry of every spatial transformer
 * and motion module: norm -> proj_in, i2v:218-226): the statistics of p->x are computed as above, then instead of
 * writing the normalised tensor, per statistics group s (= image, or clip when frames_per_stat > 1)
 *     w_out[s][n][c]  = fp16(w[n][c] * a[s][c])              a = gamma * rstd
 *     bias_out[s][n]  = fp16(bias[n] + sum_c w[n][c] b[s][c])  b = beta - mean * a
 * so that proj_in(GroupNorm(x)) = x w_out[s]^T + bias_out[s] (i2v_gemm_params.w_batch_stride / rows_per_w, bias through
 * rowvec / rows_per_vec): the normalised activations are never written to / read back from HBM.  p->y, silu, out_perm
 * are ignored.  w [n_out, ldw >= C] fp16, w_out [n_img / frames_per_stat, n_out, C], bias_out [.., n_out] fp16. */
int i2v_groupnorm_fold_f16(const i2v_gn_params* p, const void* w, int64_t ldw, const void* bias, int32_t n_out,
                           void* w_out, void* bias_out, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis (fp32 statistics), optional additive positional embedding
 * y[r, :] = LN(x[r, :]) * gamma + beta (+ pe[r % pe_period, :])
 * replaces native_layer_norm at i2v:445,514,539 and, with pe, LN + SinusoidalPositionalEmbedding of the
 * motion-module block.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_ln_params {
  const void* x; int64_t ldx;
  const void* gamma; const void* beta; /* fp16 [C] */
  const void* pe; int64_t ld_pe; int32_t pe_period; /* fp16 [pe_period, C] or NULL */
  void* y; int64_t ldy;
  int32_t rows, C;
  float eps;
  /* x_rows_per_batch > 0: the input is a batch of row blocks, output row r reads x + (r / x_rows_per_batch) *
   * x_batch_stride + (r % x_rows_per_batch) * ldx (elements).  The cross-frame attention normalises only the frame-0
   * rows of each clip (i2v:484: `norm_hidden_states[0:batch:num_frames]`): rows_per_batch = tokens per frame, batch
   * stride = one clip, read in place instead of through a gathered copy.  0: one dense matrix (the output is always dense). */
  int32_t x_rows_per_batch; int64_t x_batch_stride;
} i2v_ln_params;

int i2v_layernorm_f16(const i2v_ln_params* p, i2v_stream_t stream);

/* y[r, :cols] = softmax(scale * x[r, :cols]) row-wise, fp16 in / out, fp32 statistics (in place allowed).  The VAE
 * mid-block attention (diffusers Attention with ONE head of dim 512, AutoencoderKL decode / encode, pipe:305,627)
 * exceeds the flash kernel's head_dim limit: its QK^T and PV products run as i2v_gemm_f16 calls around this kernel. */
int i2v_softmax_rows_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, float scale,
                         i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layout edges, embeddings and the sampler step
 * ------------------------------------------------------------------------------------------------ */
/* NCHW (fp32 if src_is_f32 else fp16) [n, c, h*w] -> token-major fp16 [n, h*w, c_pad], channels >= c zeroed
 * (sample.reshape + conv_in input, unet:1358). */
int i2v_nchw_to_tokens(const void* src, int32_t src_is_f32, void* dst, int32_t n, int32_t c, int32_t hw,
                       int32_t c_pad, i2v_stream_t stream);
/* token-major (fp32 if src_is_f32 else fp16) [n, hw, ld] (first c channels) -> NCHW (fp32 if dst_is_f32 else fp16)
 * (unet:1446). */
int i2v_tokens_to_nchw(const void* src, int32_t src_is_f32, int64_t ld, void* dst, int32_t dst_is_f32, int32_t n,
                       int32_t c, int32_t hw, i2v_stream_t stream);
/* Timesteps(dim, flip_sin_to_cos=True, shift 0): out[b, :] = [cos(t*w) | sin(t*w)] fp16 (unet:763,1336).
 * t is fp32 [n]; if t_index != NULL, t is a table of t_rows entries and the single value
 * t[clamp(*t_index, 0, t_rows - 1)] is used for every row (graph replay). */
int i2v_timestep_embedding(const float* t, const int32_t* t_index, int32_t t_rows, void* out, int32_t n,
                           int32_t dim, i2v_stream_t stream);
/* out[0, :] = table[clamp(*row_index, 0, rows - 1), :] (fp16, cols % 8 == 0): the pipeline computes
 * time_emb_proj(silu(time_embedding(time_proj(t)))) of every resnet (unet:1336-1343) for ALL timesteps of the
 * schedule once per sample; a replayed step picks its row by the device-side step counter (pipe:666 `for i, t in ...`). */
int i2v_select_row_f16(const void* table, int64_t ld, int32_t rows, const int32_t* row_index, void* out, int32_t cols,
                       i2v_stream_t stream);
/* y = silu(x), fp16, n elements (nonlinearity(temb), ResnetBlock2D). */
int i2v_silu_f16(const void* x, void* y, int64_t n, i2v_stream_t stream);
/* y[r, :] = x[r / repeat, :]  (repeat_interleave of temb / context rows, unet:1344,1355). */
int i2v_repeat_rows_f16(const void* x, void* y, int64_t rows_in, int64_t cols, int32_t repeat,
                        i2v_stream_t stream);
/* dst[b, r, :cols] = src[b, r, :cols] for b < batches, r < rows: fp16 strided 3-D copy (strides in elements).
 * Used to gather the frame-0 tokens of every clip (i2v:484) and to split text / image context tokens. */
int i2v_copy3d_f16(const void* src, int64_t src_batch_stride, int64_t ld_src, void* dst, int64_t dst_batch_stride,
                   int64_t ld_dst, int64_t batches, int64_t rows, int64_t cols, i2v_stream_t stream);

/* (ABI 9) `ctx_frag` of i2v_cross_attn_fused_f16 from the projected context as the other kernels take it: k fp16 [n_ctx * ctx_len, ldk]
 * (head h in columns h * head_dim ..), vt fp16 V^T[n_ctx][heads * head_dim][>= ctx_len] (row / batch strides in elements) ->
 * out fp16 [n_ctx][heads][2 * 5 * ceil(head_dim / 16)][64][4] (i2v_pack_ctx_fragments_elems() elements), zero beyond ctx_len (<= 80)
 * and head_dim.  Serves i2v:527-532 / unet:1263-1279 (to_k / to_v of the prompt and of the IP-Adapter's image tokens); until ABI 9
 * the host mirror assembled it with torch index ops, which kept a whole-model forward from being library launches only. */
int64_t i2v_pack_ctx_fragments_elems(int32_t n_ctx, int32_t heads, int32_t head_dim);
int i2v_pack_ctx_fragments_f16(const void* k, int64_t ldk, const void* vt, int64_t vt_row_stride, int64_t vt_batch_stride, void* out,
                               int32_t n_ctx, int32_t heads, int32_t head_dim, int32_t ctx_len, i2v_stream_t stream);

/* One DDIM step around the UNet call, pipe:666-691, split in the two halves that bracket it.
 *   prep : latents[:, 0] = cond (pipe:669); model_in = tokens(cat([latents] * cfg_copies)) fp16, channels
 *          padded to c_pad (pipe:672-673; scale_model_input is the identity for DDIM).
 *   step : eps = u + g (c - u) (pipe:686-688); x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);
 *          x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps (pipe:691); then *step_index advances by one
 *          and wraps to 0 after n_steps (the table row read is clamped to [0, n_steps - 1]).
 * latents fp32 [b, f, c, hw]; cond fp32 [b, c, hw]; noise_pred tokens [cfg_copies*b*f, hw, ld_np], fp32 if np_is_f32
 * (the conv_out GEMM with c_is_f32) else fp16;
 * coef fp32 [n_steps][4] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}; step_index device int32. */
int i2v_ddim_prep(float* latents, const float* cond, void* model_in, int32_t b, int32_t f, int32_t c,
                  int32_t hw, int32_t c_pad, int32_t cfg_copies, i2v_stream_t stream);
int i2v_ddim_cfg_step(float* latents, const void* noise_pred, int32_t np_is_f32, int64_t ld_np, const float* coef, int32_t n_steps,
                      int32_t* step_index, float guidance_scale, int32_t b, int32_t f, int32_t c, int32_t hw,
                      int32_t cfg_copies, i2v_stream_t stream);

/* First-frame-similarity prior and the initial add_noise of the sampling loop, pipe:647-656:
 *   prior   = mask * GaussianBlur3x3(cond) + (1 - mask) * cond, mask = (mask_uniform < strength), per frame (pipe:648-654)
 *   latents = sqrt_alpha * prior + sqrt_one_minus_alpha * noise                          (scheduler.add_noise, pipe:656)
 * cond fp32 [b, c, h, w]; mask_uniform, noise, latents fp32 [b, f, c, h, w].  The blur is torchvision's
 * GaussianBlur(kernel_size=3) for one sigma (pipe:112): separable taps (k_edge, k_center, k_edge), reflect padding. */
int i2v_first_frame_prior_f32(const float* cond, const float* mask_uniform, const float* noise, float* latents,
                              int32_t b, int32_t f, int32_t c, int32_t h, int32_t w, float k_center, float k_edge,
                              float strength, float sqrt_alpha, float sqrt_one_minus_alpha, i2v_stream_t stream);

/* DiagonalGaussianDistribution.sample() of the VAE encoder (vae.encode(image).latent_dist.sample(), pipe:627):
 * moments fp32 [n, 2 c, hw] = (mean | logvar) along the channel axis; out = mean + exp(0.5 clamp(logvar, -30, 20)) eps,
 * eps / out fp32 [n, c, hw]. */
int i2v_gaussian_sample_f32(const float* moments, const float* eps, float* out, int32_t n, int32_t c, int32_t hw,
                            i2v_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Backward kernels of the adapter training step.  The reference trains only
 * i2v_adapter.to_q / to_out (unet:979-1026) through the whole frozen UNet with torch autograd
 * (src/train_image_to_video.py:839-884: forward with enable_cross_frame_attn, MSE without the first frame :848-856,
 * backward, clip, step); these entry points replace the aten backward kernels that autograd dispatches for one
 * I2VAdapterTransformerBlock (i2v:420-565): SDPA backward, native_layer_norm_backward, the GEGLU chunk / gelu backward,
 * bias-gradient sums, mse_loss backward.  Linear dgrad / wgrad are i2v_gemm_f16 over transposed weights / activations.
 * Gradients are fp16 (scaled by the caller's loss scale) except the fp32 statistics and the fp32 bias gradient.
 * ------------------------------------------------------------------------------------------------ */
/* lse[bq][h][l] = log2 sum_j exp2(scale * log2(e) * Q[bq, l, h, :] . K[bq / kv_group, j, h, :]), fp32
 * [batch_q, heads, lq]: the softmax statistic of the forward attention described by p (p->vt, p->o, accumulate are
 * ignored).  The backward recomputes P from it instead of storing the lq x lk scores. */
int i2v_attention_lse_f32(const i2v_attn_params* p, float* lse, i2v_stream_t stream);

/* Flash-attention backward (aten _scaled_dot_product_*_attention_backward behind i2v:468-473, 483-492, 527-532):
 *   P = exp2(scale log2e Q K^T - lse),  dP = dO V^T,  dS = P o (dP - delta),  delta[bq][h][l] = sum_i dO o O
 *   dQ = scale dS K          (one sweep over the keys per 16-query wave)
 *   dK = scale dS^T Q,  dV = P^T dO  summed over the kv_group batch entries that share K / V (the F frames of a clip for
 *                        the cross-frame adapter attention: dK0 / dV0), one sweep over the queries per 16-key wave.
 * q, k, v, dout, dq, dk, dv are token-major [batch][token][heads * head_dim] views (row / batch strides in elements);
 * qt, kt, doutt are channel-major copies [batch][heads * head_dim][token] (i2v_transpose_f16; rows zero-filled up to the
 * next multiple of 8 tokens).  dk = NULL skips the dK / dV sweep (frozen context K / V of the text cross-attention); qt,
 * doutt, dv are then unused.  Any lq, lk (short sequences -- the <= 32 frames of one pixel in the motion modules, batch =
 * pixels -- run with most of a 32-row block masked).  No atomics: gradients are run-to-run identical. */
typedef struct i2v_attn_bwd_params {
  const void* q;     int64_t q_row_stride, q_batch_stride;
  const void* qt;    int64_t qt_row_stride, qt_batch_stride;
  const void* k;     int64_t k_row_stride, k_batch_stride;
  const void* kt;    int64_t kt_row_stride, kt_batch_stride;
  const void* v;     int64_t v_row_stride, v_batch_stride;
  const void* dout;  int64_t do_row_stride, do_batch_stride;
  const void* doutt; int64_t dot_row_stride, dot_batch_stride;
  const float* lse;   /* [batch_q, heads, lq] from i2v_attention_lse_f32 */
  const float* delta; /* [batch_q, heads, lq] from i2v_rowdot_heads_f32(dO, O) */
  void* dq;          int64_t dq_row_stride, dq_batch_stride;
  void* dk;          int64_t dk_row_stride, dk_batch_stride;
  void* dv;          int64_t dv_row_stride, dv_batch_stride;
  int32_t batch_q, kv_group, heads, head_dim, lq, lk;
  float scale;
  int32_t kv_partitions; /* (ABI 6) > 1: the kv_group query batches that share one K / V (the frames of a clip in the cross-frame
                            attention) are dealt to kv_partitions workgroups per key block instead of one -- that form has
                            batch_q / kv_group times fewer workgroups than the self-attention and cannot fill the chip alone;
                            kv_group % kv_partitions == 0.  The partial dK / dV go to dkv_partial and a second kernel sums them
                            into dk / dv (fixed order: run-to-run identical).  0 / 1: off.                                       */
  float* dkv_partial;    /* scratch, fp32 [2][kv_partitions][batch_q / kv_group * lk][heads * head_dim]                       */
} i2v_attn_bwd_params;
int i2v_attention_bwd_f16(const i2v_attn_bwd_params* p, i2v_stream_t stream);

/* dst[b][c][r] = src[b][r][c] for r < rows, c < cols (fp16; strides in elements); dst columns [rows, rows rounded up to 8)
 * are zero-filled (ld_dst must cover them).  Channel-major copies of token-major activations: the K^T / Q^T / dO^T
 * operands of i2v_attention_bwd_f16 and the operands of a weight gradient dW = dY^T X as i2v_gemm_f16(a = dY^T, w = X^T). */
int i2v_transpose_f16(const void* src, int64_t src_batch_stride, int64_t ld_src, void* dst, int64_t dst_batch_stride,
                      int64_t ld_dst, int32_t batches, int32_t rows, int32_t cols, i2v_stream_t stream);
/* out[(b * heads + h) * rows_per_batch + l] = sum_i a[b * rows_per_batch + l][h * head_dim + i] * b[..][..], fp32:
 * delta = rowsum(dO o O) per head. */
int i2v_rowdot_heads_f32(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows,
                         int32_t rows_per_batch, int32_t heads, int32_t head_dim, i2v_stream_t stream);
/* LayerNorm backward, input gradient only (the norms are frozen; native_layer_norm_backward behind i2v:445, 514, 539):
 * dx[r] = rstd (gy - mean(gy) - xh mean(gy o xh)) + add[r], gy = dn o gamma, xh = (x - mean) rstd; add (the gradient
 * arriving over the residual path) may be NULL. */
int i2v_layernorm_bwd_f16(const void* x, int64_t ldx, const void* dn, int64_t lddn, const void* gamma, const void* add,
                          int64_t ldadd, void* dx, int64_t lddx, int32_t rows, int32_t C, float eps, i2v_stream_t stream);
/* GEGLU backward (i2v:554): h [rows, 2 inner] is the pre-activation in the interleaved (value_i, gate_i) column order of
 * I2V_EPI_GEGLU, dy [rows, inner] -> dh[.., 2i] = dy gelu(gate), dh[.., 2i + 1] = dy value gelu'(gate). */
int i2v_geglu_bwd_f16(const void* h, int64_t ldh, const void* dy, int64_t lddy, void* dh, int64_t lddh, int64_t rows,
                      int32_t inner, i2v_stream_t stream);
/* GEGLU forward from a stored pre-activation (ABI 6): y[r][i] = h[r][2 i] gelu_erf(h[r][2 i + 1]), h [rows][>= 2 inner] with
 * (value, gate) interleaved as the I2V_EPI_GEGLU GEMM packs its weights, y [rows][>= inner]; inner % 8 == 0.  The training
 * forward keeps h for i2v_geglu_bwd_f16 and derives y from it (diffusers GEGLU.forward, i2v:554). */
int i2v_geglu_f16(const void* h, int64_t ldh, void* y, int64_t ldy, int64_t rows, int32_t inner, i2v_stream_t stream);
/* out[c] += sum_r x[r][c], fp32 (the bias gradient of a Linear; the caller zeroes out before the first call). */
int i2v_colsum_f32(const void* x, int64_t ldx, float* out, int64_t rows, int32_t cols, i2v_stream_t stream);
/* out[c] += sum_r a[r][c] b[r][c], fp32: the gain gradient of a LayerNorm / GroupNorm, d gamma = sum dy o xhat -- the
 * motion modules' norms train under `--update_motion_modules` (train_image_to_video.py:452, 669; unet:984-999). */
int i2v_colsum_prod_f32(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int32_t cols,
                        i2v_stream_t stream);
/* Both sums with a result that does not depend on block scheduling (the two above add their 256-row block sums with fp32
 * atomics): out[c] += sum_r a[r][c] (b == NULL) or sum_r a[r][c] b[r][c]; the block sums go to `workspace`
 * (i2v_colsum_workspace_bytes(rows, cols) bytes) and are added in block order.  What the training step uses: every
 * data-parallel rank and every rerun gets the same bias / gain gradients bit for bit. */
int64_t i2v_colsum_workspace_bytes(int64_t rows, int32_t cols);
int i2v_colsum_det_f32(const void* a, int64_t lda, const void* b, int64_t ldb, float* out, int64_t rows, int32_t cols,
                       void* workspace, i2v_stream_t stream);
/* Seed gradient of the training loss (train_image_to_video.py:848-856: MSE summed over every frame but the first of each
 * clip, divided by the number of unmasked elements): grad[img][l][c] = coef (y - target) for img % frames != 0, else 0;
 * coef = 2 * loss_scale / count is the caller's.  y, target, grad fp16 [n_img, tokens, channels]. */
int i2v_masked_mse_grad_f16(const void* y, const void* target, void* grad, int64_t n_img, int32_t tokens, int32_t channels,
                            int32_t frames, float coef, i2v_stream_t stream);
/* The same seed from an fp32 prediction and an fp32 target -- F.mse_loss(model_pred.float(), target.float()),
 * train_image_to_video.py:848 -- with grad in fp16 and rowsq[img * tokens + l] = sum_c (y - target)^2 (0 on masked rows): the
 * loss is sum(rowsq) / count, summed by the caller in a fixed order. */
int i2v_masked_mse_grad_f32(const float* y, const float* target, void* grad, float* rowsq, int64_t n_img, int32_t tokens,
                            int32_t channels, int32_t frames, float coef, i2v_stream_t stream);
/* GroupNorm (+SiLU) backward, input gradient only (native_group_norm_backward + silu_backward behind ResnetBlock2D norm1 /
 * norm2, Transformer2D norm, the motion modules' clip-wide norm and conv_norm_out; the norms are frozen): p describes the
 * FORWARD call (x [, x2], gamma, beta, n_img, hw, groups, frames_per_stat, eps, silu; y and out_perm unused), dy is the
 * gradient of its output [n_img, hw, c1 + c2]; dx [n_img, hw, c1] (and dx2 [.., c2] for a channel-concatenated input).
 * The forward statistics are recomputed (the inference forward keeps none).  workspace: fp32,
 * i2v_groupnorm_bwd_workspace_bytes(...) bytes. */
int64_t i2v_groupnorm_bwd_workspace_bytes(int32_t n_img, int32_t hw, int32_t channels);
int i2v_groupnorm_bwd_f16(const i2v_gn_params* p, const void* dy, void* dx, void* dx2, i2v_stream_t stream);
/* out = a + b over n fp16 elements (n % 8 == 0): gradients meeting at a skip connection (unet:478) or a residual branch. */
int i2v_add_f16(const void* a, const void* b, void* out, int64_t n, i2v_stream_t stream);
/* row permutation (batch, frame, pixel) <-> (batch, pixel, frame) of [batches * frames * hw, channels] fp16 (the motion
 * modules run in pixel-major row order; to_pixel_major = 1: dst row (b hw + p) F + f = src row (b F + f) hw + p). */
int i2v_permute_rows_f16(const void* src, void* dst, int64_t batches, int32_t frames, int32_t hw, int32_t channels,
                         int32_t to_pixel_major, i2v_stream_t stream);
/* dst [n, 2h, 2w, C]: dst[2y][2x] = src[y][x], zero elsewhere: the input gradient of Downsample2D's stride-2 convolution
 * (unet:250-259) is the stride-1 convolution of this tensor with the flipped, transposed weights. */
int i2v_zero_insert2x_f16(const void* src, void* dst, int64_t n_img, int32_t h, int32_t w, int32_t channels, i2v_stream_t stream);
/* dst [n, h, w, C] = sums of the 2 x 2 blocks of src [n, 2h, 2w, C]: input gradient of Upsample2D's nearest-2x (unet:431-432). */
int i2v_sum_pool2x_f16(const void* src, void* dst, int64_t n_img, int32_t h, int32_t w, int32_t channels, i2v_stream_t stream);
/* Optimiser step of the adapter parameters (train_image_to_video.py:876-882: clip_grad_norm_, AdamW.step) over ONE flat fp32
 * bucket (the same bucket the RCCL all-reduce sums): i2v_sumsq_f32 adds sum x^2 into *out (zeroed by the caller);
 * i2v_adamw_f32 is torch.optim.AdamW's update (decoupled weight decay, bias correction with `step` >= 1) on
 * g = grad * grad_coef * min(1, max_norm / (sqrt(*norm_sq) * grad_coef + 1e-6)): grad_coef folds 1 / loss_scale and the
 * data-parallel mean, the clip coefficient is read from device memory (no host round trip); norm_sq = NULL or
 * max_norm <= 0: no clipping. */
int i2v_sumsq_f32(const float* x, int64_t n, float* out, i2v_stream_t stream);
int i2v_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, float grad_coef, const float* norm_sq,
                  float max_norm, i2v_stream_t stream);

/* The same step as ONE call that is safe under fp16 loss scaling and identical on every data-parallel rank
 * (train_image_to_video.py:306-308 `--mixed_precision fp16`: accelerate's GradScaler skips a step whose gradients hold
 * inf / NaN; :876-882 clip + step): sum grad^2 as per-workgroup partials (`partials`, fp32 [n_partials <= 1024]) summed
 * in a fixed order into *norm_sq -- no atomics, so every rank derives the same clip coefficient from the same all-reduced
 * bucket --; a non-finite norm sets *found_inf = 1 and the update is a no-op (parameters, both moments and the
 * bias-correction step unchanged), otherwise *found_inf = 0, *applied_steps += 1 and AdamW runs with bias correction
 * 1 - beta^(*applied_steps).  The caller reads *found_inf when it wants to back its loss scale off; nothing here syncs. */
int i2v_adamw_guarded_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float grad_coef, float max_norm,
                          float* partials, int32_t n_partials, float* norm_sq, int32_t* applied_steps, int32_t* found_inf,
                          i2v_stream_t stream);

/* y = a y + b x over n fp32 values: gradient accumulation over micro-batches (`accelerator.accumulate`,
 * train_image_to_video.py:486, 785: a = 1, b = 1 / gradient_accumulation_steps) and the exponential moving average of the
 * trained weights (`--use_ema`, :673-677, 888-889: a = decay, b = 1 - decay). */
int i2v_axpby_f32(float* y, const float* x, float a, float b, int64_t n, i2v_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Model handle (ABI 8; whole-model forward since ABI 9): the layer of SURVEY 8(b) over the entry points above for a host that is
 * NOT this package's Python mirror -- it serves `self.unet = unet`, `self.unet(latent_model_input, t, ...)` and the hipGraph per
 * DDIM step (pipe:96, 676-683; unet:1289-1451).
 *
 * The handle owns (a) the model's configuration, (b) a registry of the caller's weight buffers by key, (c) the step's problem
 * (i2v_unet_plan), (d) a LAUNCH PLAN of one forward for that problem and (e) optionally one captured step.
 *
 * The launch plan (i2v_unet_set_plan) is the forward of unet:1289-1451 as data: the exact sequence of this header's entry points
 * one `UNetMotionCrossFrameAttnModel.forward` issues for the planned problem -- entry-point id + its parameter struct (or flat
 * arguments) per launch -- with every device pointer replaced by a relocation:
 *     weight  (key index, byte offset)   resolved through the registry (i2v_unet_set_weight), so `set_weight` decides what the
 *                                        forward reads.  Keys are the KERNEL-LAYOUT packs of the host mirror
 *                                        (`<module path>#<pack name>`: conv weights in contraction order, GEGLU rows interleaved,
 *                                        LayerNorm-folded projections, MFMA-fragment-ordered operands of the fused kernels ...),
 *                                        exported once, offline, with the plan (handle.py `record_forward_plan`,
 *                                        `export_weights`); no packing arithmetic happens at run time
 *     arena   (byte offset)              an activation / workspace inside ONE caller-owned arena of i2v_unet_activation_bytes()
 *                                        bytes (i2v_unet_set_workspace): the recorded allocation pattern of the forward
 *     io      (slot, byte offset)        one of the forward's arguments (I2V_IO_*)
 * i2v_unet_forward resolves them and issues the launches in C on the caller's stream: no Python, no torch, no allocation, no
 * synchronisation -- capturable between i2v_unet_capture_step and i2v_unet_end_capture together with the host's own DDIM update
 * (i2v_ddim_prep / i2v_ddim_cfg_step).  The plan is made by recording the host mirror once per (problem, switches) in the build
 * environment -- the sequencing logic of blocks.py / kernels.py is not duplicated in C++.
 * One handle per (device, stream); not thread-safe; no call synchronises the device except i2v_unet_destroy's release of the graph.
 * ------------------------------------------------------------------------------------------------ */
typedef struct i2v_unet i2v_unet;
typedef struct i2v_unet_config {
  int32_t in_channels, out_channels;                 /* 4, 4 (unet:701-702) */
  int32_t block_out_channels[4];                     /* 320, 640, 1280, 1280 (unet:708) */
  int32_t layers_per_block, num_attention_heads;     /* 2, 8 */
  int32_t cross_attention_dim, norm_num_groups;      /* 768, 32 */
  int32_t motion_max_seq_length, motion_num_attention_heads;   /* 32, 8 (unet:725-726) */
  int32_t use_motion_mid_block, ip_num_tokens;       /* 1; 0 = no IP-Adapter, else 4 (unet:1284-1287) */
} i2v_unet_config;
typedef struct i2v_unet_plan_t {
  int32_t batch, frames, height, width;              /* latent sizes: batch counts the CFG copies */
  int32_t ctx_len, has_ip;
} i2v_unet_plan_t;
#define I2V_DTYPE_F16 0
#define I2V_DTYPE_F32 1
/* the forward's arguments as relocation slots of a launch plan */
enum { I2V_IO_SAMPLE = 0, I2V_IO_TIMESTEPS = 1, I2V_IO_CONTEXT = 2, I2V_IO_IMAGE_EMBEDS = 3, I2V_IO_OUT = 4, I2V_IO_SLOTS = 5 };

int i2v_unet_create(const i2v_unet_config* cfg, i2v_unet** out);
int i2v_unet_destroy(i2v_unet* h);
/* registers (or replaces) the caller's buffer for a key; ndim <= 4.  The library never copies or frees it. */
int i2v_unet_set_weight(i2v_unet* h, const char* key, const void* ptr, int32_t dtype, int32_t ndim, const int64_t* shape);
/* the registered buffer of a key (0 and *ptr = NULL when absent); shape may be NULL */
int i2v_unet_get_weight(const i2v_unet* h, const char* key, const void** ptr, int32_t* dtype, int32_t* ndim, int64_t* shape);
int64_t i2v_unet_num_weights(const i2v_unet* h);
/* validates and records the step's problem: frames <= motion_max_seq_length (unet:725), positive sizes (latent sizes that are
 * not multiples of 8 take the forward_upsample_size path, unet:1304-1311: a matter of the recorded plan), ctx_len >= 1; a new
 * problem drops the launch plan and the captured step */
int i2v_unet_plan(i2v_unet* h, const i2v_unet_plan_t* plan);
/* (ABI 9) installs a launch plan (the blob handle.py `record_forward_plan` writes; copied).  Checked: magic / version, the ABI
 * version it was recorded against, sizeof of every parameter struct it carries, entry-point ids, relocation targets inside their
 * payloads, and that it was recorded for exactly the planned problem (i2v_unet_plan first).  Drops the captured step. */
int i2v_unet_set_plan(i2v_unet* h, const void* blob, int64_t bytes);
/* bytes of the ONE arena the installed plan's activations and workspaces live in (0 without a plan) */
int64_t i2v_unet_activation_bytes(const i2v_unet* h);
/* launches of the installed plan / distinct weight keys it names (0 without a plan); the i-th key (NULL out of range) */
int32_t i2v_unet_plan_launches(const i2v_unet* h);
int32_t i2v_unet_plan_num_keys(const i2v_unet* h);
const char* i2v_unet_plan_key(const i2v_unet* h, int32_t i);
/* the caller's arena: >= i2v_unet_activation_bytes() bytes, 256-byte aligned, alive as long as forwards / a captured step use it */
int i2v_unet_set_workspace(i2v_unet* h, void* arena, int64_t bytes);
/* (ABI 9) unet:1289-1451 for the planned problem: sample fp16 [batch, frames, in_channels, height, width], timesteps fp32 [batch],
 * context fp16 [batch, ctx_len, cross_attention_dim], image_embeds fp16 [batch, clip_dim] or NULL (needed iff the plan was
 * recorded with the IP-Adapter), out [batch, frames, out_channels, height, width] in the sample's dtype.  Asynchronous on `stream`.
 * I2V_ERR_INVALID_ARG: no plan / arena, a weight key of the plan that is not registered (named in i2v_last_error), a NULL
 * argument the plan reads; errors of the launches themselves are returned as they come. */
int i2v_unet_forward(i2v_unet* h, const void* sample, const void* timesteps, const void* context, const void* image_embeds,
                     void* out, i2v_stream_t stream);
/* (ABI 9) the general form: runs the installed plan with `io[slot]` as its arguments (n_io <= I2V_IO_SLOTS; the rest NULL).
 * i2v_unet_forward is i2v_unet_run with {sample, timesteps, context, image_embeds, out}.  Plans of OTHER launch sequences of the host
 * mirror use it with their own slot meaning (handle.py `record_plan`): the per-sample preparation (context K / V^T of the 16
 * cross-attention layers + the time-embedding table of the schedule; `record_prepare_plan`) and one whole DDIM step -- i2v_ddim_prep,
 * the UNet on the CFG batch as the pipeline routes it, i2v_ddim_cfg_step, pipe:666-697 -- (`record_step_plan`), whose per-sample
 * buffers are registered by name like weights (`sample#...`).  tests/c_host/denoise_host.c runs the reference's whole denoising
 * loop (pipe:663-700) that way. */
int i2v_unet_run(i2v_unet* h, const void* const* io, int32_t n_io, i2v_stream_t stream);
/* begin / end the capture of one step on `stream` (hipStreamBeginCapture, relaxed mode): everything launched on the stream in
 * between -- i2v_unet_forward and the host's DDIM kernels -- becomes the handle's step; i2v_unet_abort_capture ends a capture
 * whose launches failed and discards it */
int i2v_unet_capture_step(i2v_unet* h, i2v_stream_t stream);
int i2v_unet_end_capture(i2v_unet* h);
int i2v_unet_abort_capture(i2v_unet* h);
/* launch the captured step once (asynchronous on `stream`); I2V_ERR_INVALID_ARG when nothing has been captured */
int i2v_unet_replay_step(i2v_unet* h, i2v_stream_t stream);
int32_t i2v_unet_has_step(const i2v_unet* h);

#ifdef __cplusplus
}
#endif
#endif /* I2V_HIP_H */
