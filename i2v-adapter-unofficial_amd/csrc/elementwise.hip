// Layout edges (NCHW <-> token-major), timestep embedding, SiLU, row repeat / 2-D copy, and the two halves of
// the DDIM + classifier-free-guidance step that bracket the UNet call.  All pure HBM traffic.
#include "common.h"

namespace {

constexpr int EW_MAX_BLOCKS = 4096;

inline int ew_blocks(int64_t n) {
  const int64_t b = i2v_cdiv(n, 256);
  return (int)(b < EW_MAX_BLOCKS ? (b < 1 ? 1 : b) : EW_MAX_BLOCKS);
}

template <bool SRC_F32>
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const void* __restrict__ src, f16* __restrict__ dst, int n,
                                                             int c, int hw, int c_pad) {
  const int64_t total = (int64_t)n * hw * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c_pad);
    const int64_t t = i / c_pad;
    const int p = (int)(t % hw);
    const int64_t img = t / hw;
    float v = 0.f;
    if (ch < c) {
      const int64_t s = (img * c + ch) * hw + p;
      v = SRC_F32 ? reinterpret_cast<const float*>(src)[s] : (float)reinterpret_cast<const f16*>(src)[s];
    }
    dst[i] = (f16)v;
  }
}

template <bool SRC_F32, bool DST_F32>
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const void* __restrict__ src, int64_t ld, void* __restrict__ dst,
                                                             int n, int c, int hw) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const int64_t t = i / hw;
    const int ch = (int)(t % c);
    const int64_t img = t / c;
    const int64_t si = (img * hw + p) * ld + ch;
    const float v = SRC_F32 ? reinterpret_cast<const float*>(src)[si] : (float)reinterpret_cast<const f16*>(src)[si];
    if (DST_F32)
      reinterpret_cast<float*>(dst)[i] = v;
    else
      reinterpret_cast<f16*>(dst)[i] = (f16)v;
  }
}

__global__ __launch_bounds__(256) void timestep_embedding_kernel(const float* __restrict__ t, const int32_t* __restrict__ t_index,
                                                                 int t_rows, f16* __restrict__ out, int n, int dim) {
  const int half = dim / 2;
  const int total = n * half;
  // the device-side step counter is clamped to the table: a graph replayed more often than the table is long must not
  // read past it
  const int ti = t_index ? min(max(*t_index, 0), t_rows - 1) : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int row = i / half, j = i - row * half;
    const float tv = t_index ? t[ti] : t[row];
    // freq_j = exp(-ln(10000) * j / half); embedding = [cos | sin]  (flip_sin_to_cos = True, shift 0)
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);
    const float a = tv * freq;
    out[(int64_t)row * dim + j] = (f16)cosf(a);
    out[(int64_t)row * dim + half + j] = (f16)sinf(a);
  }
}

// out[c] = table[clamp(*row_index)][c]: the row of a per-timestep table that belongs to the step a replayed hipGraph is at
__global__ __launch_bounds__(256) void select_row_kernel(const f16* __restrict__ table, int64_t ld, int rows,
                                                         const int32_t* __restrict__ row_index, f16* __restrict__ out, int cols8) {
  const int r = min(max(*row_index, 0), rows - 1);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cols8; i += gridDim.x * 256)
    *reinterpret_cast<f16x8*>(out + 8 * i) = *reinterpret_cast<const f16x8*>(table + (int64_t)r * ld + 8 * i);
}

__global__ __launch_bounds__(256) void silu_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = (f16)silu_f((float)x[i]);
}

__global__ __launch_bounds__(256) void repeat_rows_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t rows_in,
                                                          int64_t cols, int repeat) {
  const int64_t total = rows_in * repeat * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols, cidx = i - r * cols;
    y[i] = x[(r / repeat) * cols + cidx];
  }
}

// 16 bytes per lane when the row length, the strides and both bases allow it (every caller of the step does: channel counts
// are multiples of 8); the index arithmetic runs once per 8 elements and in 32 bits
__global__ __launch_bounds__(256) void copy3d_vec8_kernel(const f16* __restrict__ src, int64_t sbs, int64_t ld_src,
                                                          f16* __restrict__ dst, int64_t dbs, int64_t ld_dst, int batches,
                                                          int rows, int cols8) {
  const int64_t total = (int64_t)batches * rows * cols8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const unsigned t = (unsigned)(i / (unsigned)cols8);          // < 2^31 rows in all (checked on the host)
    const unsigned c8 = (unsigned)(i - (int64_t)t * cols8);
    const unsigned b = t / (unsigned)rows, r = t - b * (unsigned)rows;
    *reinterpret_cast<f16x8*>(dst + b * dbs + r * ld_dst + 8 * c8) =
        *reinterpret_cast<const f16x8*>(src + b * sbs + r * ld_src + 8 * c8);
  }
}

__global__ __launch_bounds__(256) void copy3d_kernel(const f16* __restrict__ src, int64_t sbs, int64_t ld_src,
                                                     f16* __restrict__ dst, int64_t dbs, int64_t ld_dst, int64_t batches,
                                                     int64_t rows, int64_t cols) {
  const int64_t total = batches * rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t cidx = i % cols;
    const int64_t t = i / cols;
    const int64_t r = t % rows, b = t / rows;
    dst[b * dbs + r * ld_dst + cidx] = src[b * sbs + r * ld_src + cidx];
  }
}

// latents fp32 [b, f, c, hw]; cond fp32 [b, c, hw]; model_in fp16 [copies*b*f, hw, c_pad]
__global__ __launch_bounds__(256) void ddim_prep_kernel(float* __restrict__ latents, const float* __restrict__ cond,
                                                        f16* __restrict__ model_in, int b, int f, int c, int hw, int c_pad,
                                                        int copies) {
  const int64_t per_copy = (int64_t)b * f * hw * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_copy; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c_pad);
    int64_t t = i / c_pad;
    const int p = (int)(t % hw);
    t /= hw;
    const int fr = (int)(t % f);
    const int bb = (int)(t / f);
    float v = 0.f;
    if (ch < c) {
      const int64_t li = (((int64_t)bb * f + fr) * c + ch) * hw + p;
      if (fr == 0) {
        v = cond[((int64_t)bb * c + ch) * hw + p];
        latents[li] = v;  // latents[:, 0] = condition_image_latents
      } else {
        v = latents[li];
      }
    }
    const f16 hv = (f16)v;
    for (int k = 0; k < copies; ++k) model_in[k * per_copy + i] = hv;
  }
}

template <typename NP>
__global__ __launch_bounds__(256) void ddim_step_kernel(float* __restrict__ latents, const NP* __restrict__ np, int64_t ld_np,
                                                        const float* __restrict__ coef, int n_steps,
                                                        const int32_t* __restrict__ step_index, float guidance, int b,
                                                        int f, int c, int hw, int copies) {
  const int64_t total = (int64_t)b * f * c * hw;
  const float* cf = coef + 4 * (int64_t)min(max(*step_index, 0), n_steps - 1);
  const float sa_t = cf[0], sb_t = cf[1], sa_p = cf[2], sb_p = cf[3];
  const int64_t bf = (int64_t)b * f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % hw);
    int64_t t = i / hw;
    const int ch = (int)(t % c);
    const int64_t img = t / c;  // b * f + fr
    float eps;
    if (copies == 2) {
      const float u = (float)np[(img * hw + p) * ld_np + ch];
      const float cnd = (float)np[((bf + img) * hw + p) * ld_np + ch];
      eps = u + guidance * (cnd - u);
    } else {
      eps = (float)np[(img * hw + p) * ld_np + ch];
    }
    const float x = latents[i];
    const float x0 = (x - sb_t * eps) / sa_t;
    latents[i] = sa_p * x0 + sb_p * eps;
  }
}

// First-frame-similarity prior + add_noise (pipe:647-656), one pass:
//   prior = mask * blur3x3(cond) + (1 - mask) * cond,  mask = (u < strength)          (per frame, per element)
//   latents = sqrt(a_t) * prior + sqrt(1 - a_t) * noise                                 (DDIMScheduler.add_noise)
// blur = torchvision GaussianBlur(kernel_size=3): separable taps (ke, kc, ke), reflect padding.
__global__ __launch_bounds__(256) void prior_kernel(const float* __restrict__ cond, const float* __restrict__ u,
                                                    const float* __restrict__ noise, float* __restrict__ latents, int b,
                                                    int f, int c, int h, int w, float kc, float ke, float strength,
                                                    float sa, float sb) {
  const int64_t total = (int64_t)b * f * c * h * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % w);
    int64_t t = i / w;
    const int y = (int)(t % h);
    t /= h;
    const int ch = (int)(t % c);
    t /= c;
    const int bb = (int)(t / f);
    const float* img = cond + ((int64_t)bb * c + ch) * h * w;
    const float center = img[y * w + x];
    float v = center;
    if (u[i] < strength) {
      // reflect (no edge repeat): index -1 -> 1, n -> n - 2; a 1-pixel axis has nothing to reflect onto
      const int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
      const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
      const float r0 = ke * img[ym * w + xm] + kc * img[ym * w + x] + ke * img[ym * w + xp];
      const float r1 = ke * img[y * w + xm] + kc * center + ke * img[y * w + xp];
      const float r2 = ke * img[yp * w + xm] + kc * img[yp * w + x] + ke * img[yp * w + xp];
      v = ke * r0 + kc * r1 + ke * r2;
    }
    latents[i] = sa * v + sb * noise[i];
  }
}

// DiagonalGaussianDistribution.sample of the VAE encoder (pipe:627): moments [n, 2 c, hw] = (mean | logvar) ->
// mean + exp(0.5 clamp(logvar, -30, 20)) eps
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ moments, const float* __restrict__ eps,
                                                              float* __restrict__ out, int n, int c, int hw) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t img = i / ((int64_t)c * hw), rem = i - img * c * hw;
    const float mean = moments[img * 2 * c * hw + rem];
    const float logvar = fminf(fmaxf(moments[img * 2 * c * hw + (int64_t)c * hw + rem], -30.f), 20.f);
    out[i] = mean + expf(0.5f * logvar) * eps[i];
  }
}

// the counter wraps at the end of the table: replay n_steps + k of a captured step restarts the schedule at entry k
__global__ void bump_step_kernel(int32_t* step_index, int n_steps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int next = *step_index + 1;
    *step_index = (next >= n_steps || next < 0) ? 0 : next;
  }
}

// `ctx_frag` of i2v_cross_attn_fused_f16 (i2v:527-532, unet:1263-1279): this layer's K / V of the context as the MFMA operand
// fragments the fused kernel keeps in registers, [n_ctx][heads][KT * DT + DT * KT][64 lanes][4], zero beyond the context's length and
// the head's width.  K fragment (kt, t): lane l holds K[key = 16 kt + (l & 15)][channel = 16 t + 4 (l >> 4) + j]; V fragment (t, kt): lane l
// holds V^T[channel = 16 t + (l & 15)][key = 16 kt + 4 (l >> 4) + j].  One thread per (fragment, lane): a gather of <= 80 keys.
__global__ __launch_bounds__(256) void pack_ctx_fragments_kernel(const f16* __restrict__ k, int64_t ldk, const f16* __restrict__ vt,
                                                                 int64_t vt_row_stride, int64_t vt_batch_stride, f16* __restrict__ out,
                                                                 int n_ctx, int heads, int d, int ctx_len) {
  constexpr int KT = 5;
  const int dt = (d + 15) / 16, nf = 2 * KT * dt;
  const int64_t total = (int64_t)n_ctx * heads * nf * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    int64_t r = i >> 6;
    const int f = (int)(r % nf);
    r /= nf;
    const int h = (int)(r % heads), n = (int)(r / heads);
    f16x4 v = {0, 0, 0, 0};
    if (f < KT * dt) {
      const int kt = f / dt, t = f - kt * dt;
      const int key = 16 * kt + (lane & 15), ch0 = 16 * t + 4 * (lane >> 4);
      if (key < ctx_len) {
        const f16* src = k + ((int64_t)n * ctx_len + key) * ldk + h * d + ch0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ch0 + j < d) v[j] = src[j];
      }
    } else {
      const int g = f - KT * dt, t = g / KT, kt = g - t * KT;
      const int ch = 16 * t + (lane & 15), key0 = 16 * kt + 4 * (lane >> 4);
      if (ch < d) {
        const f16* src = vt + (int64_t)n * vt_batch_stride + (int64_t)(h * d + ch) * vt_row_stride + key0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (key0 + j < ctx_len) v[j] = src[j];
      }
    }
    *reinterpret_cast<f16x4*>(out + i * 4) = v;
  }
}

}  // namespace

extern "C" int i2v_nchw_to_tokens(const void* src, int32_t src_is_f32, void* dst, int32_t n, int32_t c, int32_t hw,
                                  int32_t c_pad, i2v_stream_t stream) {
  I2V_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0 && c_pad >= c, "i2v_nchw_to_tokens: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)n * hw * c_pad;
  if (src_is_f32)
    hipLaunchKernelGGL(nchw_to_tokens_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, s, src,
                       reinterpret_cast<f16*>(dst), n, c, hw, c_pad);
  else
    hipLaunchKernelGGL(nchw_to_tokens_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, s, src,
                       reinterpret_cast<f16*>(dst), n, c, hw, c_pad);
  return i2v_check_launch("i2v_nchw_to_tokens");
}

extern "C" int i2v_tokens_to_nchw(const void* src, int32_t src_is_f32, int64_t ld, void* dst, int32_t dst_is_f32, int32_t n,
                                  int32_t c, int32_t hw, i2v_stream_t stream) {
  I2V_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0 && ld >= c, "i2v_tokens_to_nchw: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)n * hw * c;
  const dim3 grid(ew_blocks(total)), block(256);
  if (src_is_f32 && dst_is_f32)
    hipLaunchKernelGGL((tokens_to_nchw_kernel<true, true>), grid, block, 0, s, src, ld, dst, n, c, hw);
  else if (src_is_f32)
    hipLaunchKernelGGL((tokens_to_nchw_kernel<true, false>), grid, block, 0, s, src, ld, dst, n, c, hw);
  else if (dst_is_f32)
    hipLaunchKernelGGL((tokens_to_nchw_kernel<false, true>), grid, block, 0, s, src, ld, dst, n, c, hw);
  else
    hipLaunchKernelGGL((tokens_to_nchw_kernel<false, false>), grid, block, 0, s, src, ld, dst, n, c, hw);
  return i2v_check_launch("i2v_tokens_to_nchw");
}

extern "C" int i2v_timestep_embedding(const float* t, const int32_t* t_index, int32_t t_rows, void* out, int32_t n,
                                      int32_t dim, i2v_stream_t stream) {
  I2V_CHECK_ARG(t && out && n > 0 && dim > 0 && dim % 2 == 0, "i2v_timestep_embedding: bad arguments");
  I2V_CHECK_ARG(t_index == nullptr || t_rows > 0, "i2v_timestep_embedding: t_rows must be the length of the table t_index walks");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(ew_blocks((int64_t)n * dim / 2)), dim3(256), 0, s, t, t_index,
                     t_rows, reinterpret_cast<f16*>(out), n, dim);
  return i2v_check_launch("i2v_timestep_embedding");
}

extern "C" int i2v_select_row_f16(const void* table, int64_t ld, int32_t rows, const int32_t* row_index, void* out, int32_t cols,
                                  i2v_stream_t stream) {
  I2V_CHECK_ARG(table && row_index && out && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols,
                "i2v_select_row_f16: cols and ld must be positive multiples of 8 with ld >= cols");
  I2V_CHECK_ARG(reinterpret_cast<uintptr_t>(table) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
                "i2v_select_row_f16: 16-byte aligned pointers expected");
  hipLaunchKernelGGL(select_row_kernel, dim3(ew_blocks(cols / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const f16*>(table), ld, rows, row_index, reinterpret_cast<f16*>(out), cols / 8);
  return i2v_check_launch("i2v_select_row_f16");
}

extern "C" int i2v_silu_f16(const void* x, void* y, int64_t n, i2v_stream_t stream) {
  I2V_CHECK_ARG(x && y && n > 0, "i2v_silu_f16: bad arguments");
  hipLaunchKernelGGL(silu_kernel, dim3(ew_blocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const f16*>(x), reinterpret_cast<f16*>(y), n);
  return i2v_check_launch("i2v_silu_f16");
}

extern "C" int i2v_repeat_rows_f16(const void* x, void* y, int64_t rows_in, int64_t cols, int32_t repeat,
                                   i2v_stream_t stream) {
  I2V_CHECK_ARG(x && y && rows_in > 0 && cols > 0 && repeat > 0, "i2v_repeat_rows_f16: bad arguments");
  hipLaunchKernelGGL(repeat_rows_kernel, dim3(ew_blocks(rows_in * repeat * cols)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(x), reinterpret_cast<f16*>(y),
                     rows_in, cols, repeat);
  return i2v_check_launch("i2v_repeat_rows_f16");
}

extern "C" int i2v_copy3d_f16(const void* src, int64_t src_batch_stride, int64_t ld_src, void* dst,
                              int64_t dst_batch_stride, int64_t ld_dst, int64_t batches, int64_t rows, int64_t cols,
                              i2v_stream_t stream) {
  I2V_CHECK_ARG(src && dst && batches > 0 && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols,
                "i2v_copy3d_f16: bad arguments");
  const bool vec8 = cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && src_batch_stride % 8 == 0 && dst_batch_stride % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                    batches * rows < (1ll << 31) && cols < (1ll << 31);
  if (vec8)
    hipLaunchKernelGGL(copy3d_vec8_kernel, dim3(ew_blocks(batches * rows * (cols / 8))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(src), src_batch_stride, ld_src,
                       reinterpret_cast<f16*>(dst), dst_batch_stride, ld_dst, (int)batches, (int)rows, (int)(cols / 8));
  else
    hipLaunchKernelGGL(copy3d_kernel, dim3(ew_blocks(batches * rows * cols)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(src), src_batch_stride, ld_src,
                       reinterpret_cast<f16*>(dst), dst_batch_stride, ld_dst, batches, rows, cols);
  return i2v_check_launch("i2v_copy3d_f16");
}

extern "C" int i2v_ddim_prep(float* latents, const float* cond, void* model_in, int32_t b, int32_t f, int32_t c,
                             int32_t hw, int32_t c_pad, int32_t cfg_copies, i2v_stream_t stream) {
  I2V_CHECK_ARG(latents && cond && model_in && b > 0 && f > 0 && c > 0 && hw > 0 && c_pad >= c,
                "i2v_ddim_prep: bad arguments");
  I2V_CHECK_ARG(cfg_copies == 1 || cfg_copies == 2, "i2v_ddim_prep: cfg_copies must be 1 or 2");
  hipLaunchKernelGGL(ddim_prep_kernel, dim3(ew_blocks((int64_t)b * f * hw * c_pad)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), latents, cond, reinterpret_cast<f16*>(model_in), b, f, c, hw,
                     c_pad, cfg_copies);
  return i2v_check_launch("i2v_ddim_prep");
}

extern "C" int i2v_ddim_cfg_step(float* latents, const void* noise_pred, int32_t np_is_f32, int64_t ld_np,
                                 const float* coef, int32_t n_steps, int32_t* step_index, float guidance_scale, int32_t b,
                                 int32_t f, int32_t c, int32_t hw, int32_t cfg_copies, i2v_stream_t stream) {
  I2V_CHECK_ARG(latents && noise_pred && coef && step_index && b > 0 && f > 0 && c > 0 && hw > 0 && ld_np >= c &&
                    n_steps > 0,
                "i2v_ddim_cfg_step: bad arguments");
  I2V_CHECK_ARG(cfg_copies == 1 || cfg_copies == 2, "i2v_ddim_cfg_step: cfg_copies must be 1 or 2");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (np_is_f32)
    hipLaunchKernelGGL(ddim_step_kernel<float>, dim3(ew_blocks((int64_t)b * f * c * hw)), dim3(256), 0, s, latents,
                       reinterpret_cast<const float*>(noise_pred), ld_np, coef, n_steps, step_index, guidance_scale, b, f,
                       c, hw, cfg_copies);
  else
    hipLaunchKernelGGL(ddim_step_kernel<f16>, dim3(ew_blocks((int64_t)b * f * c * hw)), dim3(256), 0, s, latents,
                       reinterpret_cast<const f16*>(noise_pred), ld_np, coef, n_steps, step_index, guidance_scale, b, f,
                       c, hw, cfg_copies);
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, s, step_index, n_steps);
  return i2v_check_launch("i2v_ddim_cfg_step");
}

extern "C" int i2v_first_frame_prior_f32(const float* cond, const float* mask_uniform, const float* noise, float* latents,
                                         int32_t b, int32_t f, int32_t c, int32_t h, int32_t w, float k_center,
                                         float k_edge, float strength, float sqrt_alpha, float sqrt_one_minus_alpha,
                                         i2v_stream_t stream) {
  I2V_CHECK_ARG(cond && mask_uniform && noise && latents && b > 0 && f > 0 && c > 0 && h > 0 && w > 0,
                "i2v_first_frame_prior_f32: bad arguments");
  hipLaunchKernelGGL(prior_kernel, dim3(ew_blocks((int64_t)b * f * c * h * w)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), cond, mask_uniform, noise, latents, b, f, c, h, w, k_center,
                     k_edge, strength, sqrt_alpha, sqrt_one_minus_alpha);
  return i2v_check_launch("i2v_first_frame_prior_f32");
}

extern "C" int i2v_gaussian_sample_f32(const float* moments, const float* eps, float* out, int32_t n, int32_t c,
                                       int32_t hw, i2v_stream_t stream) {
  I2V_CHECK_ARG(moments && eps && out && n > 0 && c > 0 && hw > 0, "i2v_gaussian_sample_f32: bad arguments");
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3(ew_blocks((int64_t)n * c * hw)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), moments, eps, out, n, c, hw);
  return i2v_check_launch("i2v_gaussian_sample_f32");
}

extern "C" int64_t i2v_pack_ctx_fragments_elems(int32_t n_ctx, int32_t heads, int32_t head_dim) {
  if (n_ctx <= 0 || heads <= 0 || head_dim <= 0) return 0;
  return (int64_t)n_ctx * heads * (2 * 5 * ((head_dim + 15) / 16)) * 256;
}

extern "C" int i2v_pack_ctx_fragments_f16(const void* k, int64_t ldk, const void* vt, int64_t vt_row_stride, int64_t vt_batch_stride,
                                          void* out, int32_t n_ctx, int32_t heads, int32_t head_dim, int32_t ctx_len,
                                          i2v_stream_t stream) {
  I2V_CHECK_ARG(k && vt && out, "i2v_pack_ctx_fragments_f16: null pointer");
  I2V_CHECK_ARG(n_ctx > 0 && heads > 0 && head_dim > 0 && ctx_len > 0 && ctx_len <= 80,
                "i2v_pack_ctx_fragments_f16: n_ctx %d heads %d head_dim %d ctx_len %d (1 .. 80 keys)", n_ctx, heads, head_dim, ctx_len);
  I2V_CHECK_ARG(ldk >= (int64_t)heads * head_dim && vt_row_stride >= ctx_len && vt_batch_stride >= (int64_t)heads * head_dim * vt_row_stride,
                "i2v_pack_ctx_fragments_f16: strides");
  I2V_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 7) == 0, "i2v_pack_ctx_fragments_f16: out must be 8-byte aligned");
  const int64_t total = i2v_pack_ctx_fragments_elems(n_ctx, heads, head_dim) / 4;
  hipLaunchKernelGGL(pack_ctx_fragments_kernel, dim3(ew_blocks(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const f16*>(k), ldk, reinterpret_cast<const f16*>(vt), vt_row_stride, vt_batch_stride,
                     reinterpret_cast<f16*>(out), n_ctx, heads, head_dim, ctx_len);
  return i2v_check_launch("i2v_pack_ctx_fragments_f16");
}
