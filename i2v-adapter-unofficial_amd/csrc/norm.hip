// GroupNorm(+SiLU) and LayerNorm(+positional embedding) on token-major fp16, fp32 statistics.  HBM-bound:
// 16-byte vector loads over the (frames x H x W) token axis, wavefront shuffles for the row reductions.
#include <cstdlib>

#include "common.h"

namespace {

// Rows (pixels) of one image handled by one statistics workgroup.  The grid is (chunks, images): it needs ~1000+
// workgroups to cover the 256 CUs (a 16 x 16 level with 32 images used to launch 32 workgroups and ran 8x below the
// HBM rate), but the per-chunk partial sums must stay a small fraction of the data (>= 16 rows per chunk).
static inline int gn_rows_per_chunk(int n_img, int hw) {
  const int want_chunks = (int)i2v_cdiv(1024, n_img);
  int rpc = (int)i2v_cdiv(hw, want_chunks);
  rpc = (rpc + 7) / 8 * 8;
  if (rpc < 16) rpc = 16;
  if (rpc > 256) rpc = 256;
  return rpc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------- GroupNorm
// Statistics are carried as (mean, M2 = sum of squared deviations) partials and merged with Chan's formula, never as
// E[x^2] - mean^2: real SD activations have channels with |mean| >> std, where the raw-moment form loses its digits to
// cancellation in fp32 (round-1 form).
// pass 1: per (image, row-chunk, channel) partial (mean, M2) over the chunk's rows.  Inside a chunk the sums run over
// data SHIFTED by the channel's value in the chunk's first row (a sample of the same distribution, so the shifted
// values are of the size of the spread, not of the mean): mean = K + S1 / n, M2 = S2 - S1^2 / n.
// gpartial != NULL (C <= GN_MAXC): additionally the chunk's per-GROUP partial (mean, M2) over rows x channels-of-the-group
// (Chan merge of the group's channels, equal counts), [img][chunk][group][2]: what gn_apply_fused_kernel finalises itself.
constexpr int GN_MAXC = 2560, GN_MAXG = 64;
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2,
                                                       int c2, int hw, int rpc, float* __restrict__ partial,
                                                       float* __restrict__ gpartial = nullptr, int groups = 0) {
  __shared__ float red[256 * 16];
  // per-channel (mean, M2) of the chunk for the per-group merge: DYNAMIC LDS, 8 C bytes with gpartial and none without (r5, ADVICE
  // r4: as a static 20 KB array every launch paid it -- backward and fallback passes too --, 36 KB per 256-thread block)
  extern __shared__ float chan[];
  const int C = c1 + c2, nvec = C / 8;
  const int chunk = blockIdx.x, img = blockIdx.y, nchunk = gridDim.x;
  const int row_begin = chunk * rpc;
  const int row_end = min(hw, row_begin + rpc);
  const int tid = threadIdx.x;
  const int cols_per_pass = nvec < 256 ? nvec : 256;
  const int rows_par = 256 / cols_per_pass;
  const int col_lane = tid % cols_per_pass, row_lane = tid / cols_per_pass;
  const bool active = row_lane < rows_par;
  const int nv1 = c1 / 8;

  for (int col0 = 0; col0 < nvec; col0 += cols_per_pass) {
    const int col = col0 + col_lane;
    float s[8], q[8], kshift[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = kshift[e] = 0.f;
    if (active && col < nvec) {
      const f16* base;
      int64_t ld;
      int coff;
      if (col < nv1) {
        base = x1 + (int64_t)img * hw * c1;
        ld = c1;
        coff = col * 8;
      } else {
        base = x2 + (int64_t)img * hw * c2;
        ld = c2;
        coff = (col - nv1) * 8;
      }
      const f16x8 k8 = ld_global_16B(base + (int64_t)row_begin * ld + coff);   // the same shift for every row lane
#pragma unroll
      for (int e = 0; e < 8; ++e) kshift[e] = (float)k8[e];
      auto add = [&](const f16x8 v) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e] - kshift[e];
          s[e] += f;
          q[e] += f * f;
        }
      };
      // four rows' loads in flight per thread (same summation order as one at a time): with a single 16-byte load per
      // thread and iteration a CU had ~16 KB in flight and the pass read at 2.6 TB/s
      int r = row_begin + row_lane;
      for (; r + 3 * rows_par < row_end; r += 4 * rows_par) {
        const f16* src = base + (int64_t)r * ld + coff;
        const f16x8 v0 = ld_global_16B(src), v1 = ld_global_16B(src + (int64_t)rows_par * ld),
                    v2 = ld_global_16B(src + 2 * (int64_t)rows_par * ld), v3 = ld_global_16B(src + 3 * (int64_t)rows_par * ld);
        add(v0);
        add(v1);
        add(v2);
        add(v3);
      }
      for (; r < row_end; r += rows_par) add(ld_global_16B(base + (int64_t)r * ld + coff));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // [value][thread]: consecutive lanes on consecutive banks (r5: as [thread][16 values] every
      red[e * 256 + tid] = s[e];       // store and load of a wave hit 4 banks 16 deep -- SQ_LDS_BANK_CONFLICT 0.71 of the LDS cycles)
      red[(8 + e) * 256 + tid] = q[e];
    }
    __syncthreads();
    if (row_lane == 0 && col < nvec) {
      for (int rl = 1; rl < rows_par; ++rl) {
        const int o = rl * cols_per_pass + col_lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += red[e * 256 + o];
          q[e] += red[(8 + e) * 256 + o];
        }
      }
      const float n = (float)(row_end - row_begin), inv_n = 1.0f / n;
      float* dst = partial + (((int64_t)img * nchunk + chunk) * C + col * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mean_c = kshift[e] + s[e] * inv_n, m2_c = fmaxf(q[e] - s[e] * s[e] * inv_n, 0.f);
        dst[2 * e] = mean_c;                  // mean of this (chunk, channel)
        dst[2 * e + 1] = m2_c;                // M2
        // [value e][column]: consecutive lanes write consecutive 8-byte slots.  (r6, VERDICT r5 7a: as chan[2 (8 col + e)] a wave's
        // stores were 16 floats apart -- every fourth lane on the same bank, 16 deep -- which is what the 0.716 SQ_LDS_BANK_CONFLICT of
        // the round-5 collection was made of after `red` had been transposed: few LDS cycles, nearly all of them conflicts.  The
        // kernel's time is its global loads -- 71 % wait -- so the step does not move.)
        if (gpartial != nullptr) *reinterpret_cast<float2*>(chan + 2 * (e * nvec + col)) = float2{mean_c, m2_c};
      }
    }
    __syncthreads();
  }
  if (gpartial != nullptr && tid < groups) {
    const int cpg = C / groups;
    const float n = (float)(row_end - row_begin);
    auto slot = [&](int ch) { return 2 * ((ch & 7) * nvec + (ch >> 3)); };      // channel ch = 8 col + e sits at [e][col]
    float mg = 0.f;
    for (int c = 0; c < cpg; ++c) mg += chan[slot(tid * cpg + c)];
    mg /= (float)cpg;
    float m2 = 0.f;
    for (int c = 0; c < cpg; ++c) {
      const float dm = chan[slot(tid * cpg + c)] - mg;
      m2 += chan[slot(tid * cpg + c) + 1] + n * dm * dm;
    }
    float* dst = gpartial + (((int64_t)img * nchunk + chunk) * groups + tid) * 2;
    dst[0] = mg;
    dst[1] = m2;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                       // red may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// pass 2: one workgroup per (stat group, channel group): Chan merge of the (mean_i, M2_i, n_i) partials in a fixed
// order (deterministic), then per-(image, channel) scale & shift.
//   mean = sum n_i mean_i / N;   M2 = sum M2_i + sum n_i (mean_i - mean)^2;   var = M2 / N
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int C,
                                                          int groups, int fps, int hw, int rpc, float eps,
                                                          const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                          float* __restrict__ coef) {
  __shared__ float red[4];
  const int sg = blockIdx.x, grp = blockIdx.y, lane = threadIdx.x;
  const int cpg = C / groups;
  const int total = fps * nchunk * cpg;
  const float cnt = (float)fps * (float)hw * (float)cpg;
  // the motion modules normalise over all frames of a clip (fps = 16): thousands of partials per group, each an
  // 8-byte strided load, so keep many of them in flight per workgroup
  float s = 0.f;
#pragma unroll 4
  for (int i = lane; i < total; i += 256) {
    const int c = i % cpg, t = i / cpg;
    const int ch = t % nchunk, f = t / nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc);
    s += n_i * partial[((((int64_t)(sg * fps + f)) * nchunk + ch) * C + grp * cpg + c) * 2];
  }
  const float mean = block_sum_256(s, red) / cnt;
  float m2 = 0.f;
#pragma unroll 4
  for (int i = lane; i < total; i += 256) {
    const int c = i % cpg, t = i / cpg;
    const int ch = t % nchunk, f = t / nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc);
    const float2 v = *reinterpret_cast<const float2*>(
        partial + ((((int64_t)(sg * fps + f)) * nchunk + ch) * C + grp * cpg + c) * 2);
    const float dm = v.x - mean;
    m2 += v.y + n_i * dm * dm;
  }
  const float var = block_sum_256(m2, red) / cnt;
  const float rstd = rsqrtf(var + eps);
  for (int i = lane; i < fps * cpg; i += 256) {
    const int c = grp * cpg + i % cpg, f = i / cpg;
    const float ga = (float)gamma[c] * rstd;
    float* dst = coef + ((int64_t)(sg * fps + f) * C + c) * 2;
    dst[0] = ga;
    dst[1] = (float)beta[c] - mean * ga;
  }
}

// pass 2 from the per-GROUP partials of gn_stats_kernel (cpg times fewer, contiguous loads): the clip-wide statistics of the
// motion modules' entry norms and the folded norms still finalise in a launch of their own.
__global__ __launch_bounds__(256) void gn_finalize_g_kernel(const float* __restrict__ gpartial, int nchunk, int C, int groups,
                                                            int fps, int hw, int rpc, float eps, const f16* __restrict__ gamma,
                                                            const f16* __restrict__ beta, float* __restrict__ coef) {
  __shared__ float red[4];
  const int sg = blockIdx.x, grp = blockIdx.y, lane = threadIdx.x;
  const int cpg = C / groups;
  const int total = fps * nchunk;
  const float cnt = (float)fps * (float)hw;          // rows; every row contributes cpg values
  const float2* gp = reinterpret_cast<const float2*>(gpartial) + (int64_t)sg * fps * nchunk * groups + grp;
  float s = 0.f;
  for (int i = lane; i < total; i += 256) {
    const int ch = i % nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc);
    s += n_i * gp[(int64_t)i * groups].x;
  }
  const float mean = block_sum_256(s, red) / cnt;
  float m2 = 0.f;
  for (int i = lane; i < total; i += 256) {
    const int ch = i % nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc) * (float)cpg;
    const float2 v = gp[(int64_t)i * groups];
    const float dm = v.x - mean;
    m2 += v.y + n_i * dm * dm;
  }
  const float var = block_sum_256(m2, red) / (cnt * (float)cpg);
  const float rstd = rsqrtf(var + eps);
  for (int i = lane; i < fps * cpg; i += 256) {
    const int c = grp * cpg + i % cpg, f = i / cpg;
    const float ga = (float)gamma[c] * rstd;
    float* dst = coef + ((int64_t)(sg * fps + f) * C + c) * 2;
    dst[0] = ga;
    dst[1] = (float)beta[c] - mean * ga;
  }
}

// pass 3: y = silu?(x * a + b), optional (b, f, p) -> (b, p, f) row permutation on store.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2,
                                                       int c2, const float* __restrict__ coef, f16* __restrict__ y,
                                                       int n_img, int hw, int silu, int out_perm, int frames) {
  const int C = c1 + c2, nvec = C / 8, nv1 = c1 / 8;
  const int64_t total = (int64_t)n_img * hw * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int col = (int)(idx % nvec);
    const int64_t rowg = idx / nvec;
    const int row = (int)(rowg % hw), img = (int)(rowg / hw);
    f16x8 v;
    if (col < nv1)
      v = ld_global_16B(x1 + ((int64_t)img * hw + row) * c1 + col * 8);
    else
      v = ld_global_16B(x2 + ((int64_t)img * hw + row) * c2 + (col - nv1) * 8);
    const float4* cf = reinterpret_cast<const float4*>(coef + ((int64_t)img * C + col * 8) * 2);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 ab = cf[e];  // (a, b) of channel 2e, (a, b) of channel 2e + 1
      float r0 = (float)v[2 * e] * ab.x + ab.y;
      float r1 = (float)v[2 * e + 1] * ab.z + ab.w;
      if (silu) {
        r0 = silu_f(r0);
        r1 = silu_f(r1);
      }
      o[2 * e] = (f16)r0;
      o[2 * e + 1] = (f16)r1;
    }
    int64_t orow = rowg;
    if (out_perm) {
      const int b = img / frames, f = img - b * frames;
      orow = ((int64_t)b * hw + row) * frames + f;
    }
    *reinterpret_cast<f16x8*>(y + orow * C + col * 8) = o;
  }
}

// passes 2 + 3 in one launch for per-image statistics (frames_per_stat == 1: the resnet norms and the spatial transformers'
// entry norms that are not folded): one workgroup per (row chunk, image) finalises the image's statistics ITSELF from the
// per-group partials of gn_stats_kernel -- nchunk x groups pairs, a few KB: staged through LDS, Chan-merged in a fixed order
// by one thread per group, turned into this thread's per-channel (scale, shift) registers -- and applies them to its rows.
// The separate finalize launch (1024 workgroups chasing 320 strided partials each through two block reductions: 8.2 us a
// call, 52 calls and 0.43 ms per step, profiles/r3_kernel_stats.txt) and the coefficient round trip disappear.
__global__ __launch_bounds__(256) void gn_apply_fused_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2,
                                                             int c2, const float* __restrict__ gpartial, int groups,
                                                             const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                             float eps, f16* __restrict__ y, int hw, int rpc, int silu,
                                                             int pchunks, int prpc) {
  // (pchunks partials of prpc rows each per image: gn_stats_kernel's own chunks, or the row tiles of the convolution that produced
  //  x and wrote the partials from its epilogue -- i2v_gemm_params.gn_partial; rpc / gridDim.x: the rows this launch applies)
  extern __shared__ float gsm[];                 // [pchunks * groups * 2] partials, then [groups * 2] (mean, rstd)
  const int C = c1 + c2, nvec = C / 8, nv1 = c1 / 8;
  const int chunk = blockIdx.x, img = blockIdx.y;
  const int tid = threadIdx.x;
  const int cpg = C / groups;
  float* stat = gsm + pchunks * groups * 2;
  for (int i = tid; i < pchunks * groups * 2; i += 256) gsm[i] = gpartial[(int64_t)img * pchunks * groups * 2 + i];
  __syncthreads();
  if (tid < groups) {
    float sum = 0.f;
    for (int ch = 0; ch < pchunks; ++ch) {
      const float n_i = (float)(min(hw, (ch + 1) * prpc) - ch * prpc);
      sum += n_i * gsm[(ch * groups + tid) * 2];
    }
    const float mean = sum / (float)hw;
    float m2 = 0.f;
    for (int ch = 0; ch < pchunks; ++ch) {
      const float n_i = (float)(min(hw, (ch + 1) * prpc) - ch * prpc) * (float)cpg;
      const float dm = gsm[(ch * groups + tid) * 2] - mean;
      m2 += gsm[(ch * groups + tid) * 2 + 1] + n_i * dm * dm;
    }
    stat[2 * tid] = mean;
    stat[2 * tid + 1] = rsqrtf(m2 / ((float)hw * (float)cpg) + eps);
  }
  __syncthreads();
  const int row_begin = chunk * rpc, row_end = min(hw, row_begin + rpc);
  const int cols_per_pass = nvec < 256 ? nvec : 256;
  const int rows_par = 256 / cols_per_pass;
  const int col_lane = tid % cols_per_pass, row_lane = tid / cols_per_pass;
  if (row_lane >= rows_par) return;
  for (int col0 = 0; col0 < nvec; col0 += cols_per_pass) {
    const int col = col0 + col_lane;
    if (col >= nvec) continue;
    const f16x8 ga = ld_global_16B(gamma + col * 8), be = ld_global_16B(beta + col * 8);
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (col * 8 + e) / cpg;
      a[e] = (float)ga[e] * stat[2 * g + 1];
      b[e] = (float)be[e] - stat[2 * g] * a[e];
    }
    const f16* base;
    int64_t ld;
    int coff;
    if (col < nv1) {
      base = x1 + (int64_t)img * hw * c1;
      ld = c1;
      coff = col * 8;
    } else {
      base = x2 + (int64_t)img * hw * c2;
      ld = c2;
      coff = (col - nv1) * 8;
    }
    f16* yb = y + (int64_t)img * hw * C + col * 8;
    auto one = [&](const f16x8 v, int r) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (float)v[e] * a[e] + b[e];
        if (silu) t = silu_f(t);
        o[e] = (f16)t;
      }
      *reinterpret_cast<f16x8*>(yb + (int64_t)r * C) = o;
    };
    int r = row_begin + row_lane;
    for (; r + 3 * rows_par < row_end; r += 4 * rows_par) {     // four rows' loads in flight per thread
      const f16* src = base + (int64_t)r * ld + coff;
      const f16x8 v0 = ld_global_16B(src), v1 = ld_global_16B(src + (int64_t)rows_par * ld),
                  v2 = ld_global_16B(src + 2 * (int64_t)rows_par * ld), v3 = ld_global_16B(src + 3 * (int64_t)rows_par * ld);
      one(v0, r);
      one(v1, r + rows_par);
      one(v2, r + 2 * rows_par);
      one(v3, r + 3 * rows_par);
    }
    for (; r < row_end; r += rows_par) one(ld_global_16B(base + (int64_t)r * ld + coff), r);
  }
}

// ---------------------------------------------------------------------------------------------- GroupNorm backward
// Input gradient of y = silu?(xh gamma + beta), xh = (x - mu) rstd over a statistics group of n elements (the norms are
// frozen in the adapter training step): with dz = dy silu'(z), g = dz gamma, S1 = sum g, S2 = sum g xh,
//   dx = rstd (g - S1 / n - xh S2 / n) = dz P_c + x Q + R,   P_c = gamma_c rstd, Q = -rstd^2 S2 / n, R = -rstd S1 / n - mu Q.
// pass 1 (this kernel): per (image, row-chunk, channel) sums A = sum dz and B = sum dz (x - k), k = the channel's value in
// the chunk's first row (the same shift as the forward statistics: sum dz (x - mu) = B + (k - mu) A has no cancellation
// against a large channel mean); z = x a + b from the forward's per-(image, channel) coefficients.
__device__ __forceinline__ float silu_grad(float z) {
  const float sg = 1.0f / (1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2,
                                                             int c2, const f16* __restrict__ dy, const float* __restrict__ coef,
                                                             int hw, int rpc, int silu, float* __restrict__ partial) {
  __shared__ float red[256 * 16];
  const int C = c1 + c2, nvec = C / 8;
  const int chunk = blockIdx.x, img = blockIdx.y, nchunk = gridDim.x;
  const int row_begin = chunk * rpc, row_end = min(hw, row_begin + rpc);
  const int tid = threadIdx.x;
  const int cols_per_pass = nvec < 256 ? nvec : 256;
  const int rows_par = 256 / cols_per_pass;
  const int col_lane = tid % cols_per_pass, row_lane = tid / cols_per_pass;
  const bool active = row_lane < rows_par;
  const int nv1 = c1 / 8;
  for (int col0 = 0; col0 < nvec; col0 += cols_per_pass) {
    const int col = col0 + col_lane;
    float sa[8], sb[8], kshift[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sa[e] = sb[e] = kshift[e] = 0.f;
    if (active && col < nvec) {
      const f16* base;
      int64_t ld;
      int coff;
      if (col < nv1) {
        base = x1 + (int64_t)img * hw * c1;
        ld = c1;
        coff = col * 8;
      } else {
        base = x2 + (int64_t)img * hw * c2;
        ld = c2;
        coff = (col - nv1) * 8;
      }
      const f16x8 k8 = ld_global_16B(base + (int64_t)row_begin * ld + coff);
      float ca[8], cb[8];
      const float4* cf = reinterpret_cast<const float4*>(coef + ((int64_t)img * C + col * 8) * 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 ab = cf[e];
        ca[2 * e] = ab.x;
        cb[2 * e] = ab.y;
        ca[2 * e + 1] = ab.z;
        cb[2 * e + 1] = ab.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) kshift[e] = (float)k8[e];
      for (int r = row_begin + row_lane; r < row_end; r += rows_par) {
        const f16x8 v = ld_global_16B(base + (int64_t)r * ld + coff);
        const f16x8 g = ld_global_16B(dy + ((int64_t)img * hw + r) * C + col * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (float)v[e];
          float dz = (float)g[e];
          if (silu) dz *= silu_grad(xv * ca[e] + cb[e]);
          sa[e] += dz;
          sb[e] += dz * (xv - kshift[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 16 + e] = sa[e];
      red[tid * 16 + 8 + e] = sb[e];
    }
    __syncthreads();
    if (row_lane == 0 && col < nvec) {
      for (int rl = 1; rl < rows_par; ++rl) {
        const int o = (rl * cols_per_pass + col_lane) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sa[e] += red[o + e];
          sb[e] += red[o + 8 + e];
        }
      }
      float* dst = partial + (((int64_t)img * nchunk + chunk) * C + col * 8) * 3;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dst[3 * e] = sa[e];
        dst[3 * e + 1] = sb[e];
        dst[3 * e + 2] = kshift[e];
      }
    }
    __syncthreads();
  }
}

// pass 2: one workgroup per (statistics group, channel group): mean / rstd again from the forward partials (Chan merge, as
// gn_finalize_kernel), S1, S2 from the backward partials, then (P, Q, R) per (image, channel)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ fpart, const float* __restrict__ bpart,
                                                              int nchunk, int C, int groups, int fps, int hw, int rpc, float eps,
                                                              const f16* __restrict__ gamma, float* __restrict__ bcoef) {
  __shared__ float red[4];
  const int sg = blockIdx.x, grp = blockIdx.y, lane = threadIdx.x;
  const int cpg = C / groups;
  const int total = fps * nchunk * cpg;
  const float cnt = (float)fps * (float)hw * (float)cpg;
  float s = 0.f;
  for (int i = lane; i < total; i += 256) {
    const int c = i % cpg, t = i / cpg;
    const int ch = t % nchunk, f = t / nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc);
    s += n_i * fpart[((((int64_t)(sg * fps + f)) * nchunk + ch) * C + grp * cpg + c) * 2];
  }
  const float mean = block_sum_256(s, red) / cnt;
  float m2 = 0.f;
  for (int i = lane; i < total; i += 256) {
    const int c = i % cpg, t = i / cpg;
    const int ch = t % nchunk, f = t / nchunk;
    const float n_i = (float)(min(hw, (ch + 1) * rpc) - ch * rpc);
    const float2 v = *reinterpret_cast<const float2*>(fpart + ((((int64_t)(sg * fps + f)) * nchunk + ch) * C + grp * cpg + c) * 2);
    const float dm = v.x - mean;
    m2 += v.y + n_i * dm * dm;
  }
  const float rstd = rsqrtf(block_sum_256(m2, red) / cnt + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < total; i += 256) {
    const int c = i % cpg, t = i / cpg;
    const int ch = t % nchunk, f = t / nchunk;
    const float* b3 = bpart + ((((int64_t)(sg * fps + f)) * nchunk + ch) * C + grp * cpg + c) * 3;
    const float ga = (float)gamma[grp * cpg + c];
    s1 += ga * b3[0];
    s2 += ga * (b3[1] + (b3[2] - mean) * b3[0]);
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red) * rstd;
  const float q = -rstd * rstd * s2 / cnt, r = -rstd * s1 / cnt - mean * q;
  for (int i = lane; i < fps * cpg; i += 256) {
    const int c = grp * cpg + i % cpg, f = i / cpg;
    float* dst = bcoef + ((int64_t)(sg * fps + f) * C + c) * 3;
    dst[0] = (float)gamma[c] * rstd;
    dst[1] = q;
    dst[2] = r;
  }
}

// pass 3: dx = dz P + x Q + R, split back onto the two sources of a channel-concatenated input
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2, int c2,
                                                           const f16* __restrict__ dy, const float* __restrict__ coef,
                                                           const float* __restrict__ bcoef, f16* __restrict__ dx1,
                                                           f16* __restrict__ dx2, int n_img, int hw, int silu) {
  const int C = c1 + c2, nvec = C / 8, nv1 = c1 / 8;
  const int64_t total = (int64_t)n_img * hw * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int col = (int)(idx % nvec);
    const int64_t rowg = idx / nvec;
    const int img = (int)(rowg / hw);
    const bool second = col >= nv1;
    const int64_t off = second ? rowg * c2 + (col - nv1) * 8 : rowg * c1 + col * 8;
    const f16x8 v = ld_global_16B((second ? x2 : x1) + off);
    const f16x8 g = ld_global_16B(dy + rowg * C + col * 8);
    const float* cf = coef + ((int64_t)img * C + col * 8) * 2;
    const float* bc = bcoef + ((int64_t)img * C + col * 8) * 3;
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xv = (float)v[e];
      float dz = (float)g[e];
      if (silu) dz *= silu_grad(xv * cf[2 * e] + cf[2 * e + 1]);
      o[e] = (f16)(dz * bc[3 * e] + xv * bc[3 * e + 1] + bc[3 * e + 2]);
    }
    *reinterpret_cast<f16x8*>((second ? dx2 : dx1) + off) = o;
  }
}

// One-launch GroupNorm for the small levels (8 x 8 ... 32 x 32): a workgroup owns a whole (statistics group, block of
// GB channel groups) slab -- fps * hw rows x GB * cpg channels, <= 64 KiB of fp16 -- reads it ONCE into LDS, takes the
// exact two-pass statistics there (mean, then sum of squared deviations: no partials, no merge), applies scale / shift
// (+ SiLU, + the row permutation) and writes it once.  Against the three-launch form (statistics, finalize, apply: two
// reads + one write and ~3 launch latencies, 20-40 us at these sizes) it is one read + one write in one launch.
__global__ __launch_bounds__(256) void gn_slab_kernel(const f16* __restrict__ x1, int c1, const f16* __restrict__ x2, int c2,
                                                      const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                      f16* __restrict__ y, int hw, int fps, int groups, int gb, float eps,
                                                      int silu, int out_perm, int frames) {
  extern __shared__ __attribute__((aligned(16))) char gn_lds[];
  __shared__ float red[4];
  const int C = c1 + c2, cpg = C / groups, cb = gb * cpg, vpr = cb / 8;
  const int rows = fps * hw;
  const int ch0 = blockIdx.x * cb, sg = blockIdx.y, tid = threadIdx.x;
  f16* slab = reinterpret_cast<f16*>(gn_lds);
  float* coef = reinterpret_cast<float*>(gn_lds + (size_t)rows * cb * 2);   // (a, b) per channel of the block
  const int nvec = rows * vpr;
  for (int i = tid; i < nvec; i += 256) {
    const int r = i / vpr, v = i - r * vpr;
    const int ch = ch0 + v * 8;
    const int64_t grow = (int64_t)sg * rows + r;                            // = (sg * fps + frame) * hw + pixel
    const f16x8 val = ch < c1 ? ld_global_16B(x1 + grow * c1 + ch) : ld_global_16B(x2 + grow * c2 + (ch - c1));
    *reinterpret_cast<f16x8*>(slab + (size_t)r * cb + v * 8) = val;
  }
  __syncthreads();
  const int qpg = cpg / 4;                 // 8-byte quads per row and group (cpg % 4 == 0, checked on the host)
  const int nquad = rows * qpg;
  const float inv_cnt = 1.0f / ((float)rows * (float)cpg);
  for (int g = 0; g < gb; ++g) {
    float s = 0.f;
    for (int i = tid; i < nquad; i += 256) {
      const int r = i / qpg, q = i - r * qpg;
      const f16x4 v4 = *reinterpret_cast<const f16x4*>(slab + (size_t)r * cb + g * cpg + q * 4);
      s += ((float)v4[0] + (float)v4[1]) + ((float)v4[2] + (float)v4[3]);
    }
    const float mean = block_sum_256(s, red) * inv_cnt;
    float m2 = 0.f;
    for (int i = tid; i < nquad; i += 256) {
      const int r = i / qpg, q = i - r * qpg;
      const f16x4 v4 = *reinterpret_cast<const f16x4*>(slab + (size_t)r * cb + g * cpg + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = (float)v4[e] - mean;
        m2 += d * d;
      }
    }
    const float rstd = rsqrtf(block_sum_256(m2, red) * inv_cnt + eps);
    for (int i = tid; i < cpg; i += 256) {
      const int c = g * cpg + i;
      const float ga = (float)gamma[ch0 + c] * rstd;
      coef[2 * c] = ga;
      coef[2 * c + 1] = (float)beta[ch0 + c] - mean * ga;
    }
  }
  __syncthreads();
  for (int i = tid; i < nvec; i += 256) {
    const int r = i / vpr, v = i - r * vpr;
    const f16x8 val = *reinterpret_cast<const f16x8*>(slab + (size_t)r * cb + v * 8);
    const float4* cf = reinterpret_cast<const float4*>(coef + v * 16);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 ab = cf[e];
      float r0 = (float)val[2 * e] * ab.x + ab.y;
      float r1 = (float)val[2 * e + 1] * ab.z + ab.w;
      if (silu) {
        r0 = silu_f(r0);
        r1 = silu_f(r1);
      }
      o[2 * e] = (f16)r0;
      o[2 * e + 1] = (f16)r1;
    }
    int64_t orow = (int64_t)sg * rows + r;
    if (out_perm) {   // images are (b, f): output row = (b * hw + pixel) * frames + f
      const int64_t img = orow / hw;
      const int pix = (int)(orow - img * hw);
      const int64_t b = img / frames;
      const int f = (int)(img - b * frames);
      orow = (b * hw + pix) * frames + f;
    }
    *reinterpret_cast<f16x8*>(y + orow * C + ch0 + v * 8) = o;
  }
}

// channel groups per workgroup of the slab kernel for this problem, 0 = use the three-launch form
static int gn_slab_plan(int n_img, int hw, int C, int groups, int fps, size_t* lds_bytes) {
  static const int off = getenv("I2V_GN_SLAB") ? (atoi(getenv("I2V_GN_SLAB")) == 0) : 0;
  if (off) return 0;
  const int cpg = C / groups;
  if (cpg % 4 != 0) return 0;
  const int64_t rows = (int64_t)fps * hw;
  // measured (tools/gn_probe.py): slabs up to 64 KiB win (8 x 8: 15 -> 9 us, 28 -> 12 us with the skip concatenated;
  // 16 x 16: 30 -> 23, 54 -> 39, 41 -> 32 us); the 80 KiB slabs of the 32 x 32 level leave one workgroup per CU with
  // its load / statistics / store phases in series and LOSE (37 -> 65 us), so that level keeps the three-launch form.
  // Smallest block of groups whose rows are >= 64 contiguous bytes: small slabs = many workgroups per CU.
  static const int64_t MAX_LDS = getenv("I2V_GN_SLAB_MAX") ? atoi(getenv("I2V_GN_SLAB_MAX")) : 64 * 1024;
  static const int ROWB = getenv("I2V_GN_SLAB_ROWB") ? atoi(getenv("I2V_GN_SLAB_ROWB")) : 64;
  int best = 0;
  for (int gb = 1; gb <= groups; gb *= 2) {
    if (groups % gb != 0 || (gb * cpg) % 8 != 0) continue;
    const int64_t bytes = rows * gb * cpg * 2 + (int64_t)gb * cpg * 8;
    if (bytes > MAX_LDS) break;
    const int64_t wgs = (int64_t)(groups / gb) * (n_img / fps);
    if (wgs < 128) break;                          // too few workgroups for the chip: keep the many-chunk form
    best = gb;
    if (gb * cpg * 2 >= ROWB) break;   // rows of >= ROWB contiguous bytes
  }
  if (best && lds_bytes) *lds_bytes = (size_t)(rows * best * cpg * 2 + (int64_t)best * cpg * 8);
  return best;
}

// GroupNorm folded into the consuming Linear: one workgroup per (output row n, statistics group s)
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ coef, int fps, int C, const f16* __restrict__ w,
                                                      int64_t ldw, const f16* __restrict__ bias, int n_out,
                                                      f16* __restrict__ w_out, f16* __restrict__ bias_out) {
  __shared__ float red[4];
  const int n = blockIdx.x, sg = blockIdx.y;
  const float2* ab = reinterpret_cast<const float2*>(coef) + (int64_t)sg * fps * C;   // the group's first image
  const f16* wr = w + (int64_t)n * ldw;
  f16* wo = w_out + ((int64_t)sg * n_out + n) * C;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float2 v = ab[c];
    const float wv = (float)wr[c];
    wo[c] = (f16)(wv * v.x);
    acc += wv * v.y;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) bias_out[(int64_t)sg * n_out + n] = (f16)(acc + (bias ? (float)bias[n] : 0.f));
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// one wave per R consecutive rows; the rows live in registers (NV 16-byte vectors per lane and row) so the variance
// is an exact second pass over registers.  R > 1 is memory-level parallelism: a 320-channel row is one 640-byte load
// per wave, and with one row per wave a CU had ~20 KB in flight (3.2 TB/s measured); gamma / beta are read once per wave.
template <int NV, int R>
__global__ __launch_bounds__(256) void ln_kernel(const i2v_ln_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= p.rows) return;
  const int nvec = p.C / 8;
  const f16* x = reinterpret_cast<const f16*>(p.x);
  float v[R][NV][8];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const bool live = row0 + r < p.rows;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + 64 * i;
      f16x8 t = zero8();
      if (live && vi < nvec) {
        const int row = row0 + r;
        int64_t off = (int64_t)row * p.ldx;
        if (p.x_rows_per_batch > 0) {   // batched row blocks (frame-0 rows of every clip): wave-uniform arithmetic
          const int b = row / p.x_rows_per_batch;
          off = (int64_t)b * p.x_batch_stride + (int64_t)(row - b * p.x_rows_per_batch) * p.ldx;
        }
        t = ld_global_16B(x + off + vi * 8);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[r][i][e] = (float)t[e];
        s[r] += v[r][i][e];
      }
    }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mean[r] = wave_sum(s[r]) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + 64 * i < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[r][i][e] - mean[r];
          q += d * d;
        }
      }
    }
    rstd[r] = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
  }
  const f16* gamma = reinterpret_cast<const f16*>(p.gamma);
  const f16* beta = reinterpret_cast<const f16*>(p.beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nvec) {
      const f16x8 ga = ld_global_16B(gamma + vi * 8), be = ld_global_16B(beta + vi * 8);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row < p.rows) {
          f16x8 pv = zero8();
          if (p.pe) pv = ld_global_16B(reinterpret_cast<const f16*>(p.pe) + (int64_t)(row % p.pe_period) * p.ld_pe + vi * 8);
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = (v[r][i][e] - mean[r]) * rstd[r] * (float)ga[e] + (float)be[e];
            if (p.pe) t += (float)pv[e];
            o[e] = (f16)t;
          }
          *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.y) + (int64_t)row * p.ldy + vi * 8) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- row softmax
// y[r, :] = softmax(scale * x[r, :]) over `cols` columns, one wave per row, fp32 statistics; used by the VAE mid-block
// attention (one head of dim 512 > the flash kernel's 160: QK^T and PV run as GEMMs around this kernel).  Three passes
// over the row (max, sum, write): the second and third hit L2.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* __restrict__ x, int64_t ldx, f16* __restrict__ y,
                                                           int64_t ldy, int rows, int cols, float scale_log2) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f16* xr = x + (int64_t)row * ldx;
  f16* yr = y + (int64_t)row * ldy;
  const int nvec = cols / 8;
  float mx = -INFINITY;
  for (int v = lane; v < nvec; v += 64) {
    const f16x8 t = ld_global_16B(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)t[e]);
  }
  for (int c = nvec * 8 + lane; c < cols; c += 64) mx = fmaxf(mx, (float)xr[c]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  // exp(scale (x - max)) = exp2(scale_log2 x - scale_log2 max)   (scale > 0)
  const float mb = mx * scale_log2;
  float sum = 0.f;
  for (int v = lane; v < nvec; v += 64) {
    const f16x8 t = ld_global_16B(xr + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f((float)t[e] * scale_log2 - mb);
  }
  for (int c = nvec * 8 + lane; c < cols; c += 64) sum += __builtin_amdgcn_exp2f((float)xr[c] * scale_log2 - mb);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int v = lane; v < nvec; v += 64) {
    const f16x8 t = ld_global_16B(xr + v * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(__builtin_amdgcn_exp2f((float)t[e] * scale_log2 - mb) * inv);
    *reinterpret_cast<f16x8*>(yr + v * 8) = o;
  }
  for (int c = nvec * 8 + lane; c < cols; c += 64) yr[c] = (f16)(__builtin_amdgcn_exp2f((float)xr[c] * scale_log2 - mb) * inv);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int64_t i2v_groupnorm_workspace_bytes(int32_t n_img, int32_t hw, int32_t channels) {
  const int64_t nchunk = i2v_cdiv(hw, gn_rows_per_chunk(n_img, hw));
  // per-(image, chunk, channel) partials, per-(image, channel) coefficients, per-(image, chunk, group <= 64) partials
  return ((int64_t)n_img * nchunk * channels * 2 + (int64_t)n_img * channels * 2 + (int64_t)n_img * nchunk * GN_MAXG * 2) *
         (int64_t)sizeof(float);
}

extern "C" int i2v_groupnorm_fold_f16(const i2v_gn_params* pp, const void* w, int64_t ldw, const void* bias, int32_t n_out,
                                      void* w_out, void* bias_out, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_groupnorm_fold_f16: null params");
  const i2v_gn_params& p = *pp;
  const int C = p.c1 + p.c2;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.workspace && w && w_out && bias_out, "i2v_groupnorm_fold_f16: null pointer");
  I2V_CHECK_ARG(p.n_img > 0 && p.hw > 0 && p.c1 > 0 && p.c2 >= 0 && n_out > 0 && ldw >= C,
                "i2v_groupnorm_fold_f16: bad sizes");
  I2V_CHECK_ARG((p.c2 == 0) == (p.x2 == nullptr), "i2v_groupnorm_fold_f16: x2 / c2 mismatch");
  I2V_CHECK_ARG(p.c1 % 8 == 0 && p.c2 % 8 == 0, "i2v_groupnorm_fold_f16: channel counts must be multiples of 8");
  I2V_CHECK_ARG(p.groups > 0 && C % p.groups == 0, "i2v_groupnorm_fold_f16: channels %d not divisible by groups %d", C,
                p.groups);
  I2V_CHECK_ARG(p.frames_per_stat > 0 && p.n_img % p.frames_per_stat == 0,
                "i2v_groupnorm_fold_f16: n_img %d not divisible by frames_per_stat %d", p.n_img, p.frames_per_stat);
  I2V_CHECK_ARG(al16(p.x) && (!p.x2 || al16(p.x2)) && al16(p.workspace) && al16(p.gamma) && al16(p.beta),
                "i2v_groupnorm_fold_f16: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpc = gn_rows_per_chunk(p.n_img, p.hw);
  const int nchunk = (int)i2v_cdiv(p.hw, rpc);
  float* partial = reinterpret_cast<float*>(p.workspace);
  float* coef = partial + (int64_t)p.n_img * nchunk * C * 2;
  if (p.groups <= GN_MAXG && C <= GN_MAXC) {
    float* gpartial = coef + (int64_t)p.n_img * C * 2;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), (size_t)C * 2 * sizeof(float), s,
                       reinterpret_cast<const f16*>(p.x), p.c1, reinterpret_cast<const f16*>(p.x2), p.c2, p.hw, rpc, partial, gpartial,
                       p.groups);
    hipLaunchKernelGGL(gn_finalize_g_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, gpartial, nchunk, C,
                       p.groups, p.frames_per_stat, p.hw, rpc, p.eps, reinterpret_cast<const f16*>(p.gamma),
                       reinterpret_cast<const f16*>(p.beta), coef);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), 0, s, reinterpret_cast<const f16*>(p.x), p.c1,
                       reinterpret_cast<const f16*>(p.x2), p.c2, p.hw, rpc, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, partial, nchunk, C,
                       p.groups, p.frames_per_stat, p.hw, rpc, p.eps, reinterpret_cast<const f16*>(p.gamma),
                       reinterpret_cast<const f16*>(p.beta), coef);
  }
  hipLaunchKernelGGL(gn_fold_kernel, dim3(n_out, p.n_img / p.frames_per_stat), dim3(256), 0, s, coef, p.frames_per_stat, C,
                     reinterpret_cast<const f16*>(w), ldw, reinterpret_cast<const f16*>(bias), n_out,
                     reinterpret_cast<f16*>(w_out), reinterpret_cast<f16*>(bias_out));
  return i2v_check_launch("i2v_groupnorm_fold_f16");
}

extern "C" int i2v_groupnorm_f16(const i2v_gn_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_groupnorm_f16: null params");
  const i2v_gn_params& p = *pp;
  const int C = p.c1 + p.c2;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.y && p.workspace, "i2v_groupnorm_f16: null pointer");
  I2V_CHECK_ARG(p.n_img > 0 && p.hw > 0 && p.c1 > 0 && p.c2 >= 0, "i2v_groupnorm_f16: bad sizes");
  I2V_CHECK_ARG((p.c2 == 0) == (p.x2 == nullptr), "i2v_groupnorm_f16: x2 / c2 mismatch");
  I2V_CHECK_ARG(p.c1 % 8 == 0 && p.c2 % 8 == 0, "i2v_groupnorm_f16: channel counts must be multiples of 8 (%d, %d)",
                p.c1, p.c2);
  I2V_CHECK_ARG(p.groups > 0 && C % p.groups == 0, "i2v_groupnorm_f16: channels %d not divisible by groups %d", C,
                p.groups);
  I2V_CHECK_ARG(p.frames_per_stat > 0 && p.n_img % p.frames_per_stat == 0,
                "i2v_groupnorm_f16: n_img %d not divisible by frames_per_stat %d", p.n_img, p.frames_per_stat);
  if (p.out_perm) I2V_CHECK_ARG(p.frames > 0 && p.n_img % p.frames == 0, "i2v_groupnorm_f16: bad frames for out_perm");
  I2V_CHECK_ARG(al16(p.x) && al16(p.y) && (!p.x2 || al16(p.x2)) && al16(p.workspace) && al16(p.gamma) && al16(p.beta),
                "i2v_groupnorm_f16: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.gpartial_in != nullptr) {
    // the statistics arrive as per-group partials from the epilogue of the convolution that wrote x: finalise + apply only
    I2V_CHECK_ARG(p.frames_per_stat == 1 && !p.out_perm && p.x2 == nullptr && p.groups <= GN_MAXG && C <= GN_MAXC,
                  "i2v_groupnorm_f16: gpartial_in is for per-image statistics of one source without a row permutation");
    I2V_CHECK_ARG(p.gpartial_rows > 0 && p.hw % p.gpartial_rows == 0 && (reinterpret_cast<uintptr_t>(p.gpartial_in) & 7) == 0,
                  "i2v_groupnorm_f16: gpartial_rows (%d) must divide hw (%d)", p.gpartial_rows, p.hw);
    const int pchunks = p.hw / p.gpartial_rows;
    const size_t lds = ((size_t)pchunks * p.groups * 2 + (size_t)p.groups * 2) * sizeof(float);
    I2V_CHECK_ARG(lds <= 48 * 1024, "i2v_groupnorm_f16: %d partial blocks per image do not fit the finalise step", pchunks);
    const int rpc = gn_rows_per_chunk(p.n_img, p.hw);
    hipLaunchKernelGGL(gn_apply_fused_kernel, dim3((unsigned)i2v_cdiv(p.hw, rpc), p.n_img), dim3(256), lds, s,
                       reinterpret_cast<const f16*>(p.x), p.c1, static_cast<const f16*>(nullptr), 0,
                       reinterpret_cast<const float*>(p.gpartial_in), p.groups, reinterpret_cast<const f16*>(p.gamma),
                       reinterpret_cast<const f16*>(p.beta), p.eps, reinterpret_cast<f16*>(p.y), p.hw, rpc, p.silu, pchunks,
                       p.gpartial_rows);
    return i2v_check_launch("i2v_groupnorm_f16(partials from the producer)");
  }
  {
    size_t lds = 0;
    const int gb = gn_slab_plan(p.n_img, p.hw, C, p.groups, p.frames_per_stat, &lds);
    if (gb > 0) {
      static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gn_slab_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 97 * 1024) == hipSuccess;
      if (attr_ok) {
        hipLaunchKernelGGL(gn_slab_kernel, dim3(p.groups / gb, p.n_img / p.frames_per_stat), dim3(256), lds, s,
                           reinterpret_cast<const f16*>(p.x), p.c1, reinterpret_cast<const f16*>(p.x2), p.c2,
                           reinterpret_cast<const f16*>(p.gamma), reinterpret_cast<const f16*>(p.beta),
                           reinterpret_cast<f16*>(p.y), p.hw, p.frames_per_stat, p.groups, gb, p.eps, p.silu, p.out_perm,
                           p.frames);
        return i2v_check_launch("i2v_groupnorm_f16(slab)");
      }
    }
  }
  const int rpc = gn_rows_per_chunk(p.n_img, p.hw);
  const int nchunk = (int)i2v_cdiv(p.hw, rpc);
  float* partial = reinterpret_cast<float*>(p.workspace);
  float* coef = partial + (int64_t)p.n_img * nchunk * C * 2;
  const f16* x1 = reinterpret_cast<const f16*>(p.x);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);
  // per-image statistics: two launches (statistics with per-group partials; finalise + apply), see gn_apply_fused_kernel
  static const int fused_off = getenv("I2V_GN_FUSED") ? (atoi(getenv("I2V_GN_FUSED")) == 0) : 0;
  const size_t fused_lds = ((size_t)nchunk * p.groups * 2 + (size_t)p.groups * 2) * sizeof(float);
  if (!fused_off && p.frames_per_stat == 1 && !p.out_perm && p.groups <= GN_MAXG && C <= GN_MAXC && fused_lds <= 48 * 1024) {
    float* gpartial = coef + (int64_t)p.n_img * C * 2;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), (size_t)C * 2 * sizeof(float), s, x1, p.c1, x2, p.c2, p.hw,
                       rpc, partial, gpartial, p.groups);
    hipLaunchKernelGGL(gn_apply_fused_kernel, dim3(nchunk, p.n_img), dim3(256), fused_lds, s, x1, p.c1, x2, p.c2, gpartial, p.groups,
                       reinterpret_cast<const f16*>(p.gamma), reinterpret_cast<const f16*>(p.beta), p.eps,
                       reinterpret_cast<f16*>(p.y), p.hw, rpc, p.silu, nchunk, rpc);
    return i2v_check_launch("i2v_groupnorm_f16(fused finalize)");
  }
  if (!fused_off && p.groups <= GN_MAXG && C <= GN_MAXC) {
    float* gpartial = coef + (int64_t)p.n_img * C * 2;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), (size_t)C * 2 * sizeof(float), s, x1, p.c1, x2, p.c2, p.hw,
                       rpc, partial, gpartial, p.groups);
    hipLaunchKernelGGL(gn_finalize_g_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, gpartial, nchunk, C,
                       p.groups, p.frames_per_stat, p.hw, rpc, p.eps, reinterpret_cast<const f16*>(p.gamma),
                       reinterpret_cast<const f16*>(p.beta), coef);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), 0, s, x1, p.c1, x2, p.c2, p.hw, rpc, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, partial, nchunk, C,
                       p.groups, p.frames_per_stat, p.hw, rpc, p.eps, reinterpret_cast<const f16*>(p.gamma),
                       reinterpret_cast<const f16*>(p.beta), coef);
  }
  const int64_t total = (int64_t)p.n_img * p.hw * (C / 8);
  const int blocks = (int)(i2v_cdiv(total, 256) < 4096 ? i2v_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), 0, s, x1, p.c1, x2, p.c2, coef,
                     reinterpret_cast<f16*>(p.y), p.n_img, p.hw, p.silu, p.out_perm, p.frames);
  return i2v_check_launch("i2v_groupnorm_f16");
}


extern "C" int64_t i2v_groupnorm_bwd_workspace_bytes(int32_t n_img, int32_t hw, int32_t channels) {
  const int64_t nchunk = i2v_cdiv(hw, gn_rows_per_chunk(n_img, hw));
  return ((int64_t)n_img * nchunk * channels * 5 + (int64_t)n_img * channels * 5) * (int64_t)sizeof(float);
}

extern "C" int i2v_groupnorm_bwd_f16(const i2v_gn_params* pp, const void* dy, void* dx, void* dx2, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_groupnorm_bwd_f16: null params");
  const i2v_gn_params& p = *pp;
  const int C = p.c1 + p.c2;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.workspace && dy && dx, "i2v_groupnorm_bwd_f16: null pointer");
  I2V_CHECK_ARG(p.n_img > 0 && p.hw > 0 && p.c1 > 0 && p.c2 >= 0, "i2v_groupnorm_bwd_f16: bad sizes");
  I2V_CHECK_ARG((p.c2 == 0) == (p.x2 == nullptr) && (p.c2 == 0) == (dx2 == nullptr), "i2v_groupnorm_bwd_f16: x2 / dx2 / c2 mismatch");
  I2V_CHECK_ARG(p.c1 % 8 == 0 && p.c2 % 8 == 0 && p.groups > 0 && C % p.groups == 0, "i2v_groupnorm_bwd_f16: channel counts");
  I2V_CHECK_ARG(p.frames_per_stat > 0 && p.n_img % p.frames_per_stat == 0 && !p.out_perm,
                "i2v_groupnorm_bwd_f16: frames_per_stat must divide n_img; out_perm is not supported");
  I2V_CHECK_ARG(al16(p.x) && (!p.x2 || al16(p.x2)) && al16(dy) && al16(dx) && (!dx2 || al16(dx2)) && al16(p.workspace) &&
                    al16(p.gamma) && al16(p.beta), "i2v_groupnorm_bwd_f16: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpc = gn_rows_per_chunk(p.n_img, p.hw);
  const int nchunk = (int)i2v_cdiv(p.hw, rpc);
  float* fpart = reinterpret_cast<float*>(p.workspace);
  float* coef = fpart + (int64_t)p.n_img * nchunk * C * 2;
  float* bpart = coef + (int64_t)p.n_img * C * 2;
  float* bcoef = bpart + (int64_t)p.n_img * nchunk * C * 3;
  const f16* x1 = reinterpret_cast<const f16*>(p.x);
  const f16* x2 = reinterpret_cast<const f16*>(p.x2);
  const f16* g = reinterpret_cast<const f16*>(p.gamma);
  // the forward statistics again (the inference forward keeps none), then the three backward passes
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, p.n_img), dim3(256), 0, s, x1, p.c1, x2, p.c2, p.hw, rpc, fpart);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, fpart, nchunk, C,
                     p.groups, p.frames_per_stat, p.hw, rpc, p.eps, g, reinterpret_cast<const f16*>(p.beta), coef);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, p.n_img), dim3(256), 0, s, x1, p.c1, x2, p.c2,
                     reinterpret_cast<const f16*>(dy), coef, p.hw, rpc, p.silu, bpart);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(p.n_img / p.frames_per_stat, p.groups), dim3(256), 0, s, fpart, bpart, nchunk,
                     C, p.groups, p.frames_per_stat, p.hw, rpc, p.eps, g, bcoef);
  const int64_t total = (int64_t)p.n_img * p.hw * (C / 8);
  const int blocks = (int)(i2v_cdiv(total, 256) < 4096 ? i2v_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, s, x1, p.c1, x2, p.c2, reinterpret_cast<const f16*>(dy),
                     coef, bcoef, reinterpret_cast<f16*>(dx), reinterpret_cast<f16*>(dx2), p.n_img, p.hw, p.silu);
  return i2v_check_launch("i2v_groupnorm_bwd_f16");
}

extern "C" int i2v_layernorm_f16(const i2v_ln_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_layernorm_f16: null params");
  const i2v_ln_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.y, "i2v_layernorm_f16: null pointer");
  I2V_CHECK_ARG(p.rows > 0 && p.C > 0 && p.C % 8 == 0 && p.C <= 4096,
                "i2v_layernorm_f16: C (%d) must be a multiple of 8 and <= 4096", p.C);
  I2V_CHECK_ARG(p.ldx % 8 == 0 && p.ldy % 8 == 0 && p.ldx >= p.C && p.ldy >= p.C, "i2v_layernorm_f16: bad ldx / ldy");
  I2V_CHECK_ARG(al16(p.x) && al16(p.y) && al16(p.gamma) && al16(p.beta), "i2v_layernorm_f16: 16-byte alignment");
  if (p.pe) I2V_CHECK_ARG(p.pe_period > 0 && p.ld_pe % 8 == 0 && al16(p.pe), "i2v_layernorm_f16: bad pe arguments");
  I2V_CHECK_ARG(p.x_rows_per_batch >= 0 && (p.x_rows_per_batch == 0 || (p.x_batch_stride % 8 == 0 && p.x_batch_stride > 0)),
                "i2v_layernorm_f16: x_batch_stride must be a positive multiple of 8 elements");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 block(256);
  const int nv = (int)i2v_cdiv(p.C / 8, 64);
  // rows per wave: 4 for C <= 512, 2 up to 1536, then 1 (the rows must fit the register file)
  switch (nv) {
    case 1: hipLaunchKernelGGL((ln_kernel<1, 4>), dim3((unsigned)i2v_cdiv(p.rows, 16)), block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((ln_kernel<2, 2>), dim3((unsigned)i2v_cdiv(p.rows, 8)), block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((ln_kernel<3, 2>), dim3((unsigned)i2v_cdiv(p.rows, 8)), block, 0, s, p); break;
    case 4: hipLaunchKernelGGL((ln_kernel<4, 1>), dim3((unsigned)i2v_cdiv(p.rows, 4)), block, 0, s, p); break;
    case 5: hipLaunchKernelGGL((ln_kernel<5, 1>), dim3((unsigned)i2v_cdiv(p.rows, 4)), block, 0, s, p); break;
    case 6: hipLaunchKernelGGL((ln_kernel<6, 1>), dim3((unsigned)i2v_cdiv(p.rows, 4)), block, 0, s, p); break;
    case 7: hipLaunchKernelGGL((ln_kernel<7, 1>), dim3((unsigned)i2v_cdiv(p.rows, 4)), block, 0, s, p); break;
    default: hipLaunchKernelGGL((ln_kernel<8, 1>), dim3((unsigned)i2v_cdiv(p.rows, 4)), block, 0, s, p); break;
  }
  return i2v_check_launch("i2v_layernorm_f16");
}

extern "C" int i2v_softmax_rows_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols,
                                    float scale, i2v_stream_t stream) {
  I2V_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= cols, "i2v_softmax_rows_f16: bad arguments");
  I2V_CHECK_ARG(scale > 0.f, "i2v_softmax_rows_f16: scale must be positive");
  I2V_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && al16(x) && al16(y), "i2v_softmax_rows_f16: rows must be 16-byte aligned");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)i2v_cdiv(rows, 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f16*>(x), ldx,
                     reinterpret_cast<f16*>(y), ldy, rows, cols, scale * 1.4426950408889634f);
  return i2v_check_launch("i2v_softmax_rows_f16");
}
