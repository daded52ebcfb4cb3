// Pieces shared by the GEMM kernels: LDS swizzle and the fused epilogue.
#pragma once
#include "common.h"

// LDS tile rows are 128 B (64 halfs) = 8 chunks of 16 B; chunk c of row r lives at chunk c ^ ((r >> 1) & 7):
// every 16-lane group of a ds_read_b128 MFMA-fragment read then hits 16 distinct 16-byte slots of the 256-byte
// bank row (conflict-free; guide T2).
__device__ __forceinline__ int gemm_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct GemmRow {
  int m;
  int64_t m_out;
  const f16* rv;  // time-embedding row vector (already offset to the row's vector)
  const f16* rs;  // residual row
  const f16* rl;  // ... and its low half (i2v_gemm_params.residual_lo: the precise residual stream), or null
  f16* cl;        // low half of the output row (c_lo), or null
};

// the precise residual stream (i2v_gemm_params.residual_lo / c_lo): a value v as the fp16 pair (hi, lo), hi = fp16(v) the tensor every
// MFMA operand reads, lo = fp16(v - hi) the bits that rounding dropped (|lo| <= ulp(hi) / 2: exact to 2^-22 |v|, 6e-8 absolute)
__device__ __forceinline__ f16 lo_half(const float v, const f16 hi) { return (f16)(v - (float)hi); }

__device__ __forceinline__ GemmRow gemm_make_row(const i2v_gemm_params& p, int m) {
  GemmRow r;
  r.m = m;
  r.m_out = m;
  if (p.store_mode == I2V_STORE_ROWPERM) {
    const int per = p.hw * p.frames;
    const int b = m / per, rem = m - b * per;
    const int pix = rem / p.frames, f = rem - pix * p.frames;
    r.m_out = (int64_t)(b * p.frames + f) * p.hw + pix;
  }
  const f16* rowvec = reinterpret_cast<const f16*>(p.rowvec);
  const f16* resid = reinterpret_cast<const f16*>(p.residual);
  r.rv = rowvec ? rowvec + (int64_t)(p.rowvec_period > 0 ? (m & (p.rowvec_period - 1)) : m / p.rows_per_vec) * p.ld_rowvec
                : nullptr;
  r.rs = resid ? resid + r.m_out * p.ldr : nullptr;
  r.rl = (resid && p.residual_lo) ? reinterpret_cast<const f16*>(p.residual_lo) + r.m_out * p.ldr : nullptr;
  r.cl = p.c_lo ? reinterpret_cast<f16*>(p.c_lo) + r.m_out * p.ldc : nullptr;
  return r;
}

// v[0..3] = accumulators of row `row.m`, columns n .. n + 3 (n % 4 == 0).
__device__ __forceinline__ void gemm_store4(const i2v_gemm_params& p, const int vec4, const GemmRow& row, const int n,
                                            float v[4]) {
  const int N = p.N;
  if (n >= N) return;
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
  f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
  const float oscale = p.out_scale;
  if (vec4) {
    if (bias) {
      const f16x4 b4 = *reinterpret_cast<const f16x4*>(bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
    }
    if (row.rv) {
      const f16x4 t4 = *reinterpret_cast<const f16x4*>(row.rv + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)t4[r];
    }
    if (row.rs) {
      const f16x4 r4 = *reinterpret_cast<const f16x4*>(row.rs + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
      if (row.rl) {
        const f16x4 l4 = *reinterpret_cast<const f16x4*>(row.rl + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)l4[r];
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < N) {
        if (bias) v[r] += (float)bias[n + r];
        if (row.rv) v[r] += (float)row.rv[n + r];
        if (row.rs) v[r] += (float)row.rs[n + r];
        if (row.rl) v[r] += (float)row.rl[n + r];
      }
    }
  }
  if (p.epilogue == I2V_EPI_GEGLU) {
    // rows of W interleaved (value, gate): (v0, v1) and (v2, v3) are (value, gate) pairs
    const float o0 = v[0] * gelu_erf(v[1]) * oscale;
    const float o1 = v[2] * gelu_erf(v[3]) * oscale;
    f16* dst = C + row.m_out * p.ldc + (n >> 1);
    if (vec4) {
      f16x2 o = {(f16)o0, (f16)o1};
      *reinterpret_cast<f16x2*>(dst) = o;
    } else {
      if (n + 1 < N) dst[0] = (f16)o0;
      if (n + 3 < N) dst[1] = (f16)o1;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (p.epilogue == I2V_EPI_GELU) v[r] = gelu_erf(v[r]);
    v[r] *= oscale;
  }
  if (p.store_mode == I2V_STORE_VT) {
    // element (m, n) -> ((n / L) * M + m) * ld + n % L
    if (vec4) {
      const int bt = n / p.vt_len, kk = n - bt * p.vt_len;
      f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(C + ((int64_t)bt * p.M + row.m) * p.vt_ld + kk) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < N) {
          const int bt = (n + r) / p.vt_len, kk = (n + r) - bt * p.vt_len;
          C[((int64_t)bt * p.M + row.m) * p.vt_ld + kk] = (f16)v[r];
        }
      }
    }
  } else {
    f16* dst = C + row.m_out * p.ldc + n;
    if (vec4) {
      f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(dst) = o;
      if (row.cl) *reinterpret_cast<f16x4*>(row.cl + n) = f16x4{lo_half(v[0], o[0]), lo_half(v[1], o[1]), lo_half(v[2], o[2]), lo_half(v[3], o[3])};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) {
          dst[r] = (f16)v[r];
          if (row.cl) row.cl[n + r] = lo_half(v[r], (f16)v[r]);
        }
    }
  }
}

// implemented in gemm_big.hip: 256-thread-pair (8-wave) LDS-DMA kernel for N % 320 == 0; returns 1 if it took the
// problem, 0 if the caller should use the generic kernel, < 0 on error.
int i2v_gemm_big_try(const i2v_gemm_params& p, int vec4, hipStream_t s);
int i2v_gemm_big_gn_rows(const i2v_gemm_params& p, int vec4);
// conv_thin.hip: 3x3 convolutions with <= 16 output channels (same return convention)
int i2v_conv_thin_try(const i2v_gemm_params& p, hipStream_t s);
// 1 if gemm_big.hip takes this problem AND implements the LayerNorm fold (ln_wsum) for its epilogue
int i2v_gemm_big_ln_ok(const i2v_gemm_params& p, int vec4);
int i2v_gemm_big_unsplit_ok(const i2v_gemm_params& p, int vec4);
// fp32 scratch bytes with which gemm_big.hip would split K for this problem (0: no split)
int64_t i2v_gemm_big_workspace_bytes(const i2v_gemm_params& p, int vec4);
// implemented in gemm_ws.hip: the weight-stationary kernel for K = 320 row-major projections with >= 16384 rows (W slices in
// registers, A streamed through three LDS stages); same return convention as i2v_gemm_big_try
int i2v_gemm_ws_try(const i2v_gemm_params& p, hipStream_t s);
int i2v_gemm_ws_ok(const i2v_gemm_params& p);
