// MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogues for gfx950 (MI355X).
//
//   C[m, n] = epi(sum_k A[m, k] W[n, k] + bias[n] + rowvec[m / rpv, n] + residual[m', n]) * out_scale
//
// Both operands are K-contiguous ("NT" form: torch Linear weights are [N, K]), so both MFMA fragments are
// 16-byte LDS reads.  Tile BM x BN x 64, 256 threads = 4 waves (2 x 2), v_mfma_f32_16x16x32_f16.
// The MFMA is issued as D = Wfrag * Afrag, i.e. the accumulator holds C^T tiles: each lane owns 4 CONSECUTIVE
// n for one m, which makes bias / residual / output accesses 8-byte vectors and lets the GEGLU gate pair
// (value, gate interleaved rows of W) sit in one lane.
// Staging: global -> registers (issued before the MFMA phase of the current tile) -> XOR-swizzled LDS
// (written after it), two LDS stages, one barrier per K-tile (guide T14 / T2).
// A-operand loaders: plain row-major (optionally two K-ranges = skip-connection concat without a cat copy),
// or im2col-on-the-fly over an NHWC image for 3x3 / stride 1|2 / nearest-2x-upsample convolutions.
#include <cstdlib>

#include "gemm_common.h"

namespace {

constexpr int BK = 64;  // halfs per K tile = 128 B per LDS row = 8 chunks of 16 B

template <int BM, int BN>
struct Smem {
  f16 a[2][BM * BK];
  f16 w[2][BN * BK];
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct ConvRow {
  int pix_base;  // image index * in_h * in_w
  int oy, ox;
  bool valid;
};

template <int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const i2v_gemm_params p, const int tiles_n, const int vec4) {
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
  constexpr int AR = BM / 32, WR = BN / 32;  // 16-byte chunks per thread per tile
  __shared__ __attribute__((aligned(16))) Smem<BM, BN> sm;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int M = p.M, N = p.N, K = p.K;

  const f16* __restrict__ A = reinterpret_cast<const f16*>(p.a);
  const f16* __restrict__ A2 = reinterpret_cast<const f16*>(p.a2);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);
  const int ksp = (A2 != nullptr) ? p.k_split : K;

  // ---- staging assignment: thread -> chunk column cc, rows r0 + 32 i
  const int cc = tid & 7, r0 = tid >> 3;

  ConvRow crow[AR];
  int64_t arow_off[AR];
  bool arow_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + r0 + 32 * i;
    arow_ok[i] = m < M;
    if (AMODE == I2V_A_CONV3X3) {
      const int ohw = p.out_h * p.out_w;
      const int mm = arow_ok[i] ? m : 0;
      const int img = mm / ohw, rem = mm % ohw;
      crow[i].pix_base = img * p.in_h * p.in_w;
      crow[i].oy = rem / p.out_w;
      crow[i].ox = rem % p.out_w;
      crow[i].valid = arow_ok[i];
      arow_off[i] = 0;
    } else {
      arow_off[i] = (int64_t)m;
    }
  }
  int64_t wrow_off[WR];
  bool wrow_ok[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + r0 + 32 * i;
    wrow_ok[i] = n < N;
    wrow_off[i] = (int64_t)n * p.ldw;
  }

  f16x8 ra[AR], rw[WR];

  auto load_tile = [&](int kt) {
    const int k = kt * BK + cc * 8;
    const bool kok = k < K;
    if (AMODE == I2V_A_CONV3X3) {
      int tap, ci;
      if (p.conv_kblock) {   // k = ((ci / 64) * 9 + tap) * 64 + ci % 64
        const int blk = k >> 6;
        tap = blk % 9;
        ci = (blk / 9) * 64 + (k & 63);
      } else {
        tap = k / p.cin;
        ci = k - tap * p.cin;
      }
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        f16x8 v = zero8();
        if (kok && crow[i].valid) {
          int iy, ix;
          bool ok;
          if (p.upsample) {
            const int uy = crow[i].oy - 1 + dy, ux = crow[i].ox - 1 + dx;
            ok = (uy >= 0) && (ux >= 0) && (uy < p.out_h) && (ux < p.out_w);      // (out = 2 in, or 2 in - 1: forward_upsample_size)
            iy = uy >> 1;
            ix = ux >> 1;
          } else {
            iy = crow[i].oy * p.stride - (p.asym_pad ? 0 : 1) + dy;
            ix = crow[i].ox * p.stride - (p.asym_pad ? 0 : 1) + dx;
            ok = (iy >= 0) && (ix >= 0) && (iy < p.in_h) && (ix < p.in_w);
          }
          if (ok) v = ld_global_16B(A + (int64_t)(crow[i].pix_base + iy * p.in_w + ix) * p.lda + ci);
        }
        ra[i] = v;
      }
    } else {
      const bool first = k < ksp;
      const f16* src = first ? A : A2;
      const int64_t ld = first ? p.lda : p.lda2;
      const int kk = first ? k : k - ksp;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        f16x8 v = zero8();
        if (kok && arow_ok[i]) v = ld_global_16B(src + arow_off[i] * ld + kk);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      f16x8 v = zero8();
      if (kok && wrow_ok[i]) v = ld_global_16B(W + wrow_off[i] + k);
      rw[i] = v;
    }
  };

  auto store_tile = [&](int stage) {
    char* sa = reinterpret_cast<char*>(sm.a[stage]);
    char* sw = reinterpret_cast<char*>(sm.w[stage]);
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f16x8*>(sa + swz(r0 + 32 * i, cc)) = ra[i];
#pragma unroll
    for (int i = 0; i < WR; ++i) *reinterpret_cast<f16x8*>(sw + swz(r0 + 32 * i, cc)) = rw[i];
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nkt;
    if (more) load_tile(kt + 1);

    const char* sa = reinterpret_cast<const char*>(sm.a[cur]);
    const char* sw = reinterpret_cast<const char*>(sm.w[cur]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 wf[NI], af[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = wn * WN + i * 16 + l15;
        wf[i] = *reinterpret_cast<const f16x8*>(sw + swz(row, ks * 4 + g));
      }
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int row = wm * WM + j * 16 + l15;
        af[j] = *reinterpret_cast<const f16x8*>(sa + swz(row, ks * 4 + g));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = mfma16x16x32(wf[i], af[j], acc[i][j]);
    }

    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
  const f16* __restrict__ resid = reinterpret_cast<const f16*>(p.residual);
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(p.rowvec);
  f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
  const float oscale = p.out_scale;

#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * WM + j * 16 + l15;
    if (m >= M) continue;
    int64_t m_out = m;
    if (p.store_mode == I2V_STORE_ROWPERM) {
      const int per = p.hw * p.frames;
      const int b = m / per, rem = m - b * per;
      const int pix = rem / p.frames, f = rem - pix * p.frames;
      m_out = (int64_t)(b * p.frames + f) * p.hw + pix;
    }
    const f16* rv = rowvec ? rowvec + (int64_t)(p.rowvec_period > 0 ? (m & (p.rowvec_period - 1)) : m / p.rows_per_vec) * p.ld_rowvec
                           : nullptr;
    const f16* rs = resid ? resid + m_out * p.ldr : nullptr;
    // the precise residual stream (i2v_gemm_params.residual_lo / c_lo)
    const f16* rl = (resid && p.residual_lo) ? reinterpret_cast<const f16*>(p.residual_lo) + m_out * p.ldr : nullptr;
    f16* cl = p.c_lo ? reinterpret_cast<f16*>(p.c_lo) + m_out * p.ldc : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + wn * WN + i * 16 + g * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (vec4) {
        if (bias) {
          const f16x4 b4 = *reinterpret_cast<const f16x4*>(bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
        }
        if (rv) {
          const f16x4 t4 = *reinterpret_cast<const f16x4*>(rv + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)t4[r];
        }
        if (rs) {
          const f16x4 r4 = *reinterpret_cast<const f16x4*>(rs + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
          if (rl) {
            const f16x4 l4 = *reinterpret_cast<const f16x4*>(rl + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)l4[r];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < N) {
            if (bias) v[r] += (float)bias[n + r];
            if (rv) v[r] += (float)rv[n + r];
            if (rs) v[r] += (float)rs[n + r];
            if (rl) v[r] += (float)rl[n + r];
          }
        }
      }
      if (p.epilogue == I2V_EPI_GEGLU) {
        // rows of W interleaved (value, gate): (v0, v1) and (v2, v3) are (value, gate) pairs
        const float o0 = v[0] * gelu_erf(v[1]) * oscale;
        const float o1 = v[2] * gelu_erf(v[3]) * oscale;
        f16* dst = C + m_out * p.ldc + (n >> 1);
        if (vec4) {
          f16x2 o = {(f16)o0, (f16)o1};
          *reinterpret_cast<f16x2*>(dst) = o;
        } else {
          if (n + 1 < N) dst[0] = (f16)o0;
          if (n + 3 < N) dst[1] = (f16)o1;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (p.epilogue == I2V_EPI_GELU) v[r] = gelu_erf(v[r]);
        v[r] *= oscale;
      }
      if (p.store_mode == I2V_STORE_VT_T) {
        // element (m, n) -> ((m / L) * N + n) * ld + m % L   (scalar form; the vector form lives in gemm_big.hip)
        const int bt = m / p.vt_len, kk = m - bt * p.vt_len;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) C[((int64_t)bt * N + (n + r)) * p.vt_ld + kk] = (f16)v[r];
      } else if (p.store_mode == I2V_STORE_VT) {
        // element (m, n) -> ((n / L) * M + m) * ld + n % L
        if (vec4) {
          const int bt = n / p.vt_len, kk = n - bt * p.vt_len;
          f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          *reinterpret_cast<f16x4*>(C + ((int64_t)bt * M + m) * p.vt_ld + kk) = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (n + r < N) {
              const int bt = (n + r) / p.vt_len, kk = (n + r) - bt * p.vt_len;
              C[((int64_t)bt * M + m) * p.vt_ld + kk] = (f16)v[r];
            }
          }
        }
      } else if (p.c_is_f32) {
        float* dst = reinterpret_cast<float*>(p.c) + m_out * p.ldc + n;
        if (vec4) {
          *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) dst[r] = v[r];
        }
      } else {
        f16* dst = C + m_out * p.ldc + n;
        if (vec4) {
          f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          *reinterpret_cast<f16x4*>(dst) = o;
          if (cl) *reinterpret_cast<f16x4*>(cl + n) = f16x4{lo_half(v[0], o[0]), lo_half(v[1], o[1]), lo_half(v[2], o[2]), lo_half(v[3], o[3])};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) {
              dst[r] = (f16)v[r];
              if (cl) cl[n + r] = lo_half(v[r], (f16)v[r]);
            }
        }
      }
    }
  }
}

struct TileCfg {
  int bm, bn, blocks_per_cu;
  float eff;  // relative MFMA efficiency of the tile shape (operand reuse)
};

template <int BM, int BN>
int launch(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = (int)i2v_cdiv(p.N, BN);
  const dim3 grid(tiles_m * tiles_n), block(256);
  if (p.a_mode == I2V_A_CONV3X3)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, I2V_A_CONV3X3>), grid, block, 0, s, p, tiles_n, vec4);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, I2V_A_PLAIN>), grid, block, 0, s, p, tiles_n, vec4);
  return i2v_check_launch("i2v_gemm_f16");
}

inline bool aligned_to(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// 8-byte vector epilogue is legal when every lane group of 4 consecutive n is complete and aligned
int vector_epilogue_ok(const i2v_gemm_params& p) {
  int vec4 = (p.N % 4 == 0) ? 1 : 0;
  if (p.bias && !aligned_to(p.bias, 8)) vec4 = 0;
  if (p.residual && (!aligned_to(p.residual, 8) || p.ldr % 4 != 0)) vec4 = 0;
  if ((p.residual_lo && !aligned_to(p.residual_lo, 8)) || (p.c_lo && !aligned_to(p.c_lo, 8))) vec4 = 0;
  if (p.rowvec && (!aligned_to(p.rowvec, 8) || p.ld_rowvec % 4 != 0)) vec4 = 0;
  if (p.store_mode == I2V_STORE_VT) {
    if (p.vt_len % 4 != 0 || p.vt_ld % 4 != 0 || !aligned_to(p.c, 8)) vec4 = 0;
  } else if (p.store_mode == I2V_STORE_VT_T) {
    // here `vec4` means "4 consecutive m (keys) form an aligned 8-byte store" -- only gemm_big.hip uses it
    vec4 = (p.vt_len % 4 == 0 && p.vt_ld % 4 == 0 && aligned_to(p.c, 8) && p.M % 4 == 0) ? 1 : 0;
  } else if (p.epilogue == I2V_EPI_GEGLU) {
    if (p.ldc % 2 != 0 || !aligned_to(p.c, 4)) vec4 = 0;
  } else if (p.c_is_f32) {
    if (p.ldc % 4 != 0 || !aligned_to(p.c, 16)) vec4 = 0;
  } else {
    if (p.ldc % 4 != 0 || !aligned_to(p.c, 8)) vec4 = 0;
  }
  return vec4;
}

}  // namespace

extern "C" int64_t i2v_gemm_workspace_bytes(const i2v_gemm_params* pp) {
  if (pp == nullptr || pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
  return i2v_gemm_big_workspace_bytes(*pp, vector_epilogue_ok(*pp));
}

extern "C" int i2v_gemm_ln_supported(const i2v_gemm_params* pp) {
  if (pp == nullptr || pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
  return i2v_gemm_big_ln_ok(*pp, vector_epilogue_ok(*pp));
}

extern "C" int i2v_gemm_batch_supported(const i2v_gemm_params* pp) {
  if (pp == nullptr || pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
  return i2v_gemm_big_unsplit_ok(*pp, vector_epilogue_ok(*pp));
}

extern "C" int32_t i2v_gemm_gn_partial_rows(const i2v_gemm_params* pp) {
  if (pp == nullptr) return 0;
  return i2v_gemm_big_gn_rows(*pp, vector_epilogue_ok(*pp));
}

extern "C" int i2v_gemm_f16(const i2v_gemm_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_gemm_f16: null params");
  const i2v_gemm_params& p = *pp;
  I2V_CHECK_ARG(p.a && p.w && p.c, "i2v_gemm_f16: null a/w/c pointer");
  I2V_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "i2v_gemm_f16: M, N, K must be positive (got %d, %d, %d)", p.M, p.N,
                p.K);
  I2V_CHECK_ARG(p.K % 8 == 0, "i2v_gemm_f16: K (%d) must be a multiple of 8", p.K);
  I2V_CHECK_ARG(p.ldw % 8 == 0 && p.ldw >= p.K, "i2v_gemm_f16: ldw (%lld) must be >= K and a multiple of 8",
                (long long)p.ldw);
  I2V_CHECK_ARG(aligned_to(p.a, 16) && aligned_to(p.w, 16), "i2v_gemm_f16: a / w must be 16-byte aligned");
  I2V_CHECK_ARG(p.lda % 8 == 0, "i2v_gemm_f16: lda (%lld) must be a multiple of 8", (long long)p.lda);
  if (p.a_mode == I2V_A_CONV3X3) {
    I2V_CHECK_ARG(p.a2 == nullptr, "i2v_gemm_f16: conv mode takes a single source");
    I2V_CHECK_ARG(p.cin > 0 && p.cin % 8 == 0 && p.K == 9 * p.cin, "i2v_gemm_f16: conv needs cin %% 8 == 0, K == 9 cin");
    I2V_CHECK_ARG(p.stride == 1 || p.stride == 2, "i2v_gemm_f16: conv stride must be 1 or 2");
    I2V_CHECK_ARG(!(p.upsample && p.stride != 1), "i2v_gemm_f16: upsample conv must have stride 1");
    I2V_CHECK_ARG(!p.asym_pad || (p.stride == 2 && !p.upsample), "i2v_gemm_f16: asym_pad needs stride 2, no upsample");
    I2V_CHECK_ARG(p.conv_kblock == 0 || (p.conv_kblock == 64 && p.cin % 64 == 0),
                  "i2v_gemm_f16: conv_kblock must be 0 or 64 (with cin %% 64 == 0), got %d for cin %d", p.conv_kblock, p.cin);
    const int padsum = p.asym_pad ? 1 : 2;
    const int eh = p.upsample ? 2 * p.in_h : (p.in_h + padsum - 3) / p.stride + 1;
    const int ew = p.upsample ? 2 * p.in_w : (p.in_w + padsum - 3) / p.stride + 1;
    // upsample: nearest to 2 in, or to 2 in - 1 (unet:1304-1311, 1414-1415 `forward_upsample_size`: the skip tensor of a level whose
    // size was odd before its stride-2 down-sampler; F.interpolate(size=2 in - 1, mode="nearest") reads source floor(i in / (2 in - 1))
    // = i >> 1, the same gather, with the zero padding at the smaller image's border)
    I2V_CHECK_ARG((p.out_h == eh || (p.upsample && p.out_h == eh - 1)) && (p.out_w == ew || (p.upsample && p.out_w == ew - 1)),
                  "i2v_gemm_f16: conv output size mismatch (%d x %d vs %d x %d)", p.out_h, p.out_w, eh, ew);
    I2V_CHECK_ARG((int64_t)p.n_img * p.out_h * p.out_w == p.M, "i2v_gemm_f16: conv M != n_img*out_h*out_w");
    I2V_CHECK_ARG(p.lda >= p.cin, "i2v_gemm_f16: conv pixel stride lda < cin");
    I2V_CHECK_ARG((int64_t)p.n_img * p.in_h * p.in_w < (1ll << 31), "i2v_gemm_f16: conv image too large");
  } else {
    I2V_CHECK_ARG(p.a_mode == I2V_A_PLAIN, "i2v_gemm_f16: bad a_mode %d", p.a_mode);
    if (p.a2) {
      I2V_CHECK_ARG(p.k_split > 0 && p.k_split < p.K && p.k_split % 8 == 0 && p.lda2 % 8 == 0 && aligned_to(p.a2, 16),
                    "i2v_gemm_f16: bad dual-source split (k_split %d, K %d)", p.k_split, p.K);
      I2V_CHECK_ARG(p.lda >= p.k_split && p.lda2 >= p.K - p.k_split, "i2v_gemm_f16: dual-source ld too small");
    } else {
      I2V_CHECK_ARG(p.lda >= p.K, "i2v_gemm_f16: lda (%lld) < K (%d)", (long long)p.lda, p.K);
    }
  }
  I2V_CHECK_ARG(p.epilogue >= I2V_EPI_NONE && p.epilogue <= I2V_EPI_GEGLU, "i2v_gemm_f16: bad epilogue");
  I2V_CHECK_ARG(p.store_mode >= I2V_STORE_ROWMAJOR && p.store_mode <= I2V_STORE_VT_T, "i2v_gemm_f16: bad store_mode");
  if (p.epilogue == I2V_EPI_GEGLU)
    I2V_CHECK_ARG(p.N % 2 == 0 && p.store_mode != I2V_STORE_VT && p.store_mode != I2V_STORE_VT_T,
                  "i2v_gemm_f16: GEGLU needs even N, non-VT store");
  if (p.store_mode == I2V_STORE_VT_T) {
    I2V_CHECK_ARG(p.vt_len > 0 && p.vt_ld >= p.vt_len && p.M % p.vt_len == 0,
                  "i2v_gemm_f16: VT_T store needs M %% vt_len == 0 and vt_ld >= vt_len");
    I2V_CHECK_ARG(p.residual == nullptr && (p.rowvec == nullptr || (p.rowvec_period > 0 && p.ln_wsum)),
                  "i2v_gemm_f16: VT_T store takes no residual, and a rowvec only as the transposed positional table of a "
                  "LayerNorm-folded projection");
  }
  if (p.c_is_f32)
    I2V_CHECK_ARG(p.c_is_f32 == 1 && p.epilogue != I2V_EPI_GEGLU && p.store_mode == I2V_STORE_ROWMAJOR && p.ln_wsum == nullptr &&
                      p.rows_per_w == 0 && p.a_perm_frames == 0 && p.N <= 64 && aligned_to(p.c, 4),
                  "i2v_gemm_f16: an fp32 result (c_is_f32) is a row-major store of a narrow (N <= 64), plain problem");
  if (p.residual_lo || p.c_lo) {
    I2V_CHECK_ARG(p.residual_lo == nullptr || p.residual != nullptr, "i2v_gemm_f16: residual_lo needs residual");
    I2V_CHECK_ARG(p.epilogue == I2V_EPI_NONE && (p.store_mode == I2V_STORE_ROWMAJOR || p.store_mode == I2V_STORE_ROWPERM) && !p.c_is_f32 &&
                      p.ln_wsum == nullptr && p.gn_partial == nullptr,
                  "i2v_gemm_f16: the precise residual stream (residual_lo / c_lo) takes plain row-major / row-permuted fp16 stores only");
  }
  if (p.rowvec) I2V_CHECK_ARG(p.rows_per_vec > 0 || p.rowvec_period > 0, "i2v_gemm_f16: rows_per_vec must be positive");
  if (p.store_mode == I2V_STORE_ROWPERM) {
    I2V_CHECK_ARG(p.frames > 0 && p.hw > 0 && p.M % (p.frames * p.hw) == 0,
                  "i2v_gemm_f16: ROWPERM needs M %% (frames*hw) == 0");
    I2V_CHECK_ARG(p.rowvec == nullptr, "i2v_gemm_f16: ROWPERM does not take rowvec");
  }
  if (p.store_mode == I2V_STORE_VT) {
    I2V_CHECK_ARG(p.vt_len > 0 && p.vt_ld >= p.vt_len && p.N % p.vt_len == 0,
                  "i2v_gemm_f16: VT store needs N %% vt_len == 0 and vt_ld >= vt_len");
    I2V_CHECK_ARG(p.residual == nullptr && p.rowvec == nullptr, "i2v_gemm_f16: VT store takes no residual/rowvec");
  }

  int vec4 = vector_epilogue_ok(p);
  if (p.rowvec && p.rowvec_period > 0)
    I2V_CHECK_ARG(p.store_mode != I2V_STORE_ROWPERM && p.store_mode != I2V_STORE_VT &&
                      (p.rowvec_period & (p.rowvec_period - 1)) == 0,
                  "i2v_gemm_f16: a periodic rowvec (positional table) needs a power-of-two period (the epilogue masks the "
                  "row index: an integer modulo there cost the 256-row kernels 40 %%) and a row-major / VT_T store");
  if (p.ln_wsum) {
    I2V_CHECK_ARG(aligned_to(p.ln_wsum, 16) && p.ln_eps > 0.f, "i2v_gemm_f16: ln_wsum must be 16-byte aligned, ln_eps > 0");
    if (!i2v_gemm_big_ln_ok(p, vec4))
      I2V_FAIL(I2V_ERR_INVALID_ARG, "i2v_gemm_f16: LayerNorm-folded GEMM is not implemented for this problem "
               "(M %d N %d K %d, epilogue %d, store %d): ask i2v_gemm_ln_supported() first", p.M, p.N, p.K, p.epilogue,
               p.store_mode);
  }

  if (p.gn_partial) {
    I2V_CHECK_ARG(aligned_to(p.gn_partial, 8), "i2v_gemm_f16: gn_partial must be 8-byte aligned");
    if (!i2v_gemm_big_gn_rows(p, vec4))
      I2V_FAIL(I2V_ERR_INVALID_ARG, "i2v_gemm_f16: GroupNorm partials (gn_partial) are not implemented for this problem (M %d N %d K %d, "
               "groups %d): ask i2v_gemm_gn_partial_rows() first", p.M, p.N, p.K, p.gn_groups);
  }
  if (p.rows_per_w > 0 || p.a_perm_frames > 0) {
    I2V_CHECK_ARG(p.rows_per_w >= 0 && p.a_perm_frames >= 0 && p.w_batch_stride >= 0, "i2v_gemm_f16: negative batch / permutation field");
    if (!i2v_gemm_big_unsplit_ok(p, vec4))
      I2V_FAIL(I2V_ERR_INVALID_ARG, "i2v_gemm_f16: per-batch weights / the permuted A gather are not implemented for this "
               "problem (M %d N %d K %d, rows_per_w %d, a_perm %d x %d): they need the 8-wave kernel with full row tiles",
               p.M, p.N, p.K, p.rows_per_w, p.a_perm_frames, p.a_perm_hw);
  }
#ifdef I2V_VARIANTS
  // the K = 320 projections of the 64 x 64 level: weight-stationary kernel (variants/gemm_ws.hip; opt-in, I2V_GEMM_WS=1)
  {
    const int ws = i2v_gemm_ws_try(p, reinterpret_cast<hipStream_t>(stream));
    if (ws != 0) return ws < 0 ? ws : I2V_OK;
  }
#endif
  // narrow-output 3x3 convolutions (conv_out: 4 channels): a halo-tile kernel of their own (conv_thin.hip)
  {
    const int thin = i2v_conv_thin_try(p, reinterpret_cast<hipStream_t>(stream));
    if (thin != 0) return thin < 0 ? thin : I2V_OK;
  }
  // large problems whose N is a multiple of 320 go to the 8-wave LDS-DMA kernel (gemm_big.hip)
  {
    const int big = i2v_gemm_big_try(p, vec4, reinterpret_cast<hipStream_t>(stream));
    if (big != 0) return big < 0 ? big : I2V_OK;
  }
  if (p.store_mode == I2V_STORE_VT_T) vec4 = 0;   // the generic kernel stores this mode element by element
  // tile selection: modelled time = waves of tiles over 256 CUs x tile area / shape efficiency
  // `eff` is measured relative throughput per tile area on MI355X at the UNet's shapes (tools/kernel_bench.py sweep,
  // profiles/r1_tile_sweep.txt): the loop is latency-bound, so the smaller tiles with 3 blocks / CU win except
  // for very long K; the im2col loader favours a short M tile (fewer gathered rows per block).
  const bool conv = p.a_mode == I2V_A_CONV3X3;
  const bool long_k = p.K >= 2048;
  const TileCfg cfgs[4] = {{128, 128, 2, long_k ? 1.15f : 1.00f},
                           {128, 64, 3, conv ? 1.00f : 1.30f},
                           {64, 128, 3, conv ? 1.25f : 1.25f},
                           {64, 64, 5, 1.10f}};
  int best = 0;
  double best_t = 1e300;
  for (int i = 0; i < 4; ++i) {
    const double tiles = (double)i2v_cdiv(p.M, cfgs[i].bm) * (double)i2v_cdiv(p.N, cfgs[i].bn);
    const double slots = 256.0 * cfgs[i].blocks_per_cu;
    const double rounds = tiles <= slots ? 1.0 : tiles / slots;  // partial last round amortised when large
    const double conc = tiles < slots ? tiles : slots;
    // per-round time ~ tile area / eff, divided by how busy the CUs are (at most blocks_per_cu blocks share a CU)
    const double per_cu = (conc / 256.0 < 1.0) ? 1.0 : conc / 256.0;
    const double t = rounds * per_cu * cfgs[i].bm * cfgs[i].bn / cfgs[i].eff;
    if (t < best_t) {
      best_t = t;
      best = i;
    }
  }
  static const int tile_env = getenv("I2V_GEMM_TILE") ? atoi(getenv("I2V_GEMM_TILE")) : -1;  // tuning override
  if (tile_env >= 0 && tile_env < 4) best = tile_env;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (best) {
    case 0: return launch<128, 128>(p, vec4, s);
    case 1: return launch<128, 64>(p, vec4, s);
    case 2: return launch<64, 128>(p, vec4, s);
    default: return launch<64, 64>(p, vec4, s);
  }
}
