// Large-tile MFMA GEMM / implicit-GEMM conv for gfx950: BM x 320 x 64 tiles, 8 waves (2 x 4, per-wave (BM/2) x 80),
// operands staged by LDS-DMA (global_load_lds_dwordx4: global -> LDS with no VGPR / ds_write pass), two LDS stages,
// the next K tile's DMA in flight under the current tile's MFMAs behind a counted vmcnt and raw s_barriers.
//
// Why this shape (measured in round 1, profiles/r1_tile_sweep.txt): the 128-wide register-staged kernel of gemm.hip is
// bound by the LDS write path (ds_write_b128 ~79 B/clk/CU) and by L2->CU operand bandwidth (43-65 FLOP/B per tile vs
// the ~95 B/clk a CU's MFMA pipes consume); a 256 x 320 tile needs 146 FLOP per staged byte and LDS-DMA removes the
// ds_write pass.  Every channel count of the SD-1.5 topology is a multiple of 320 (320/640/960/1280/2560/5120/10240),
// so BN = 320 = 4 waves x 5 MFMA columns tiles N with no padding.
// LDS image: rows of 128 B, chunk c of row r at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments).  LDS-DMA
// writes lane-linearly (wave base + 16 * lane), so the swizzle is applied on the SOURCE side: lane l of the
// instruction for 8-row group j fetches logical chunk (l & 7) ^ ((row >> 1) & 7) of row 8 j + (l >> 3) (guide rule 21).
// Masked lanes (row >= M, k >= K, conv halo) fetch from a zero page instead, because LDS-DMA cannot skip a lane.
#include <cstdlib>
#include <utility>

#include "gemm_common.h"

namespace {

__device__ __attribute__((aligned(64))) f16 g_zero_page[64];  // zero-initialised: source for masked LDS-DMA lanes

constexpr int BIG_BN = 320;

__device__ __forceinline__ void glds16(const f16* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// SPLIT: blockIdx.y selects a range of `kps` K tiles; the fp32 accumulators go to p.workspace[split][M][N] and
// splitk_reduce_kernel applies the epilogue (small-M / long-K problems that cannot fill the chip with output tiles).
template <int BM, int AMODE, int EPI, int STORE, bool SPLIT = false>
__global__ __launch_bounds__(512) void gemm_big_kernel(const i2v_gemm_params p, const int tiles_n, const int kps) {
  constexpr int BN = BIG_BN;
  constexpr int WM = BM / 2, MI = WM / 16, NI = 5;
  constexpr int AG = BM / 64;  // 8-row (1 KiB) A groups per wave: (BM / 8) groups over 8 waves
  constexpr int WG = 5;        // W groups per wave: 40 over 8 waves
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, l15 = lane & 15;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int M = p.M, N = p.N, K = p.K;

  const f16* __restrict__ A = reinterpret_cast<const f16*>(p.a);
  const f16* __restrict__ A2 = reinterpret_cast<const f16*>(p.a2);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);
  const int ksp = (A2 != nullptr) ? p.k_split : K;
  const f16* zero = g_zero_page;

  // ---- per-lane DMA source description: lane -> (row of the 8-row group, logical 16-byte chunk)
  const int lr = lane >> 3, lc = lane & 7;
  int a_k[AG];          // k offset (halfs) of this lane's chunk inside a K tile
  bool a_ok[AG];
  int64_t a_off[AG], a_off2[AG];   // plain: row offsets in the two sources
  int c_pix[AG], c_oy[AG], c_ox[AG], c_tap[AG], c_ci[AG];   // conv: output pixel + running (tap, channel) of the chunk
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int r = 8 * (wave + 8 * i) + lr;
    const int clog = lc ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_k[i] = clog * 8;
    a_ok[i] = m < M;
    if (AMODE == I2V_A_CONV3X3) {
      const int ohw = p.out_h * p.out_w;
      const int mm = a_ok[i] ? m : 0;
      const int img = mm / ohw, rem = mm - img * ohw;
      c_pix[i] = img * p.in_h * p.in_w;
      c_oy[i] = rem / p.out_w;
      c_ox[i] = rem - c_oy[i] * p.out_w;
      const int k_first = a_k[i] + (SPLIT ? (int)blockIdx.y * kps * 64 : 0);
      c_tap[i] = k_first / p.cin;
      c_ci[i] = k_first - c_tap[i] * p.cin;
      a_off[i] = a_off2[i] = 0;
    } else {
      a_off[i] = (int64_t)m * p.lda;
      a_off2[i] = (int64_t)m * p.lda2;
      c_pix[i] = c_oy[i] = c_ox[i] = c_tap[i] = c_ci[i] = 0;
    }
  }
  int w_k[WG];
  bool w_ok[WG];
  int64_t w_off[WG];
#pragma unroll
  for (int i = 0; i < WG; ++i) {
    const int r = 8 * (wave + 8 * i) + lr;
    const int clog = lc ^ ((r >> 1) & 7);
    const int n = n0 + r;
    w_k[i] = clog * 8;
    w_ok[i] = n < N;
    w_off[i] = (int64_t)n * p.ldw;
  }

  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sw = sa + BM * 128;
    const int kb = kt * 64;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const f16* src = zero;
      if (AMODE == I2V_A_CONV3X3) {
        // (tap, ci) of this lane's chunk were advanced incrementally; tap >= 9 <=> k >= K
        if (a_ok[i] && c_tap[i] < 9) {
          const int dy = c_tap[i] / 3, dx = c_tap[i] - dy * 3;
          int iy, ix;
          bool ok;
          if (p.upsample) {
            const int uy = c_oy[i] - 1 + dy, ux = c_ox[i] - 1 + dx;
            ok = (uy >= 0) && (ux >= 0) && (uy < 2 * p.in_h) && (ux < 2 * p.in_w);
            iy = uy >> 1;
            ix = ux >> 1;
          } else {
            iy = c_oy[i] * p.stride - 1 + dy;
            ix = c_ox[i] * p.stride - 1 + dx;
            ok = (iy >= 0) && (ix >= 0) && (iy < p.in_h) && (ix < p.in_w);
          }
          if (ok) src = A + (int64_t)(c_pix[i] + iy * p.in_w + ix) * p.lda + c_ci[i];
        }
        c_ci[i] += 64;
        while (c_ci[i] >= p.cin) {
          c_ci[i] -= p.cin;
          c_tap[i] += 1;
        }
      } else {
        const int k = kb + a_k[i];
        if (a_ok[i] && k < K) src = (k < ksp) ? (A + a_off[i] + k) : (A2 + a_off2[i] + (k - ksp));
      }
      glds16(src, sa + (wave + 8 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < WG; ++i) {
      const int k = kb + w_k[i];
      const f16* src = (w_ok[i] && k < K) ? (W + w_off[i] + k) : zero;
      glds16(src, sw + (wave + 8 * i) * 1024);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt_all = (K + 63) / 64;
  const int kt0 = SPLIT ? (int)blockIdx.y * kps : 0;
  const int nkt = SPLIT ? min(nkt_all, kt0 + kps) : nkt_all;
  issue(kt0, 0);
  for (int kt = kt0; kt < nkt; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < nkt) {
      issue(kt + 1, cur ^ 1);       // stage cur^1 was last read in iteration kt-1, closed by its trailing barrier
      wait_vmcnt<AG + WG>();        // everything older than the tile just issued (= tile kt) has landed
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();   // every wave's share of tile kt is in LDS

    const char* sa = smem + cur * STAGE;
    const char* sw = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 wf[NI], af[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sw + gemm_swz(wn * 80 + i * 16 + l15, ks * 4 + g));
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = *reinterpret_cast<const f16x8*>(sa + gemm_swz(wm * WM + j * 16 + l15, ks * 4 + g));
      // D = W_frag * A_frag: lane owns 4 consecutive n of one row m.  For the transposed V^T store the operands are
      // swapped (D = A_frag * W_frag): lane owns 4 consecutive m (keys) of one channel n = an 8-byte run of a V^T row.
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = (STORE == I2V_STORE_VT_T) ? mfma16x16x32(af[j], wf[i], acc[i][j])
                                                : mfma16x16x32(wf[i], af[j], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all fragment reads of stage `cur` done before it is refilled
  }

  // ---------------------------------------------------------------- epilogue (lane: row m, 4 consecutive n)
  // The (epilogue, store mode) pair is a template parameter and the accumulator indices are compile-time constants
  // (static_for): with the generic runtime-switched store the 8 x 5 loop was not unrolled, the 160 accumulators went
  // through scratch memory and every tile paid ~50 us for it.
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
  const f16* __restrict__ resid = reinterpret_cast<const f16*>(p.residual);
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(p.rowvec);
  f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
  const float oscale = p.out_scale;
  if (SPLIT) {
    float* __restrict__ ws = reinterpret_cast<float*>(p.workspace) + (int64_t)blockIdx.y * M * N;
    static_for<MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int m = m0 + wm * WM + j * 16 + l15;
      if (m < M) {
        static_for<NI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          *reinterpret_cast<f32x4*>(ws + (int64_t)m * N + n0 + wn * 80 + i * 16 + g * 4) = acc[i][j];
        });
      }
    });
    return;
  }
  if (STORE == I2V_STORE_VT_T) {
    // accumulator rows = m (4 g + r), column = n (l15): element (m, n) -> C[((m / L) * N + n) * ld + m % L]
    static_for<NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int n = n0 + wn * 80 + i * 16 + l15;
      const float bn = bias ? (float)bias[n] : 0.f;
      static_for<MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = m0 + wm * WM + j * 16 + 4 * g;
        if (m < M) {   // M % 4 == 0 and vt_len % 4 == 0: the 4 keys are in range and in one batch
          const int bt = m / p.vt_len, kk = m - bt * p.vt_len;
          const f16x4 o4 = {(f16)((acc[i][j][0] + bn) * oscale), (f16)((acc[i][j][1] + bn) * oscale),
                            (f16)((acc[i][j][2] + bn) * oscale), (f16)((acc[i][j][3] + bn) * oscale)};
          *reinterpret_cast<f16x4*>(C + ((int64_t)bt * N + n) * p.vt_ld + kk) = o4;
        }
      });
    });
    return;
  }
  const int ncol0 = n0 + wn * 80 + g * 4;
  f16x4 b4[NI];
  static_for<NI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    b4[i] = bias ? *reinterpret_cast<const f16x4*>(bias + ncol0 + i * 16) : f16x4{0, 0, 0, 0};
  });
  static_for<MI>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int m = m0 + wm * WM + j * 16 + l15;
    if (m < M) {
      int64_t m_out = m;
      if (STORE == I2V_STORE_ROWPERM) {
        const int per = p.hw * p.frames;
        const int b = m / per, rem = m - b * per;
        const int pix = rem / p.frames, f = rem - pix * p.frames;
        m_out = (int64_t)(b * p.frames + f) * p.hw + pix;
      }
      const f16* rv = rowvec ? rowvec + (int64_t)(m / p.rows_per_vec) * p.ld_rowvec : nullptr;
      const f16* rs = resid ? resid + m_out * p.ldr : nullptr;
      static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int n = ncol0 + i * 16;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + (float)b4[i][r];
        if (rv) {
          const f16x4 t4 = *reinterpret_cast<const f16x4*>(rv + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)t4[r];
        }
        if (rs) {
          const f16x4 r4 = *reinterpret_cast<const f16x4*>(rs + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
        }
        if (EPI == I2V_EPI_GEGLU) {
          const f16x2 o2 = {(f16)(v[0] * gelu_erf(v[1]) * oscale), (f16)(v[2] * gelu_erf(v[3]) * oscale)};
          *reinterpret_cast<f16x2*>(C + m_out * p.ldc + (n >> 1)) = o2;
        } else {
          const f16x4 o4 = {(f16)(v[0] * oscale), (f16)(v[1] * oscale), (f16)(v[2] * oscale), (f16)(v[3] * oscale)};
          if (STORE == I2V_STORE_VT) {
            const int bt = n / p.vt_len, kk = n - bt * p.vt_len;
            *reinterpret_cast<f16x4*>(C + ((int64_t)bt * M + m) * p.vt_ld + kk) = o4;
          } else {
            *reinterpret_cast<f16x4*>(C + m_out * p.ldc + n) = o4;
          }
        }
      });
    }
  });
}

template <int BM, int AMODE>
int launch_big_mode(const i2v_gemm_params& p, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = p.N / BIG_BN;
  const dim3 grid(tiles_m * tiles_n), block(512);
  if (p.epilogue == I2V_EPI_GEGLU) {
    if constexpr (AMODE == I2V_A_PLAIN)
      hipLaunchKernelGGL((gemm_big_kernel<BM, AMODE, I2V_EPI_GEGLU, I2V_STORE_ROWMAJOR>), grid, block, 0, s, p, tiles_n, 0);
  } else if (p.store_mode == I2V_STORE_ROWPERM) {
    if constexpr (AMODE == I2V_A_PLAIN)
      hipLaunchKernelGGL((gemm_big_kernel<BM, AMODE, I2V_EPI_NONE, I2V_STORE_ROWPERM>), grid, block, 0, s, p, tiles_n, 0);
  } else if (p.store_mode == I2V_STORE_VT) {
    if constexpr (AMODE == I2V_A_PLAIN)
      hipLaunchKernelGGL((gemm_big_kernel<BM, AMODE, I2V_EPI_NONE, I2V_STORE_VT>), grid, block, 0, s, p, tiles_n, 0);
  } else if (p.store_mode == I2V_STORE_VT_T) {
    if constexpr (AMODE == I2V_A_PLAIN)
      hipLaunchKernelGGL((gemm_big_kernel<BM, AMODE, I2V_EPI_NONE, I2V_STORE_VT_T>), grid, block, 0, s, p, tiles_n, 0);
  } else {
    hipLaunchKernelGGL((gemm_big_kernel<BM, AMODE, I2V_EPI_NONE, I2V_STORE_ROWMAJOR>), grid, block, 0, s, p, tiles_n, 0);
  }
  const int rc = i2v_check_launch("i2v_gemm_f16(big)");
  return rc < 0 ? rc : 1;
}

template <int BM>
int launch_big(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  (void)vec4;
  if (p.a_mode == I2V_A_CONV3X3) return launch_big_mode<BM, I2V_A_CONV3X3>(p, s);
  return launch_big_mode<BM, I2V_A_PLAIN>(p, s);
}

// ---- split-K: sum the fp32 partial tiles and apply the fused epilogue (bias / time-embedding vector / residual)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const i2v_gemm_params p, const int splits, const int vec4) {
  const int64_t groups = (int64_t)p.M * (p.N / 4);
  const float* __restrict__ ws = reinterpret_cast<const float*>(p.workspace);
  const int64_t slab = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < groups; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / (p.N / 4)), n = (int)(i - (int64_t)m * (p.N / 4)) * 4;
    f32x4 s4 = *reinterpret_cast<const f32x4*>(ws + (int64_t)m * p.N + n);
    for (int sidx = 1; sidx < splits; ++sidx) s4 += *reinterpret_cast<const f32x4*>(ws + sidx * slab + (int64_t)m * p.N + n);
    float v[4] = {s4[0], s4[1], s4[2], s4[3]};
    const GemmRow row = gemm_make_row(p, m);
    gemm_store4(p, vec4, row, n, v);
  }
}

// number of K splits the split path would use (0: not a split-K problem)
int splitk_plan(const i2v_gemm_params& p, int vec4, int* kps_out) {
  if (p.N % BIG_BN != 0 || !vec4 || p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR) return 0;
  const int nkt = (p.K + 63) / 64;
  const int64_t t128 = i2v_cdiv(p.M, 128) * (p.N / BIG_BN);
  // measured: K = 1280 (20 K tiles) loses to the unsplit generic kernel (the reduce pass costs more than it saves);
  // K >= 2560 gains 1.6 - 2.3x (8 x 8 level convs 316 -> 654 TFLOP/s)
  if (nkt < 40 || t128 > 128) return 0;
  int splits = (int)(256 / t128);
  if (splits > 8) splits = 8;
  if (splits > nkt / 8) splits = nkt / 8;
  if (splits < 2) return 0;
  const int kps = (int)i2v_cdiv(nkt, splits);
  splits = (int)i2v_cdiv(nkt, kps);          // no empty split
  if (kps_out) *kps_out = kps;
  return splits;
}

int launch_split(const i2v_gemm_params& p, int vec4, int splits, int kps, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, 128), tiles_n = p.N / BIG_BN;
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  if (p.a_mode == I2V_A_CONV3X3)
    hipLaunchKernelGGL((gemm_big_kernel<128, I2V_A_CONV3X3, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps);
  else
    hipLaunchKernelGGL((gemm_big_kernel<128, I2V_A_PLAIN, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps);
  const int64_t groups = (int64_t)p.M * (p.N / 4);
  const int blocks = (int)(i2v_cdiv(groups, 256) < 2048 ? i2v_cdiv(groups, 256) : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, splits, vec4);
  const int rc = i2v_check_launch("i2v_gemm_f16(split-K)");
  return rc < 0 ? rc : 1;
}

}  // namespace

int64_t i2v_gemm_big_workspace_bytes(const i2v_gemm_params& p, int vec4) {
  static const int off = getenv("I2V_GEMM_SPLITK") ? (atoi(getenv("I2V_GEMM_SPLITK")) == 0) : 0;
  if (off) return 0;
  const int splits = splitk_plan(p, vec4, nullptr);
  return splits ? (int64_t)splits * p.M * p.N * (int64_t)sizeof(float) : 0;
}

int i2v_gemm_big_try(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  static const int mode = getenv("I2V_GEMM_BIG") ? atoi(getenv("I2V_GEMM_BIG")) : -1;  // 0 off, 256 / 128 force
  if (mode == 0) return 0;
  if (p.N % BIG_BN != 0) return 0;
  // the specialised epilogue handles the vector (8-byte) forms only; anything else stays on the generic kernel
  if (!vec4 || p.epilogue == I2V_EPI_GELU) return 0;
  if (p.a_mode == I2V_A_CONV3X3 && (p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR)) return 0;
  if (p.epilogue == I2V_EPI_GEGLU && p.store_mode != I2V_STORE_ROWMAJOR) return 0;
  const int64_t tn = p.N / BIG_BN;
  const int64_t t256 = i2v_cdiv(p.M, 256) * tn, t128 = i2v_cdiv(p.M, 128) * tn;
  if (mode == 256) return launch_big<256>(p, vec4, s);
  if (mode == 128) return launch_big<128>(p, vec4, s);
  static const int min_k = getenv("I2V_GEMM_BIG_MINK") ? atoi(getenv("I2V_GEMM_BIG_MINK")) : 128;
  if (p.a_mode != I2V_A_CONV3X3 && p.K < min_k) return 0;   // a single K tile cannot hide its own DMA latency
  // one 8-wave block per CU: a tile count just above a multiple of 256 wastes most of the last round.  Pick the
  // tile height by (fill of the last round) x (measured relative rate: 256-row 1.0, 128-row 0.82,
  // profiles/r1_tile_sweep.txt); below 45 % the 3-blocks-per-CU kernel of gemm.hip is faster.
  const double e256 = (double)t256 / (double)(i2v_cdiv(t256, 256) * 256);
  const double e128 = 0.82 * (double)t128 / (double)(i2v_cdiv(t128, 256) * 256);
  if (e256 >= e128 && e256 >= 0.45) return launch_big<256>(p, vec4, s);
  if (e128 >= 0.45) return launch_big<128>(p, vec4, s);
  // too few output tiles for the chip: split K when the caller supplied the fp32 scratch
  int kps = 0;
  const int splits = splitk_plan(p, vec4, &kps);
  if (splits && p.workspace && p.workspace_bytes >= (int64_t)splits * p.M * p.N * (int64_t)sizeof(float) &&
      (reinterpret_cast<uintptr_t>(p.workspace) % 16) == 0)
    return launch_split(p, vec4, splits, kps, s);
  return 0;
}
