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

template <int BM, int AMODE>
__global__ __launch_bounds__(512) void gemm_big_kernel(const i2v_gemm_params p, const int tiles_n, const int vec4) {
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
      c_tap[i] = a_k[i] / p.cin;
      c_ci[i] = a_k[i] - c_tap[i] * p.cin;
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

  const int nkt = (K + 63) / 64;
  issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
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
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = mfma16x16x32(wf[i], af[j], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all fragment reads of stage `cur` done before it is refilled
  }

  // ---------------------------------------------------------------- epilogue (lane: row m, 4 consecutive n)
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * WM + j * 16 + l15;
    if (m >= M) continue;
    const GemmRow row = gemm_make_row(p, m);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      gemm_store4(p, vec4, row, n0 + wn * 80 + i * 16 + g * 4, v);
    }
  }
}

template <int BM>
int launch_big(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = p.N / BIG_BN;
  const dim3 grid(tiles_m * tiles_n), block(512);
  if (p.a_mode == I2V_A_CONV3X3)
    hipLaunchKernelGGL((gemm_big_kernel<BM, I2V_A_CONV3X3>), grid, block, 0, s, p, tiles_n, vec4);
  else
    hipLaunchKernelGGL((gemm_big_kernel<BM, I2V_A_PLAIN>), grid, block, 0, s, p, tiles_n, vec4);
  const int rc = i2v_check_launch("i2v_gemm_f16(big)");
  return rc < 0 ? rc : 1;
}

}  // namespace

int i2v_gemm_big_try(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  static const int mode = getenv("I2V_GEMM_BIG") ? atoi(getenv("I2V_GEMM_BIG")) : -1;  // 0 off, 256 / 128 force
  if (mode == 0) return 0;
  if (p.N % BIG_BN != 0) return 0;
  const int64_t tn = p.N / BIG_BN;
  const int64_t t256 = i2v_cdiv(p.M, 256) * tn, t128 = i2v_cdiv(p.M, 128) * tn;
  if (mode == 256) return launch_big<256>(p, vec4, s);
  if (mode == 128) return launch_big<128>(p, vec4, s);
  // measured (profiles/r1_tile_sweep.txt, "big" columns): with one 8-wave block per CU and one K tile of DMA in
  // flight the kernel needs a long K loop to amortise its prologue / epilogue; short-K plain GEMMs (K <= 640) are
  // latency-bound and run faster as 3 small blocks per CU in gemm.hip.  The im2col conv always has K >= 9 * cin.
  if (p.a_mode != I2V_A_CONV3X3 && p.K < 1280) return 0;
  if (t256 >= 192) return launch_big<256>(p, vec4, s);   // >= 0.75 wave of 256-row tiles over the 256 CUs
  if (t128 >= 128) return launch_big<128>(p, vec4, s);
  return 0;
}
