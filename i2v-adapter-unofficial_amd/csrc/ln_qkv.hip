// LayerNorm 1 and the projections in front of the spatial block's self- / cross-frame attention (i2v:444-445, 468-473, 483-492)
//     n = LayerNorm(x);   [q | k | q_adapter] = n W_qkq^T   (row-major, 640 or 960 columns);   V^T[image][channel][key] = (n W_v^T)^T
// in ONE launch at the 64^2 level of SD-1.5 (C = 320).  Un-fused these were `131072 x 960 x 320 +ln` (142 us: 1536 tiles of five
// K tiles each, 45 % of a tile's time outside its K loop -- 569 TFLOP/s and 2.4 TB/s, bound by neither) and `131072 x 320 x 320
// +ln, V^T store` (57 us): 199 us for 107 GFLOP, x read four times.
//
// Structure: the persistent 128-row-tile form of motion_attn.hip.  8 waves own 128 rows at a time; their LayerNorm-ed rows sit in
// LDS once (80 KB, 16-byte chunks XOR-swizzled by the row), a second panel takes the next tile's rows by LDS-DMA under the last
// pass.  Wave w owns the 160 output columns 160 w .. 160 w + 159 of [q | k | q_adapter | v] -- ten 16-column tiles, no padding:
// waves 0-5 the 960 columns of q | k | q_adapter, waves 6-7 the 320 of v (without the adapter's q: 0-3 and 4-5, two waves idle) --
// in three passes of 4, 4 and 2 tiles (a pass's accumulators are 128 registers; all ten at once would be 320), and streams only
// its own rows of the weights, in fragment order, straight from L2 into registers; a panel fragment read feeds 4 MFMAs.
//   * q / k / q_adapter groups are projected TRANSPOSED (D[channel][row] = W tile x n^T): a lane holds 4 consecutive channels of
//     one row, v_permlane16_swap pairs neighbouring lane groups into 16-byte stores, a wave writes 160 contiguous bytes per row;
//   * v groups are projected the other way round (D[row][channel] = n x W tile^T): a lane holds 4 consecutive KEYS of one channel,
//     and the same lane pairing gives 16-byte stores into V^T[channel][key] -- 64 contiguous bytes per channel and instruction.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void lq_dma16(__amdgpu_buffer_rsrc_t rsrc, f16* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// (a lane index the compiler cannot see through: address arithmetic derived from it is redone where it is used instead of being
//  hoisted out of the tile loop and spilled -- a reload from scratch is followed by `s_waitcnt vmcnt(0)`, see ff_fused.hip)
__device__ __forceinline__ int lq_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr int LQ_PIX = 8;              // 16-row MFMA tiles per workgroup tile (128 rows)
constexpr int LQ_TPW = 10;             // 16-column tiles per wave (160 columns)
constexpr int LQ_DT = 4;               // ... of which a pass takes 4, 4, 2
constexpr int LQ_PD = 2;               // weight fragment sets in flight beyond the one in use (K steps ahead)
constexpr int LQ_AD = 4;               // panel fragments in flight

template <int C, int H>
__global__ __launch_bounds__(64 * H) void ln_qkv_kernel(const i2v_ln_qkv_params p, const int ntiles) {
  constexpr int KS = C / 32, NJ = C / 64, DT = LQ_DT;
  static_assert(LQ_TPW == 10 && C % 64 == 0 && H == 8, "8 lanes x C / 64 chunks per row in the LayerNorm pass");
  extern __shared__ __attribute__((aligned(16))) f16 panels[];       // 2 x [128][C], chunk index ^= row & 7
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15, sub = lane & 7;
  const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x);
  const float* __restrict__ gamma = reinterpret_cast<const float*>(p.gamma);
  const float* __restrict__ beta = reinterpret_cast<const float*>(p.beta);
  const int qk_waves = p.n_qk / (16 * LQ_TPW), live_waves = (p.n_qk + p.channels) / (16 * LQ_TPW);     // 6 / 8 (4 / 6)

  // ---- LayerNorm of the wave's 16 rows into a panel (as motion_attn.hip: raw rows by LDS-DMA into the wave's own 10 KB, then
  // normalised in place)
  auto fetch_rows = [&](const int tile, f16* panel) {
    int64_t r0 = (int64_t)tile * (LQ_PIX * 16) + 16 * wave;            // (images are whole tiles: a wave's 16 rows are in one)
    const f16* base = X + r0 * p.ldx;
    if (p.x_image_stride > 0) {                                         // images at their own stride (frame-0 rows in place)
      const int64_t img = r0 / p.rows_per_image;
      base = X + img * p.x_image_stride + (r0 - img * p.rows_per_image) * p.ldx;
    }
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)((15 * p.ldx + C) * 2), 0x00020000);
    const int ln = lq_opaque(lane);
    const unsigned voff = (unsigned)(((ln >> 3) * p.ldx + (ln & 7) * 8) * 2);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        lq_dma16(rs, panel + 16 * wave * C + (half * NJ + j) * 512, voff, (unsigned)((8 * half * p.ldx + 8 * j * 8) * 2));
  };
  auto normalise_rows = [&](f16* panel) {
    f16x8 xv[2][NJ];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j) xv[half][j] = *reinterpret_cast<const f16x8*>(panel + 16 * wave * C + ((half * NJ + j) * 64 + lane) * 8);
    const float* gp = gamma;
    const float* bp = beta;
    asm volatile("" : "+s"(gp), "+s"(bp));          // (loop invariants: keep their 80 registers out of the passes)
    f32x4 ga[NJ][2], be[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      ga[j][0] = *reinterpret_cast<const f32x4*>(gp + (sub + 8 * j) * 8);
      ga[j][1] = *reinterpret_cast<const f32x4*>(gp + (sub + 8 * j) * 8 + 4);
      be[j][0] = *reinterpret_cast<const f32x4*>(bp + (sub + 8 * j) * 8);
      be[j][1] = *reinterpret_cast<const f32x4*>(bp + (sub + 8 * j) * 8 + 4);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 16 * wave + 8 * half + (lane >> 3);
      float v[NJ][8];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[j][e] = (float)xv[half][j][e];
          s += v[j][e];
        }
      s = sum_lanes8(s);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[j][e] -= mean;
          q = fmaf(v[j][e], v[j][e], q);
        }
      q = sum_lanes8(q);
      const float rstd = rsqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = sub + 8 * j;
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)fmaf(v[j][e] * rstd, ga[j][e >> 2][e & 3], be[j][e >> 2][e & 3]);
        *reinterpret_cast<f16x8*>(panel + row * C + ((ch ^ (row & 7)) * 8)) = o;
      }
    }
  };

  // ---- NTL tiles of 16 output columns (the wave's tiles t0 .. t0 + NTL - 1) against the 128 rows of the panel;
  // weights [wave][tile of the wave][K step][lane][8]
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, live_waves * (LQ_TPW * 16 * C * 2), 0x00020000);
  const int w_lane = lane * 16;
  const int sw = l15 & 7;
  const int swz[2] = {(g ^ sw) * 8, ((4 + g) ^ sw) * 8};
  f32x4 acc[LQ_PIX][DT];
  auto project = [&](const f16* panel, const int t0, auto ntl, auto transposed) {
    constexpr bool TR = decltype(transposed)::value;
    constexpr int NTL = decltype(ntl)::value;
    const int wp = (wave * LQ_TPW + t0) * (16 * C * 2);       // bytes; + t (16 C 2) + s 1024
    auto ldw = [&](const int s, const int t) {
      return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, wp + t * (16 * C * 2) + s * 1024, 0));
    };
    const f16* alane = panel + l15 * C;
    f16x8 wf[LQ_PD + 1][NTL];
#pragma unroll
    for (int pix = 0; pix < LQ_PIX; ++pix)
#pragma unroll
      for (int t = 0; t < NTL; ++t) acc[pix][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < LQ_PD; ++s)
#pragma unroll
      for (int t = 0; t < NTL; ++t) wf[s][t] = ldw(s, t);
    constexpr int AD = LQ_AD, NI = KS * LQ_PIX;
    // (chunk 4 s + g swizzled by the row: ((4 s + g) ^ sw) = 8 (s >> 1) + ((4 (s & 1) + g) ^ sw) -- TWO lane-dependent offsets and an
    //  immediate, not one precomputed register per K step: those ten were hoisted out of the tile loop and spilled)
    auto lda = [&](const int i) {
      const int s = i / LQ_PIX;
      return *reinterpret_cast<const f16x8*>(alane + 16 * (i % LQ_PIX) * C + 64 * (s >> 1) + swz[s & 1]);
    };
    f16x8 af[AD + 1];
#pragma unroll
    for (int i = 0; i < AD; ++i) af[i] = lda(i);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int s = i / LQ_PIX, pix = i % LQ_PIX;
      if (pix == 0 && s + LQ_PD < KS) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) wf[(s + LQ_PD) % (LQ_PD + 1)][t] = ldw(s + LQ_PD, t);
      }
      if (i + AD < NI) af[(i + AD) % (AD + 1)] = lda(i + AD);
#pragma unroll
      for (int t = 0; t < NTL; ++t)
        acc[pix][t] = TR ? mfma16x16x32(wf[s % (LQ_PD + 1)][t], af[i % (AD + 1)], acc[pix][t])
                         : mfma16x16x32(af[i % (AD + 1)], wf[s % (LQ_PD + 1)][t], acc[pix][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto to_half = [](const f32x4 a) { return __builtin_bit_cast(u32x2, f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]}); };
  // v_permlane16_swap: the odd 16-lane rows of the first operand change places with the even rows of the second
  auto swap16 = [](u32x2& a, u32x2& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]));
  };

  // (r5, measured and dropped -- the kernel without its stores runs in 108 us, with them in 166 - 171, whatever their schedule:
  //  the two waves of a SIMD taking their passes in different orders (172 us); a pass's results held back as fp16 and stored one
  //  instruction per few MFMA steps inside the NEXT pass, in passes of 3, 3, 2, 2 tiles (176 - 190 us); workgroups of an XCD started
  //  up to one pass apart (175 - 185 us).  gfx9 counts loads and stores on ONE counter, so a wait for a weight fragment is a
  //  wait for every store issued before it; spreading the stores spreads the waiting.)
  int tile = blockIdx.x;
  if (tile >= ntiles) return;        // (workgroup-uniform)
  fetch_rows(tile, panels);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  normalise_rows(panels);
  lds_barrier();
  using T4 = std::integral_constant<int, 4>;
  using T2 = std::integral_constant<int, 2>;
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    const f16* panel = panels + (it & 1) * (LQ_PIX * 16 * C);
    f16* other = panels + ((it + 1) & 1) * (LQ_PIX * 16 * C);
    const int next = tile + (int)gridDim.x;
#ifdef LQ_SMALLOUT
    const int64_t row0 = 0;      // (timing experiment: every tile stores to the first tile's rows -- the stores stay in L2)
#else
    const int64_t row0 = (int64_t)tile * (LQ_PIX * 16);
#endif
    // q | k | q_adapter tiles: D[channel][row]; lane: channels 16 t + 4 g .. + 3 of row l15.  v_permlane16_swap pairs the lane
    // groups: even g ends with tile t channels 4 g .. + 7, odd g with tile t + 1 channels 4 (g - 1) .. + 7 -- one 16-byte store
    auto store_qk = [&](const int t0, auto ntl) {
      constexpr int NTL = decltype(ntl)::value;
      static_assert(NTL % 2 == 0, "tiles are stored in pairs");
      const int sln = lq_opaque(lane), sg = sln >> 4, sl15 = sln & 15;
      f16* __restrict__ O = reinterpret_cast<f16*>(p.qk) + (row0 + sl15) * p.ld_qk + (wave * LQ_TPW + t0) * 16;
#pragma unroll
      for (int pix = 0; pix < LQ_PIX; ++pix) {
        f16* orow = O + (int64_t)(16 * pix) * p.ld_qk;
#pragma unroll
        for (int t = 0; t + 1 < NTL; t += 2) {
          u32x2 a = to_half(acc[pix][t]), b = to_half(acc[pix][t + 1]);
          swap16(a, b);
#ifdef LQ_NOSTORE
          if (p.ld_qk < 0)
#endif
          *reinterpret_cast<u32x4*>(orow + (t + (sg & 1)) * 16 + 4 * (sg & ~1)) = u32x4{a[0], a[1], b[0], b[1]};
        }
      }
    };
    // v tiles: D[row][channel]; lane: keys (rows) 4 g .. + 3 of channel 16 t + l15 -> V^T[image][channel][key]; the same pairing
    // over two 16-row steps: even g ends with keys 4 g .. + 7 of step pix, odd g with keys 4 (g - 1) .. + 7 of step pix + 1
    auto store_vt = [&](const int t0, auto ntl) {
      constexpr int NTL = decltype(ntl)::value;
      const int sln = lq_opaque(lane), sg = sln >> 4, sl15 = sln & 15;
      const int64_t img = row0 / p.rows_per_image;
      const int key0 = (int)(row0 - img * p.rows_per_image);
      f16* __restrict__ VT = reinterpret_cast<f16*>(p.vt) + img * p.vt_batch_stride + key0 +
                             ((int64_t)((wave - qk_waves) * LQ_TPW + t0) * 16 + sl15) * p.vt_row_stride;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        f16* vrow = VT + (int64_t)(16 * t) * p.vt_row_stride;
#pragma unroll
        for (int pix = 0; pix < LQ_PIX; pix += 2) {
          u32x2 a = to_half(acc[pix][t]), b = to_half(acc[pix + 1][t]);
          swap16(a, b);
#ifdef LQ_NOSTORE
          if (p.ld_qk < 0)
#endif
          *reinterpret_cast<u32x4*>(vrow + 16 * (pix + (sg & 1)) + 4 * (sg & ~1)) = u32x4{a[0], a[1], b[0], b[1]};
        }
      }
    };
    // the wave's ten tiles in three passes of 4, 4 and 2.  The next tile's rows leave HBM in front of the last pass and land (in the
    // other panel, which nobody reads any more) under it; no wait of their own: loads return in order, and that pass waits for
    // weight fragments it requested after them
    auto run = [&](auto transposed) {
      auto pass = [&](const int t0, auto ntl) {
        project(panel, t0, ntl, transposed);
        if constexpr (decltype(transposed)::value) store_qk(t0, ntl); else store_vt(t0, ntl);
      };
      pass(0, T4{});
      pass(4, T4{});
      if (next < ntiles) fetch_rows(next, other);
      __builtin_amdgcn_sched_barrier(0);
      pass(8, T2{});
    };
    if (wave < qk_waves) {                          // (wave-uniform)
      run(std::true_type{});
    } else if (wave < live_waves) {
      run(std::false_type{});
    } else {                                        // (without the adapter's query two waves only normalise their rows)
      if (next < ntiles) fetch_rows(next, other);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (next < ntiles) normalise_rows(other);
    lds_barrier();       // (LDS only: the stores stay in flight across it)
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

constexpr size_t LQ_LDS = 2 * (size_t)LQ_PIX * 16 * 320 * sizeof(f16);
int lq_cus() { return i2v_big_lds_kernel_cus(reinterpret_cast<const void*>(ln_qkv_kernel<320, 8>), LQ_LDS); }

}  // namespace

extern "C" int32_t i2v_ln_qkv_supported(int64_t rows, int32_t channels, int32_t n_qk, int64_t rows_per_image) {
  return rows > 0 && rows % (LQ_PIX * 16) == 0 && rows / (LQ_PIX * 16) < (1 << 24) && channels == 320 &&
         (n_qk == channels || n_qk == 2 * channels || n_qk == 3 * channels) && rows_per_image > 0 && rows_per_image % (LQ_PIX * 16) == 0 &&
         rows % rows_per_image == 0 && lq_cus() > 0;
}

extern "C" int i2v_ln_qkv_f16(const i2v_ln_qkv_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_ln_qkv_f16: null params");
  const i2v_ln_qkv_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.w && p.qk && p.vt, "i2v_ln_qkv_f16: null pointer");
  I2V_CHECK_ARG(i2v_ln_qkv_supported(p.rows, p.channels, p.n_qk, p.rows_per_image),
                "i2v_ln_qkv_f16: rows %lld channels %d n_qk %d rows_per_image %lld is not a fused shape (i2v_ln_qkv_supported)",
                (long long)p.rows, p.channels, p.n_qk, (long long)p.rows_per_image);
  I2V_CHECK_ARG(p.ldx >= p.channels && p.ldx % 8 == 0 && p.ldx < (1 << 24) && p.ld_qk >= p.n_qk && p.ld_qk % 8 == 0 &&
                    p.vt_row_stride >= p.rows_per_image && p.vt_row_stride % 8 == 0 &&
                    p.vt_batch_stride >= (int64_t)p.channels * p.vt_row_stride && p.vt_batch_stride % 8 == 0 &&
                    (p.x_image_stride == 0 || (p.x_image_stride >= p.rows_per_image * p.ldx && p.x_image_stride % 8 == 0)),
                "i2v_ln_qkv_f16: strides");
  I2V_CHECK_ARG(al16(p.x) && al16(p.gamma) && al16(p.beta) && al16(p.w) && al16(p.qk) && al16(p.vt),
                "i2v_ln_qkv_f16: pointers must be 16-byte aligned");
  const int cus = lq_cus();
  if (cus <= 0) I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_ln_qkv_f16: %zu bytes of LDS refused by this device", LQ_LDS);
  const int ntiles = (int)(p.rows / (LQ_PIX * 16));
  const int grid = i2v_persistent_grid(ntiles, cus);
  hipLaunchKernelGGL((ln_qkv_kernel<320, 8>), dim3((unsigned)grid), dim3(512), LQ_LDS, reinterpret_cast<hipStream_t>(stream), p, ntiles);
  return i2v_check_launch("i2v_ln_qkv_f16");
}
