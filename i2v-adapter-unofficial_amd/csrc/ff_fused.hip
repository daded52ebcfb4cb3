// The GEGLU feed-forward of a transformer block as ONE kernel (i2v:539-561 / the temporal block's FeedForward):
//     out = x + W2 (value o gelu(gate)) + b2,   [value | gate] = LayerNorm(x) W1^T + b1
// At the 64^2 level this was `131072 x 2560 x 320 geglu +ln` (343 us) and `131072 x 320 x 1280 +res` (137 us): the 335 MB
// intermediate went out to HBM and came back.  Here it never leaves the CU.
//
// Structure (the persistent 128-row-tile form of motion_attn.hip): 8 waves own 128 rows at a time; their LayerNorm-ed rows sit
// in LDS (80 KB, XOR-swizzled chunks).  The inner dimension (1280) is walked in 20 chunks of 64 channels.  Per chunk
//   * FF1: wave w projects ONE 16-row tile of W1 -- 8 (value, gate) pairs, rows interleaved so that a pair sits in one lane of the
//     TRANSPOSED product D[inner channel][row] -- against all 128 rows (80 MFMAs), applies bias and GEGLU in registers and writes
//     its 8 x 128 slice of the chunk to a 16 KB LDS buffer (two of them: one barrier per chunk);
//   * FF2: wave w owns 40 output channels (48 with the padding rows of the packed W2): out^T[n][row] += W2 tile x h^T with the
//     chunk as B operand from LDS (48 MFMAs); the 96 accumulators stay in registers across the 20 chunks.
// Every weight fragment is streamed once per tile by exactly one wave, in fragment order (1 KB per load instruction).
// Epilogue: + b2 + x (the residual rows, re-read: they are L2-warm), 16-byte stores assembled by v_permlane16_swap.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void ff_dma16(__amdgpu_buffer_rsrc_t rsrc, f16* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

constexpr int FF_PIX = 8;              // 16-row tiles per workgroup tile (128 rows)
constexpr int FF_CH = 128;             // inner channels per chunk: 8 waves x 2 tiles x 8 (value, gate) pairs
constexpr int FF_U = 2;                // W1 tiles per wave and chunk: a panel fragment read feeds FF_U MFMAs (with one, FF1 ran at
                                       // 28 % of the MFMA rate: 357k of a workgroup's 704k cycles)
constexpr int FF_PD = 3;               // W1 fragments in flight (K steps ahead)
constexpr int FF_AD = 4;               // panel fragments in flight

template <int C, int INNER, int H>
__global__ __launch_bounds__(64 * H) void ff_fused_kernel(const i2v_ff_fused_params p, const int ntiles, long long* __restrict__ stamps) {
  constexpr int KS = C / 32, NJ = C / 64, NCH = INNER / FF_CH, DN = C / H, DT = (DN + 15) / 16;
  static_assert(C % 64 == 0 && H == 8 && INNER % FF_CH == 0 && DT == 3 && DN == 40, "SD-1.5 64^2 level: C = 320, inner = 1280");
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  f16* panel = lds;                                   // [128][C], 16-byte chunk index ^= row & 7
  f16* hbuf = lds + FF_PIX * 16 * C;                  // 2 x [128][128], 16-byte chunk index ^= row & 15
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15, sub = lane & 7, sw = l15 & 7;
  const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x);

  // ---- rows of a tile by LDS-DMA into the wave's own 10 KB of the panel, normalised in place (as motion_attn.hip)
  auto fetch_rows = [&](const int tile) {
    const f16* base = X + ((int64_t)tile * (FF_PIX * 16) + 16 * wave) * p.ldx;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)((15 * p.ldx + C) * 2), 0x00020000);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        ff_dma16(rs, panel + 16 * wave * C + (half * NJ + j) * 512, (unsigned)(((8 * half + (lane >> 3)) * p.ldx + (sub + 8 * j) * 8) * 2));
  };
  auto normalise_rows = [&]() {
    f16x8 xv[2][NJ];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j) xv[half][j] = *reinterpret_cast<const f16x8*>(panel + 16 * wave * C + ((half * NJ + j) * 64 + lane) * 8);
    const float* gp = reinterpret_cast<const float*>(p.gamma);
    const float* bp = reinterpret_cast<const float*>(p.beta);
    asm volatile("" : "+s"(gp), "+s"(bp));             // (loop invariants: keep their 80 registers out of the chunk loop)
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

  // weights: fragment order, one buffer descriptor each; the lane offset in one register, the fragment's place in the scalar offset
  //   W1: [chunk][wave][tile u][K step][64][8]     (16 rows = 8 (value, gate) pairs of inner channels 128 ch + 16 w + 8 u ..)
  //   W2: [wave][chunk][k step 4][tile 3][64][8]   (rows = the wave's 40 output channels padded to 48, k = the chunk's 128 channels)
  constexpr int KS2 = FF_CH / 32;
  const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, NCH * H * FF_U * KS * 1024, 0x00020000);
  const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, H * NCH * KS2 * DT * 1024, 0x00020000);
  const int w_lane = lane * 16;
  const float* __restrict__ b1 = reinterpret_cast<const float*>(p.b1);      // [chunk][wave][tile][16] fp32, interleaved like W1's rows
  const f16* alane = panel + l15 * C;
  auto to_half = [](const f32x4 a) { return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]}; };

#ifdef I2V_FF_STAMPS
  long long t_ln = 0, t_ff1 = 0, t_glu = 0, t_bar = 0, t_ff2 = 0, t_epi = 0, t0;
#define FF_T(acc) { const long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define FF_T(acc)
#endif
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef I2V_FF_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif
    fetch_rows(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    normalise_rows();
    __syncthreads();
    FF_T(t_ln);

    f32x4 acc2[FF_PIX][DT];
#pragma unroll
    for (int pix = 0; pix < FF_PIX; ++pix)
#pragma unroll
      for (int t = 0; t < DT; ++t) acc2[pix][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- FF1: this wave's FF_U tiles of W1 (8 pairs each) against all 128 rows: D[inner][row]
      const int w1o = (ch * H + wave) * (FF_U * KS * 1024);
      auto ldw1 = [&](const int u, const int s) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, w_lane, w1o + (u * KS + s) * 1024, 0));
      };
      const int w2o = (wave * NCH + ch) * (KS2 * DT * 1024);
      auto ldw2 = [&](const int ks, const int t) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, w_lane, w2o + (ks * DT + t) * 1024, 0));
      };
      f32x4 acc1[FF_U][FF_PIX];
#pragma unroll
      for (int u = 0; u < FF_U; ++u) {
        const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + ((ch * H + wave) * FF_U + u) * 16 + 4 * g);
#pragma unroll
        for (int pix = 0; pix < FF_PIX; ++pix) acc1[u][pix] = bias1;
      }
      f16x8 wf[FF_PD][FF_U];
#pragma unroll
      for (int s = 0; s < FF_PD - 1; ++s)
#pragma unroll
        for (int u = 0; u < FF_U; ++u) wf[s][u] = ldw1(u, s);
      constexpr int NI = KS * FF_PIX;
      auto lda = [&](const int i) {
        return *reinterpret_cast<const f16x8*>(alane + 16 * (i % FF_PIX) * C + (((4 * (i / FF_PIX) + g) ^ sw) * 8));
      };
      f16x8 af[FF_AD + 1];
#pragma unroll
      for (int i = 0; i < FF_AD; ++i) af[i] = lda(i);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int s = i / FF_PIX, pix = i % FF_PIX;
        if (pix == 0 && s + FF_PD - 1 < KS) {
#pragma unroll
          for (int u = 0; u < FF_U; ++u) wf[(s + FF_PD - 1) % FF_PD][u] = ldw1(u, s + FF_PD - 1);
        }
        if (i + FF_AD < NI) af[(i + FF_AD) % (FF_AD + 1)] = lda(i + FF_AD);
#pragma unroll
        for (int u = 0; u < FF_U; ++u) acc1[u][pix] = mfma16x16x32(wf[s % FF_PD][u], af[i % (FF_AD + 1)], acc1[u][pix]);
        __builtin_amdgcn_sched_barrier(0);
      }
      FF_T(t_ff1);
      // the first half of this chunk's W2 fragments is requested now and arrives under the GEGLU
      f16x8 w2f[2][DT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < DT; ++t) w2f[ks][t] = ldw2(ks, t);
      // GEGLU in the lane: rows 4 g + {0, 1} = (value, gate) of inner channel 16 w + 8 u + 2 g, rows {2, 3} of the next one
      f16* hb = hbuf + (ch & 1) * (FF_PIX * 16 * FF_CH);
#pragma unroll
      for (int u = 0; u < FF_U; ++u)
#pragma unroll
        for (int pix = 0; pix < FF_PIX; ++pix) {
          const f16x2 hv = {(f16)(acc1[u][pix][0] * gelu_erf(acc1[u][pix][1])), (f16)(acc1[u][pix][2] * gelu_erf(acc1[u][pix][3]))};
          *reinterpret_cast<f16x2*>(hb + (16 * pix + l15) * FF_CH + (((FF_U * wave + u) ^ l15) * 8) + 2 * g) = hv;
        }
      FF_T(t_glu);
      __syncthreads();          // the chunk is complete (and chunk ch - 1's buffer free again: every wave is past its FF2)
      FF_T(t_bar);

      // ---- FF2: out^T[n][row] += W2 tile x h^T, the chunk as B operand; W2 fragments two k steps at a time
#pragma unroll
      for (int half = 0; half < KS2 / 2; ++half) {
        f16x8 w2n[2][DT];
        if (half + 1 < KS2 / 2) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < DT; ++t) w2n[ks][t] = ldw2(2 * (half + 1) + ks, t);
        }
#pragma unroll
        for (int pix = 0; pix < FF_PIX; ++pix) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const f16x8 bf = *reinterpret_cast<const f16x8*>(hb + (16 * pix + l15) * FF_CH + (((4 * (2 * half + ks) + g) ^ l15) * 8));
#pragma unroll
            for (int t = 0; t < DT; ++t) acc2[pix][t] = mfma16x16x32(w2f[ks][t], bf, acc2[pix][t]);
          }
        }
        if (half + 1 < KS2 / 2) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < DT; ++t) w2f[ks][t] = w2n[ks][t];
        }
      }
      FF_T(t_ff2);
    }

    // ---- epilogue: + b2 + x, stored 16 bytes per lane (the lane pairing of motion_attn.hip)
    const float* b2 = reinterpret_cast<const float*>(p.b2) + wave * DN;
    const f16* xr0 = X + (int64_t)tile * (FF_PIX * 16) * p.ldx + wave * DN;
    f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + (int64_t)tile * (FF_PIX * 16) * p.ldo + wave * DN;
    f32x4 bias2[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
      bias2[t] = (16 * t + 4 * g < DN) ? *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    // the residual rows in the stores' own 16-byte lane pairing (even g: tile-0 channels 4 g .. + 7, odd g: tile-1 channels
    // 4 (g - 1) .. + 7, g = 0 also tile 2), all requested before the first use; the same v_permlane16_swap that assembles the
    // stores takes them apart again (it is its own inverse)
    const int c01 = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;
    u32x4 r01[FF_PIX], r2[FF_PIX];
#pragma unroll
    for (int pix = 0; pix < FF_PIX; ++pix) {
      const f16* xrow = xr0 + (int64_t)(16 * pix + l15) * p.ldx;
      r01[pix] = *reinterpret_cast<const u32x4*>(xrow + c01);
      r2[pix] = (g == 0) ? *reinterpret_cast<const u32x4*>(xrow + 32) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int pix = 0; pix < FF_PIX; ++pix) {
      u32x2 xa = {r01[pix][0], r01[pix][1]}, xb = {r01[pix][2], r01[pix][3]}, xc = {r2[pix][0], r2[pix][1]}, xd = {r2[pix][2], r2[pix][3]};
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                   "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                   : "+v"(xa[0]), "+v"(xb[0]), "+v"(xa[1]), "+v"(xb[1]), "+v"(xc[0]), "+v"(xd[0]), "+v"(xc[1]), "+v"(xd[1]));
      // now xa = this lane's tile-0 channels 4 g .. + 3, xb = its tile-1 channels, xc = its tile-2 channels (g = 0 kept its own
      // half, g = 1 received the other half of g = 0's load)
      const f16x4 xres[DT] = {__builtin_bit_cast(f16x4, xa), __builtin_bit_cast(f16x4, xb), __builtin_bit_cast(f16x4, xc)};
      u32x2 oh[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f32x4 o = acc2[pix][t] + bias2[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += (float)xres[t][r];
        oh[t] = __builtin_bit_cast(u32x2, to_half(o));
      }
      u32x2 a = oh[0], b = oh[1], c2 = oh[2], d2 = oh[2];
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                   "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                   : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(c2[0]), "+v"(d2[0]), "+v"(c2[1]), "+v"(d2[1]));
      const u32x4 v01 = {a[0], a[1], b[0], b[1]};
      const u32x4 v2 = {c2[0], c2[1], d2[0], d2[1]};
      f16* orow = O + (int64_t)(16 * pix + l15) * p.ldo;
      *reinterpret_cast<u32x4*>(orow + c01) = v01;
      if (g == 0) *reinterpret_cast<u32x4*>(orow + 32) = v2;
    }
    __syncthreads();            // the panel and the chunk buffers are free for the next tile
    FF_T(t_epi);
  }
#ifdef I2V_FF_STAMPS
  if (stamps != nullptr && lane == 0) {
    long long* st = stamps + ((int64_t)blockIdx.x * H + wave) * 8;
    st[0] = t_ln; st[1] = t_ff1; st[2] = t_glu; st[3] = t_bar; st[4] = t_ff2; st[5] = t_epi;
  }
#endif
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int32_t i2v_ff_fused_supported(int64_t rows, int32_t channels, int32_t inner) {
  return rows > 0 && rows % (FF_PIX * 16) == 0 && rows / (FF_PIX * 16) < (1 << 24) && channels == 320 && inner == 1280;
}

extern "C" int i2v_ff_fused_f16(const i2v_ff_fused_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_ff_fused_f16: null params");
  const i2v_ff_fused_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.w1 && p.b1 && p.w2 && p.b2 && p.out, "i2v_ff_fused_f16: null pointer");
  I2V_CHECK_ARG(i2v_ff_fused_supported(p.rows, p.channels, p.inner),
                "i2v_ff_fused_f16: rows %lld channels %d inner %d is not a fused shape (i2v_ff_fused_supported)", (long long)p.rows,
                p.channels, p.inner);
  I2V_CHECK_ARG(p.ldx >= p.channels && p.ldx % 8 == 0 && p.ldo >= p.channels && p.ldo % 8 == 0, "i2v_ff_fused_f16: row strides");
  I2V_CHECK_ARG(al16(p.x) && al16(p.gamma) && al16(p.beta) && al16(p.w1) && al16(p.b1) && al16(p.w2) && al16(p.b2) && al16(p.out),
                "i2v_ff_fused_f16: pointers must be 16-byte aligned");
  constexpr int C = 320, INNER = 1280;
  const size_t lds = ((size_t)FF_PIX * 16 * C + 2 * (size_t)FF_PIX * 16 * FF_CH) * sizeof(f16);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_fused_kernel<C, INNER, 8>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!attr_ok) I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_ff_fused_f16: %zu bytes of LDS refused", lds);
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  const int ntiles = (int)(p.rows / (FF_PIX * 16));
  const int per = (ntiles + cus - 1) / cus;
  const int grid = (ntiles + per - 1) / per;
  long long* stamps = nullptr;
#ifdef I2V_FF_STAMPS
  stamps = getenv("I2V_FF_STAMP_PTR") ? reinterpret_cast<long long*>(strtoull(getenv("I2V_FF_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
  hipLaunchKernelGGL((ff_fused_kernel<C, INNER, 8>), dim3((unsigned)grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), p, ntiles,
                     stamps);
  return i2v_check_launch("i2v_ff_fused_f16");
}
