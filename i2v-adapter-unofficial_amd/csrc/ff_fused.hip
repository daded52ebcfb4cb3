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
// Epilogue: + b2 + x, 16-byte stores assembled by v_permlane16_swap.
//
// (r5) What the round-4 form left on the table (profiles/r4_traffic.json: 410 MB below L2 per launch against 168 algorithmic; of a
// workgroup's 643k cycles 58k were the LayerNorm phase and 85k the epilogue, both behind an exposed HBM round trip):
//   * all 160 KB of LDS are used: the chunk buffers (64 KB) + 16 KB are a SECOND 80 KB region.  The next tile's raw rows are
//     requested into it (LDS-DMA) as soon as the last chunk's FF2 has released the buffers and arrive under the epilogue; the
//     tile starts by normalising them into the panel -- no wait on HBM at the top of a tile.
//   * the residual rows are not re-read in 80-byte head slices by every wave: when the last chunk's FF1 is done the panel is
//     free, the tile's raw rows come back into it by DMA as WHOLE rows (16-byte chunks at the panel's own swizzled places) under
//     the last FF2, and the epilogue takes the residual from LDS in the accumulator layout (no lane exchange to take it apart).
//   * PROJ: the Linear that follows the block -- the spatial transformer's proj_out (+ its residual, i2v:298-314) or the motion
//     module's (+ residual, rows stored back in (batch, frame, pixel) order) -- runs in the same launch: the block's fp16 output
//     replaces the residual in the panel in place, and a fourth pass (wave w: 40 output channels, 240 MFMAs) projects it.  The
//     un-fused GEMM (131072 x 320 x 320 + residual, 61 - 67 us, HBM-bound) read and wrote 250 MB for 27 GFLOP.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void ff_dma16(__amdgpu_buffer_rsrc_t rsrc, f16* lds_wave_base, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// a copy of a lane-dependent value that the compiler cannot see through: address arithmetic derived from it is redone where it
// is used instead of being hoisted out of the tile loop as a loop invariant.  (r5: hoisted, the ten DMA offsets and the sixteen
// store addresses of a tile were SPILLED, and every reload was followed by `s_waitcnt vmcnt(0)` -- scratch loads count on vmcnt --
// so each DMA instruction and each store waited for the one before it to complete: ten HBM round trips in series per tile.)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr int FF_PIX = 8;              // 16-row tiles per workgroup tile (128 rows)
constexpr int FF_CH = 128;             // inner channels per chunk: 8 waves x 2 tiles x 8 (value, gate) pairs
constexpr int FF_U = 2;                // W1 tiles per wave and chunk: a panel fragment read feeds FF_U MFMAs (with one, FF1 ran at
                                       // 28 % of the MFMA rate: 357k of a workgroup's 704k cycles)
constexpr int FF_PD = 3;               // W1 fragments in flight (K steps ahead)
constexpr int FF_AD = 4;               // panel fragments in flight

// HILO (PROJ only; i2v_ff_fused_params.res2_lo / out_lo): the module's residual stream as an fp16 pair -- the low halves of the outer
// residual rows are added with the high ones and the low half of the result is stored beside it (see i2v_gemm_params.residual_lo).
template <int C, int INNER, int H, bool PROJ, bool HILO = false>
__global__ __launch_bounds__(64 * H) void ff_fused_kernel(const i2v_ff_fused_params p, const int ntiles, long long* __restrict__ stamps) {
  constexpr int KS = C / 32, NJ = C / 64, NCH = INNER / FF_CH, DN = C / H, DT = (DN + 15) / 16;
  static_assert(C % 64 == 0 && H == 8 && INNER % FF_CH == 0 && DT == 3 && DN == 40, "SD-1.5 64^2 level: C = 320, inner = 1280");
  static_assert(PROJ || !HILO, "the precise stream enters and leaves through the tail");
  extern __shared__ __attribute__((aligned(16))) f16 lds[];
  f16* panel = lds;                                   // [128][C], 16-byte chunk index ^= row & 7
  f16* hbuf = lds + FF_PIX * 16 * C;                  // 2 x [128][128], 16-byte chunk index ^= row & 15
  f16* raw = lds + FF_PIX * 16 * C;                   // ... and, between two tiles' chunk loops, the next tile's raw rows [128][C]
  static_assert(2 * FF_PIX * 16 * FF_CH <= FF_PIX * 16 * C, "the raw rows of a tile cover the chunk buffers (160 KB of LDS in all)");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15, sub = lane & 7, sw = l15 & 7;
  // (chunk 4 s + g of a K step swizzled by the row: ((4 s + g) ^ sw) = 8 (s >> 1) + ((4 (s & 1) + g) ^ sw) -- TWO lane-dependent offsets
  //  and an immediate instead of one hoisted (and spilled) register per K step)
  const int swz[2] = {(g ^ sw) * 8, ((4 + g) ^ sw) * 8};
  const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x);

  // ---- rows of a tile by LDS-DMA into the wave's own 10 KB of the panel, normalised in place (as motion_attn.hip)
  auto fetch_rows = [&](const int tile) {
    const f16* base = X + ((int64_t)tile * (FF_PIX * 16) + 16 * wave) * p.ldx;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)((15 * p.ldx + C) * 2), 0x00020000);
    const int ln = opaque(lane);
    const unsigned voff = (unsigned)(((ln >> 3) * p.ldx + (ln & 7) * 8) * 2);        // the instruction's part is a scalar offset
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        ff_dma16(rs, raw + 16 * wave * C + (half * NJ + j) * 512, voff, (unsigned)((8 * half * p.ldx + 8 * j * 8) * 2));
  };
  // the wave's 16 rows again, this time to their SWIZZLED places in the panel (row r of the panel, 16-byte chunk c at position
  // c ^ (r & 7)): the destination of an LDS-DMA is lane-linear (64 consecutive 16-byte slots per instruction), so the swizzle is
  // applied to the source address
  auto fetch_residual = [&](const int tile) {
    const f16* base = X + ((int64_t)tile * (FF_PIX * 16) + 16 * wave) * p.ldx;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)((15 * p.ldx + C) * 2), 0x00020000);
    constexpr int CPR = C / 8;                         // 16-byte chunks per row
    static_assert((16 * CPR) % 64 == 0 && CPR % 8 == 0, "whole DMA instructions per wave, swizzle inside groups of 8 chunks");
    const int ln = opaque(lane);
#pragma unroll
    for (int i = 0; i < 16 * CPR / 64; ++i) {
      const int q = i * 64 + ln, r = q / CPR, cp = q - r * CPR;
      ff_dma16(rs, panel + 16 * wave * C + i * 512, (unsigned)((r * p.ldx + (cp ^ (r & 7)) * 8) * 2));
    }
  };
  auto normalise_rows = [&]() {
    f16x8 xv[2][NJ];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j) xv[half][j] = *reinterpret_cast<const f16x8*>(raw + 16 * wave * C + ((half * NJ + j) * 64 + lane) * 8);
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
  long long t_ln = 0, t_ff1 = 0, t_glu = 0, t_bar = 0, t_ff2 = 0, t_epi = 0, t_w = 0, t_p = 0, t0;
#define FF_T(acc) { const long long t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define FF_T(acc)
#endif
  // W3 (PROJ): [wave][K step][tile 3][64][8], the wave's 40 output rows padded to 48 (the layout of i2v_cross_attn_fused's w_q)
  const auto rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PROJ ? p.w3 : p.w2), 0, H * KS * DT * 1024, 0x00020000);
  // output row of tile row r (PROJ with perm_frames > 0: rows arrive in (batch, pixel, frame) order and leave -- like the residual
  // res2 is read -- in (batch, frame, pixel) order, the I2V_STORE_ROWPERM of i2v_gemm_f16)
  // (32-bit throughout, one division per tile: as a 64-bit division per row the index arithmetic was 1900 instructions of a tile)
  const int perm_shift = (PROJ && p.perm_frames > 0) ? __builtin_ctz((unsigned)p.perm_frames) : 0;
  auto out_rows = [&](const int tile, const int l15, int (&orow)[FF_PIX]) {
    const unsigned r0 = (unsigned)tile * (FF_PIX * 16);
    if (!PROJ || p.perm_frames <= 0) {
#pragma unroll
      for (int pix = 0; pix < FF_PIX; ++pix) orow[pix] = (int)(r0 + 16 * pix + l15);
      return;
    }
    const unsigned hw = (unsigned)p.perm_hw, pp0 = r0 >> perm_shift;
    const unsigned b0 = __builtin_amdgcn_readfirstlane(pp0 / hw), px0 = pp0 - b0 * hw;      // (tile-uniform)
#pragma unroll
    for (int pix = 0; pix < FF_PIX; ++pix) {
      const unsigned r = r0 + 16 * pix + l15, f = r & (unsigned)(p.perm_frames - 1);
      unsigned px = px0 + ((r >> perm_shift) - pp0), bb = b0;
      while (px >= hw) {          // a tile holds at most 16 pixels: it seldom crosses into the next clip
        px -= hw;
        ++bb;
      }
      orow[pix] = (int)((bb * (unsigned)p.perm_frames + f) * hw + px);
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;        // (workgroup-uniform)
  fetch_rows(tile);
  for (; tile < ntiles; tile += gridDim.x) {
#ifdef I2V_FF_STAMPS
    t0 = __builtin_amdgcn_s_memtime();
#endif
    // the tile's raw rows were requested under the previous tile's epilogue (the first ones just above)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    normalise_rows();
    lds_barrier();
    FF_T(t_ln);

    // (the accumulators start from the bias: loaded in the epilogue -- behind the DMA of the next tile's rows, loads return in
    //  order -- these twelve L2-warm values waited for an HBM round trip, and the whole epilogue with them)
    f32x4 acc2[FF_PIX][DT];
    {
      const float* b2 = reinterpret_cast<const float*>(p.b2) + wave * DN;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const f32x4 bias2 = (16 * t + 4 * g < DN) ? *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pix = 0; pix < FF_PIX; ++pix) acc2[pix][t] = bias2;
      }
    }

    // (r5, measured and dropped: the two waves of a SIMD -- w and w + 4 -- taking the pieces between two barriers in different
    // orders, waves 0-3 FF2(ch - 1), FF1(ch), GEGLU(ch) and waves 4-7 FF1(ch), GEGLU(ch), FF2(ch - 1), so that one's GEGLU would run
    // beside the other's MFMAs: 369 -> 400 us.  The group that reaches the barrier first waits for the other in every chunk.)
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
        return *reinterpret_cast<const f16x8*>(alane + 16 * (i % FF_PIX) * C + 64 * ((i / FF_PIX) >> 1) + swz[(i / FF_PIX) & 1]);
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
      lds_barrier();          // the chunk is complete (and chunk ch - 1's buffer free again: every wave is past its FF2)
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
        // last chunk: every wave is past its FF1 (the barrier above), nobody reads the panel any more -- the tile's raw rows
        // come back into it for the epilogue.  Requested BEHIND this chunk's last weight fragments: loads return in order, and
        // FF2 must not wait for an HBM round trip.
        if (half == 0 && ch == NCH - 1) {
          fetch_residual(tile);
          __builtin_amdgcn_sched_barrier(0);
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

    // every wave's residual rows have landed, and every wave has left the chunk buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    FF_T(t_w);
    const int next = tile + (int)gridDim.x;
    // the next tile's raw rows: into the chunk buffers' region, under this epilogue.  (PROJ: requested further down, behind the
    // third projection's last weight fragments -- loads return in order, and in front of them this HBM round trip stalled the
    // projection at its first fragment wait: 30k cycles per tile for 8k of MFMA.)
    if (!PROJ && next < ntiles) fetch_rows(next);
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: + b2 + x (from the panel, in the accumulator layout).  This lane's places in the panel (row 16 pix + l15,
    // channels 40 wave + 16 t + 4 g .. + 3): the 8-byte half (g & 1) of chunk 5 wave + 2 t + (g >> 1), swizzled by the row -- from an
    // opaque copy of the lane index (see `opaque`)
    static_assert(DN == 40, "5 chunks of 8 channels per wave");
    const int eln = opaque(lane), eg = eln >> 4, el15 = eln & 15;
    int res_off[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) res_off[t] = el15 * C + (((5 * wave + 2 * t + (eg >> 1)) ^ (el15 & 7)) * 8) + 4 * (eg & 1);
    const int c01 = (eg & 1) ? 16 + 4 * (eg - 1) : 4 * eg;
    // 16-byte stores: even g ends with tile-0 channels 4 g .. + 7, odd g with tile-1 channels 4 (g - 1) .. + 7, g = 0 also tile 2
    auto store_row = [&](f16* orow, const u32x2 (&oh)[DT]) {
      u32x2 a = oh[0], b = oh[1], c2 = oh[2], d2 = oh[2];
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                   "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                   : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(c2[0]), "+v"(d2[0]), "+v"(c2[1]), "+v"(d2[1]));
      const u32x4 v01 = {a[0], a[1], b[0], b[1]};
      const u32x4 v2 = {c2[0], c2[1], d2[0], d2[1]};
      *reinterpret_cast<u32x4*>(orow + c01) = v01;
      if (eg == 0) *reinterpret_cast<u32x4*>(orow + 32) = v2;
    };
    if constexpr (!PROJ) {
      f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + (int64_t)tile * (FF_PIX * 16) * p.ldo + wave * DN;
#pragma unroll
      for (int pix = 0; pix < FF_PIX; ++pix) {
        u32x2 oh[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const f16x4 xres = (16 * t + 4 * eg < DN) ? *reinterpret_cast<const f16x4*>(panel + 16 * pix * C + res_off[t]) : f16x4{0, 0, 0, 0};
          f32x4 o = acc2[pix][t];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] += (float)xres[r];
          oh[t] = __builtin_bit_cast(u32x2, to_half(o));
        }
        store_row(O + (int64_t)(16 * pix + el15) * p.ldo, oh);
      }
    } else {
      // the outer residual rows (in OUTPUT order) in the stores' own 16-byte lane pairing, requested inside the projection (below)
      const f16* __restrict__ R2 = reinterpret_cast<const f16*>(p.res2) + wave * DN;
      f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + wave * DN;
      int orow[FF_PIX];                          // (rows < 2^31: ntiles < 2^24)
      constexpr int HP = FF_PIX / 2;
      u32x4 r01[2][HP], r2[2][HP];
      auto load_res2 = [&](const int h) {
#pragma unroll
        for (int q = 0; q < HP; ++q) {
          const f16* rrow = R2 + (int64_t)orow[h * HP + q] * p.ld_res2;
          r01[h][q] = *reinterpret_cast<const u32x4*>(rrow + c01);
          r2[h][q] = (eg == 0) ? *reinterpret_cast<const u32x4*>(rrow + 32) : u32x4{0u, 0u, 0u, 0u};
        }
      };
      // HILO: the low halves of the same rows, four rows at a time after the projection (all eight held beside the accumulators and
      // the high halves spilled 60 registers): the second four are requested when the first four rows have been stored -- one
      // exposed round trip per tile
      const f16* __restrict__ R2L = reinterpret_cast<const f16*>(HILO ? p.res2_lo : p.res2) + wave * DN;
      f16* __restrict__ OL = reinterpret_cast<f16*>(HILO ? p.out_lo : p.out) + wave * DN;
      u32x4 l01[HP], l2[HP];
      auto load_res2_lo = [&](const int h) {
        if constexpr (HILO) {
#pragma unroll
          for (int q = 0; q < HP; ++q) {
            const f16* rrow = R2L + (int64_t)orow[h * HP + q] * p.ld_res2;
            l01[q] = *reinterpret_cast<const u32x4*>(rrow + c01);
            l2[q] = (eg == 0) ? *reinterpret_cast<const u32x4*>(rrow + 32) : u32x4{0u, 0u, 0u, 0u};
          }
        }
      };
      out_rows(tile, el15, orow);
      // the block's output y = x + FF(LayerNorm(x)), rounded to fp16 as the un-fused kernel stores it, replaces x in the panel
      // in place (each wave touches its own 40 channels only)
#pragma unroll
      for (int pix = 0; pix < FF_PIX; ++pix)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          if (16 * t + 4 * eg < DN) {
            f16x4* slot = reinterpret_cast<f16x4*>(panel + 16 * pix * C + res_off[t]);
            const f16x4 xres = *slot;
            f32x4 o = acc2[pix][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += (float)xres[r];
            *slot = to_half(o);
          }
        }
      lds_barrier();
      // ---- out^T[n][row] = W3 tile x y^T: the third projection (240 MFMAs per wave)
      f32x4 acc3[FF_PIX][DT];
      {
        const float* b3 = reinterpret_cast<const float*>(p.b3) + wave * DN;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const f32x4 bias3 = (16 * t + 4 * eg < DN) ? *reinterpret_cast<const f32x4*>(b3 + 16 * t + 4 * eg) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int pix = 0; pix < FF_PIX; ++pix) acc3[pix][t] = bias3;
        }
      }
      {
        const int w3o = wave * (KS * DT * 1024);
        auto ldw3 = [&](const int s, const int t) {
          return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, w_lane, w3o + (s * DT + t) * 1024, 0));
        };
        f16x8 wf[FF_PD][DT];
#pragma unroll
        for (int s = 0; s < FF_PD - 1; ++s)
#pragma unroll
          for (int t = 0; t < DT; ++t) wf[s][t] = ldw3(s, t);
        constexpr int NI = KS * FF_PIX;
        auto lda = [&](const int i) {
          return *reinterpret_cast<const f16x8*>(alane + 16 * (i % FF_PIX) * C + 64 * ((i / FF_PIX) >> 1) + swz[(i / FF_PIX) & 1]);
        };
        f16x8 af[FF_AD + 1];
#pragma unroll
        for (int i = 0; i < FF_AD; ++i) af[i] = lda(i);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int s = i / FF_PIX, pix = i % FF_PIX;
          if (pix == 0 && s + FF_PD - 1 < KS) {
#pragma unroll
            for (int t = 0; t < DT; ++t) wf[(s + FF_PD - 1) % FF_PD][t] = ldw3(s + FF_PD - 1, t);
          }
          if (pix == 0 && s + FF_PD - 1 == KS - 1) {      // the last weight fragments are on their way: now the HBM requests
            if (next < ntiles) fetch_rows(next);
            load_res2(0);
          }
          if (i + FF_AD < NI) af[(i + FF_AD) % (FF_AD + 1)] = lda(i + FF_AD);
#pragma unroll
          for (int t = 0; t < DT; ++t) acc3[pix][t] = mfma16x16x32(wf[s % FF_PD][t], af[i % (FF_AD + 1)], acc3[pix][t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FF_T(t_p);
      load_res2(1);       // (all 16 loads held across the projection's tail spilled 100 registers)
      load_res2_lo(0);
#pragma unroll
      for (int pix = 0; pix < FF_PIX; ++pix) {
        if (pix == HP) {
          __builtin_amdgcn_sched_barrier(0);
          load_res2_lo(1);
        }
        // the same v_permlane16_swap that assembles the stores takes the residual's 16-byte pieces apart (it is its own inverse)
        const u32x4 q01 = r01[pix / HP][pix % HP], q2 = r2[pix / HP][pix % HP];
        u32x2 xa = {q01[0], q01[1]}, xb = {q01[2], q01[3]}, xc = {q2[0], q2[1]}, xd = {q2[2], q2[3]};
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                     "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                     : "+v"(xa[0]), "+v"(xb[0]), "+v"(xa[1]), "+v"(xb[1]), "+v"(xc[0]), "+v"(xd[0]), "+v"(xc[1]), "+v"(xd[1]));
        const f16x4 xres[DT] = {__builtin_bit_cast(f16x4, xa), __builtin_bit_cast(f16x4, xb), __builtin_bit_cast(f16x4, xc)};
        f16x4 xlo[DT] = {};
        if constexpr (HILO) {
          const u32x4 p01 = l01[pix % HP], p2 = l2[pix % HP];
          u32x2 la = {p01[0], p01[1]}, lb = {p01[2], p01[3]}, lc = {p2[0], p2[1]}, ld = {p2[2], p2[3]};
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                       "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                       : "+v"(la[0]), "+v"(lb[0]), "+v"(la[1]), "+v"(lb[1]), "+v"(lc[0]), "+v"(ld[0]), "+v"(lc[1]), "+v"(ld[1]));
          xlo[0] = __builtin_bit_cast(f16x4, la);
          xlo[1] = __builtin_bit_cast(f16x4, lb);
          xlo[2] = __builtin_bit_cast(f16x4, lc);
        }
        u32x2 oh[DT], ol[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          f32x4 o = acc3[pix][t];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] += (float)xres[t][r];
          if constexpr (HILO) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += (float)xlo[t][r];
          }
          const f16x4 o16 = to_half(o);
          oh[t] = __builtin_bit_cast(u32x2, o16);
          if constexpr (HILO) {
            f32x4 d;
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = o[r] - (float)o16[r];
            ol[t] = __builtin_bit_cast(u32x2, to_half(d));
          }
        }
        store_row(O + (int64_t)orow[pix] * p.ldo, oh);
        if constexpr (HILO) store_row(OL + (int64_t)orow[pix] * p.ldo, ol);
      }
    }
    lds_barrier();            // the panel is free for the next tile's normalised rows
    FF_T(t_epi);
  }
#ifdef I2V_FF_STAMPS
  if (stamps != nullptr && lane == 0) {
    long long* st = stamps + ((int64_t)blockIdx.x * H + wave) * 8;
    st[0] = t_ln; st[1] = t_ff1; st[2] = t_glu; st[3] = t_bar; st[4] = t_ff2; st[5] = t_epi; st[6] = t_w; st[7] = t_p;
  }
#endif
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

constexpr size_t FF_LDS = 2 * (size_t)FF_PIX * 16 * 320 * sizeof(f16);      // panel + (chunk buffers | raw rows): 160 KB

template <bool PROJ, bool HILO = false>
int ff_cus() { return i2v_big_lds_kernel_cus(reinterpret_cast<const void*>(ff_fused_kernel<320, 1280, 8, PROJ, HILO>), FF_LDS); }

}  // namespace

extern "C" int32_t i2v_ff_fused_supported(int64_t rows, int32_t channels, int32_t inner) {
  return rows > 0 && rows % (FF_PIX * 16) == 0 && rows / (FF_PIX * 16) < (1 << 24) && channels == 320 && inner == 1280 &&
         ff_cus<false>() > 0;
}

extern "C" int32_t i2v_ff_fused_tail_supported(int64_t rows, int32_t channels, int32_t inner, int32_t perm_frames, int32_t perm_hw) {
  if (!(rows > 0 && rows % (FF_PIX * 16) == 0 && rows / (FF_PIX * 16) < (1 << 24) && channels == 320 && inner == 1280)) return 0;
  if (perm_frames != 0 || perm_hw != 0) {
    // (batch, pixel, frame) rows: whole clips, a power-of-two frame count
    if (perm_frames <= 0 || (perm_frames & (perm_frames - 1)) || perm_hw <= 0 || rows % ((int64_t)perm_frames * perm_hw) != 0) return 0;
  }
  return ff_cus<true>() > 0;
}

extern "C" int i2v_ff_fused_f16(const i2v_ff_fused_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_ff_fused_f16: null params");
  const i2v_ff_fused_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.w1 && p.b1 && p.w2 && p.b2 && p.out, "i2v_ff_fused_f16: null pointer");
  const bool tail = p.w3 != nullptr;
  I2V_CHECK_ARG(p.rows > 0 && p.rows % (FF_PIX * 16) == 0 && p.rows / (FF_PIX * 16) < (1 << 24) && p.channels == 320 && p.inner == 1280,
                "i2v_ff_fused_f16: rows %lld channels %d inner %d is not a fused shape (i2v_ff_fused_supported)", (long long)p.rows,
                p.channels, p.inner);
  I2V_CHECK_ARG(p.ldx >= p.channels && p.ldx % 8 == 0 && p.ldo >= p.channels && p.ldo % 8 == 0, "i2v_ff_fused_f16: row strides");
  I2V_CHECK_ARG(al16(p.x) && al16(p.gamma) && al16(p.beta) && al16(p.w1) && al16(p.b1) && al16(p.w2) && al16(p.b2) && al16(p.out),
                "i2v_ff_fused_f16: pointers must be 16-byte aligned");
  // one workgroup reads a tile's rows through a 32-bit buffer descriptor
  I2V_CHECK_ARG(p.ldx < (1 << 24), "i2v_ff_fused_f16: ldx");
  if (tail) {
    I2V_CHECK_ARG(p.b3 && p.res2, "i2v_ff_fused_f16: the tail needs w3, b3 and res2");
    I2V_CHECK_ARG(al16(p.w3) && al16(p.b3) && al16(p.res2) && p.ld_res2 >= p.channels && p.ld_res2 % 8 == 0,
                  "i2v_ff_fused_f16: tail operands must be 16-byte aligned, ld_res2 a multiple of 8");
    I2V_CHECK_ARG(i2v_ff_fused_tail_supported(p.rows, p.channels, p.inner, p.perm_frames, p.perm_hw) ||
                      ff_cus<true>() == 0,
                  "i2v_ff_fused_f16: perm_frames %d / perm_hw %d do not describe rows %lld (i2v_ff_fused_tail_supported)", p.perm_frames,
                  p.perm_hw, (long long)p.rows);
    I2V_CHECK_ARG(!(p.perm_frames > 0 && p.out == p.x), "i2v_ff_fused_f16: out must not alias x when the tail permutes the rows");
    I2V_CHECK_ARG((p.res2_lo == nullptr) == (p.out_lo == nullptr) && al16(p.res2_lo) && al16(p.out_lo) && (p.out_lo == nullptr || p.out_lo != p.out),
                  "i2v_ff_fused_f16: res2_lo and out_lo come together, 16-byte aligned");
  } else {
    I2V_CHECK_ARG(p.b3 == nullptr && p.res2 == nullptr && p.perm_frames == 0 && p.perm_hw == 0 && p.res2_lo == nullptr && p.out_lo == nullptr,
                  "i2v_ff_fused_f16: tail fields set without w3");
  }
  const bool hilo = tail && p.out_lo != nullptr;
  const int cus = hilo ? ff_cus<true, true>() : tail ? ff_cus<true>() : ff_cus<false>();
  if (cus <= 0) I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_ff_fused_f16: %zu bytes of LDS refused by this device", FF_LDS);
  const int ntiles = (int)(p.rows / (FF_PIX * 16));
  const int grid = i2v_persistent_grid(ntiles, cus);
  long long* stamps = nullptr;
#ifdef I2V_FF_STAMPS
  stamps = getenv("I2V_FF_STAMP_PTR") ? reinterpret_cast<long long*>(strtoull(getenv("I2V_FF_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hilo)
    hipLaunchKernelGGL((ff_fused_kernel<320, 1280, 8, true, true>), dim3((unsigned)grid), dim3(512), FF_LDS, s, p, ntiles, stamps);
  else if (tail)
    hipLaunchKernelGGL((ff_fused_kernel<320, 1280, 8, true>), dim3((unsigned)grid), dim3(512), FF_LDS, s, p, ntiles, stamps);
  else
    hipLaunchKernelGGL((ff_fused_kernel<320, 1280, 8, false>), dim3((unsigned)grid), dim3(512), FF_LDS, s, p, ntiles, stamps);
  return i2v_check_launch("i2v_ff_fused_f16");
}
