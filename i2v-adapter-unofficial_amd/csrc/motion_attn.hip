// Fused attention sub-block of the motion module (TransformerTemporalModel -> BasicTransformerBlock with
// double self-attention over the FRAMES of one pixel, unet:232-244 / 413-425 / 607-619):
//     n = LayerNorm(t) + pe[frame];   q, k, v = n Wq^T, n Wk^T, n Wv^T;   o = softmax(q k^T / sqrt(d)) v      per pixel and head
// in ONE launch: t is read once, o is written once.  Un-fused this was four launches at the 64^2 level (q|k GEMM 105 us, V^T
// GEMM 59 us, temporal attention 74 us: q, k and V^T -- 250 MB -- written and read back) for 80 GFLOP.
//
// Rows are in (batch, pixel, frame) order, so with 16 frames a 16-row MFMA tile IS one pixel's sequence.  A workgroup of 8
// waves owns 128 rows (8 pixels) at a time: their LayerNorm-ed rows sit in LDS once (80 KB, 16-byte chunks XOR-swizzled by the
// row so the 16 rows of a fragment read land on distinct banks), wave w owns head w.  No intermediate leaves the registers:
//   * q and k are projected TRANSPOSED, D[channel][frame] = W_tile (A operand, straight from global: each wave streams only its
//     own head's rows of W) x n^T (B operand from LDS).  In the accumulator layout lane l then holds channels 4 (l >> 4) + r of
//     frame l & 15 -- which is exactly the A / B operand layout of v_mfma_f32_16x16x16_f16 with the CHANNEL as contraction
//     index: S^T[key][query] = sum_c K[key][c] Q[query][c] takes the converted accumulators as they are.
//   * softmax over the keys = over the 4 accumulator rows of a lane and the 4 lane groups (two lane-xor steps); the fp16
//     P^T[key][query] is again in B-operand layout, now with the KEY as contraction index.
//   * v is projected NON-transposed, D[key frame][channel]: its accumulators are the A operand V^T[channel][key] of
//     O^T[channel][query] = sum_key V^T[channel][key] P^T[key][query].
// head_dim 40 is padded to 48 per q / k / v by zero rows in the packed weights (i2v_motion_attn_pack_rows), which are stored in
// FRAGMENT order -- [head][q|k|v][K step][16-row tile][lane][8] -- so that a wave's load instruction reads 1 KB of whole lines
// (row-major, an instruction touched 16 rows x 64 B: with those loads a workgroup took 67k cycles per tile, without any 50k).
//
// What the in-kernel stamps (tools/motion_attn_probe.py, -DI2V_MA_STAMPS) showed and the structure answers:
//   * the two waves of a SIMD (w and w + 4) do not share the matrix pipe evenly -- the older one runs its three passes in 25k
//     cycles, the younger needs 38k; alternating s_setprio per K step made both slower (121 -> 145 us), raising the younger wave
//     through the q pass evened them out (45k / 48k) at the same total: the SIMD's issue port is the bound -- so no wave waits
//     for another inside a tile: the workgroup is persistent over its tiles with TWO panels, a wave that has finished tile i
//     normalises its 16 rows of tile i + 1 into the other panel straight away, and there is one barrier per tile;
//   * the LayerNorm phase was a quarter of the tile (12k of 50k cycles: an HBM round trip, then 11 VALU operations per element):
//     the next tile's rows are fetched into registers before the v pass and arrive under it, gamma and (beta + pe[frame]) come
//     as fp32 tables (no conversions, one add less): 6.5 operations per element.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16x16x16(f16x4 a, f16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane, global -> LDS at (wave-uniform LDS base) + 16 * lane (a plain function
// on purpose, as in gemm_big.hip: called directly from the kernel template, the builtin makes hipcc drop the launch stub)
__device__ __forceinline__ void ma_dma16(__amdgpu_buffer_rsrc_t rsrc, f16* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
// a copy of a lane-dependent value that the compiler cannot see through: address arithmetic derived from it is redone where it is
// used instead of being hoisted out of the tile loop.  (r5: hoisted, the ten DMA offsets of `fetch_rows` and the output rows' store
// addresses were SPILLED, and each reload was followed by `s_waitcnt vmcnt(0)` -- scratch loads count on vmcnt -- so every DMA
// instruction waited for the one before it to land and every store for the weight fragments in flight.)
__device__ __forceinline__ int ma_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr int MA_PIX = 8;              // pixels per tile (x 16 frames = 128 rows)
constexpr int MA_F = 16;               // frames per pixel: one MFMA tile
constexpr int MA_PD = 3;               // weight fragments in flight (K steps ahead)
constexpr int MA_AD = 4;               // panel fragments in flight (reads ahead of the MFMAs that take them)

// what both entry points hand the kernel (the fields of i2v_motion_attn_params / i2v_cross_attn_fused_params)
struct ma_args {
  const void* x; int64_t ldx;
  const void* gamma; const void* shift; int64_t ld_shift;
  const void* w;
  void* out; int64_t ldo;
  float eps;
  // CROSS only: the context's projected keys and values as MFMA fragments (i2v_cross_attn_fused_params.ctx_frag)
  const void* ctx_frag;
  int32_t lt; int32_t tiles_per_ctx;
  // CROSS, optional: the IP-Adapter's image tokens of the same rows' context (a second softmax, added with its own weight)
  const void* ip_frag; int32_t ip_len; float ip_scale;
  // OUTP: the out-projection that follows (to_out[0]: weight fragments in the order of the q part, fp32 bias); out = x + o Wo^T + bo
  const void* w_o; const void* b_o;
};

constexpr int MA_KT = 5;               // CROSS: key tiles of 16 (<= 80 context tokens: the 77 of CLIP)

// CROSS = false: the motion module's temporal self-attention described above.
// CROSS = true: `norm2 -> attn2` of the spatial block (i2v:510-533) for a context that fits the registers -- LayerNorm, to_q, then
//   per 16 queries S^T = K_ctx Q^T (15 MFMAs of 16x16x16 against key fragments that stay in registers for the whole tile), softmax
//   over the <= 80 keys in the lane, O^T = V_ctx^T P^T (15 more): one pass over the panel instead of three, no q in memory.
// F (motion form): frames per pixel -- 16: a 16-row MFMA tile IS one pixel's sequence; 8: a tile holds two pixels and the scores
//   between them are masked; 32: a pixel is two tiles, its scores four 16 x 16 blocks (r5: the 8-frame / 256^2 and the 32-frame /
//   768^2 configurations took the un-fused chain)
// OUTP (r5): the sub-block's out-projection and residual in the same launch, out = x + o Wo^T + bo -- the heads'
//   outputs meet in the tile's panel once every wave has left it (barrier), a fourth pass projects them (wave w: output channels
//   40 w .. + 39), the residual rows are requested while the attention runs and are the accumulators' initial values.  o never
//   goes to memory (84 MB each way at the 64^2 level) and the HBM-bound 131072 x 320 x 320 + residual GEMM launch disappears.
template <int C, int D, int H, bool CROSS, int F = 16, bool OUTP = false>
__global__ __launch_bounds__(64 * H) void motion_attn_kernel(const ma_args p, const float scale_log2, const int ntiles,
                                                             long long* __restrict__ stamps) {
  constexpr int DT = (D + 15) / 16, DP = 16 * DT, KS = C / 32, NJ = C / 64, PARTS = CROSS ? 1 : 3;
  static_assert(C % 64 == 0 && H == 8, "one wave per head, 8 lanes x C / 64 chunks per row in the LayerNorm pass");
  static_assert(F == 8 || F == 16 || F == 32, "frames per pixel");
  extern __shared__ __attribute__((aligned(16))) f16 panels[];       // 2 x [128][C], chunk index ^= row & 7
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: SGPR)
  const int g = lane >> 4, l15 = lane & 15, sub = lane & 7;
  const f16* __restrict__ X = reinterpret_cast<const f16*>(p.x);
  const float* __restrict__ gamma = reinterpret_cast<const float*>(p.gamma);
  const float* __restrict__ shift = reinterpret_cast<const float*>(p.shift);
#ifdef I2V_MA_STAMPS
  long long stamp[8];
#define MA_STAMP(k) stamp[k] = __builtin_amdgcn_s_memtime()
#else
#define MA_STAMP(k)
#endif

  // ---- LayerNorm (+ positional table) of the wave's 16 rows into a panel: 8 lanes per row, two groups of 8 rows.  The raw rows
  // come by LDS-DMA into the wave's own 10 KB of the panel (16 rows x 640 B, in [group][chunk column][lane] order) and are
  // normalised in place: read whole into registers, then written to their swizzled places -- no register holds a row while the
  // DMA is in flight under the v pass of the previous tile (held in registers there, they spilled: 259 dwords, 121 -> 230 us).
  auto fetch_rows = [&](const int tile, f16* panel) {
    const f16* base = X + ((int64_t)tile * (MA_PIX * MA_F) + 16 * wave) * p.ldx;        // (wave-uniform)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)((15 * p.ldx + C) * 2), 0x00020000);
    const int ln = ma_opaque(lane);
    const unsigned voff = (unsigned)(((ln >> 3) * p.ldx + (ln & 7) * 8) * 2);      // the instruction's part is a scalar offset
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        ma_dma16(rs, panel + 16 * wave * C + (half * NJ + j) * 512, voff, (unsigned)((8 * half * p.ldx + 8 * j * 8) * 2));
  };
  auto normalise_rows = [&](f16* panel) {
    f16x8 xv[2][NJ];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < NJ; ++j) xv[half][j] = *reinterpret_cast<const f16x8*>(panel + 16 * wave * C + ((half * NJ + j) * 64 + lane) * 8);
    // (the tables are loop invariants: without the opaque copies of their addresses the compiler keeps all 120 registers of
    // them live through the three passes of every tile -- 207 spilled dwords)
    const float* gp = gamma;
    const float* sp = shift;
    asm volatile("" : "+s"(gp), "+s"(sp));
    f32x4 ga[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      ga[j][0] = *reinterpret_cast<const f32x4*>(gp + (sub + 8 * j) * 8);
      ga[j][1] = *reinterpret_cast<const f32x4*>(gp + (sub + 8 * j) * 8 + 4);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 16 * wave + 8 * half + (lane >> 3);
      const float* sh = sp + (int64_t)((16 * wave + 8 * half + (lane >> 3)) & (F - 1)) * p.ld_shift;   // (tiles start at frame 0)
      f32x4 sv[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        sv[j][0] = *reinterpret_cast<const f32x4*>(sh + (sub + 8 * j) * 8);
        sv[j][1] = *reinterpret_cast<const f32x4*>(sh + (sub + 8 * j) * 8 + 4);
      }
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
        for (int e = 0; e < 8; ++e) o[e] = (f16)fmaf(v[j][e] * rstd, ga[j][e >> 2][e & 3], sv[j][e >> 2][e & 3]);
        *reinterpret_cast<f16x8*>(panel + row * C + ((ch ^ (row & 7)) * 8)) = o;
      }
    }
  };

  // ---- wave = head: projections with a panel as one operand and the head's weight fragments (global) as the other
  // (buffer loads: lane offset in ONE register, the fragment's place in the scalar offset -- as flat loads the 90 fragment
  // addresses of a tile are loop invariants that the compiler hoists as 64-bit pairs: 180 registers, spilled)
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, H * PARTS * DP * C * 2, 0x00020000);
  const int w_lane = lane * 16, w_wave = wave * (PARTS * DP * C * 2);
  const int sw = l15 & 7;
  // (chunk 4 s + g of a K step swizzled by the row: ((4 s + g) ^ sw) = 8 (s >> 1) + ((4 (s & 1) + g) ^ sw) -- TWO lane-dependent offsets
  //  and an immediate instead of one hoisted register per K step)
  const int swz[2] = {(g ^ sw) * 8, ((4 + g) ^ sw) * 8};
  f32x4 acc[MA_PIX][DT];
  auto zero_init = [](const int, const int) { return f32x4{0.f, 0.f, 0.f, 0.f}; };
  // (wp: byte offset of the wave's fragments of this pass in `rs`; (DP C = KS DT 512 halfs per part) + (s DT + t) 1024)
  auto no_hook = []() {};
  auto project = [&](const f16* panel, const __amdgpu_buffer_rsrc_t rs, const int wp, auto transposed, auto init, auto hook) {
    constexpr bool TR = decltype(transposed)::value;
    auto ldw = [&](const int s, const int t) {
      return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, w_lane, wp + (s * DT + t) * 1024, 0));
    };
    const f16* alane = panel + l15 * C;                  // + (16 pix) C + (((4 s + g) ^ (l15 & 7)) * 8)
    f16x8 wf[MA_PD][DT];
#pragma unroll
    for (int pix = 0; pix < MA_PIX; ++pix)
#pragma unroll
      for (int t = 0; t < DT; ++t) acc[pix][t] = init(pix, t);
#pragma unroll
    for (int s = 0; s < MA_PD - 1; ++s)
#pragma unroll
      for (int t = 0; t < DT; ++t) wf[s][t] = ldw(s, t);
    // (r5: without this fence the scheduler may pair each of these loads with the first MFMA that takes it -- load, s_waitcnt vmcnt(0),
    //  MFMA, three L2 round trips in series at the top of the pass: seen in the v pass of the OUTP form)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int AD = MA_AD, NI = KS * MA_PIX;
    auto lda = [&](const int i) {
      return *reinterpret_cast<const f16x8*>(alane + 16 * (i % MA_PIX) * C + 64 * ((i / MA_PIX) >> 1) + swz[(i / MA_PIX) & 1]);
    };
    f16x8 af[AD + 1];
#pragma unroll
    for (int i = 0; i < AD; ++i) af[i] = lda(i);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int s = i / MA_PIX, pix = i % MA_PIX;
      if (pix == 0 && s + MA_PD - 1 < KS) {
#pragma unroll
        for (int t = 0; t < DT; ++t) wf[(s + MA_PD - 1) % MA_PD][t] = ldw(s + MA_PD - 1, t);
      }
#ifndef I2V_MA_HOOK_S
#define I2V_MA_HOOK_S 0
#endif
      if (pix == 0 && s == I2V_MA_HOOK_S) hook();       // (memory requests that may return behind the K steps' fragments requested so far)
      if (i + AD < NI) af[(i + AD) % (AD + 1)] = lda(i + AD);
#pragma unroll
      for (int t = 0; t < DT; ++t)
        acc[pix][t] = TR ? mfma16x16x32(wf[s % MA_PD][t], af[i % (AD + 1)], acc[pix][t])
                         : mfma16x16x32(af[i % (AD + 1)], wf[s % MA_PD][t], acc[pix][t]);
      __builtin_amdgcn_sched_barrier(0);      // (unfenced, the scheduler hoists every load of the pass to its top: 947 spills)
    }
  };
  auto to_half = [](const f32x4 a) { return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]}; };
  // Output row of one query: lane (g, l15) holds channels 16 t + 4 g .. + 3 of query l15, 8 bytes per tile.  v_permlane16_swap
  // pairs the lane groups g, g ^ 1 so that even groups end with 8 consecutive channels of tile 0 and odd groups with 8 of tile 1
  // -- one 16-byte store per lane for two tiles (as 8-byte stores the output cost 6.5k of a tile's 50k cycles: issue-bound).
  static_assert(DT == 3 && D == 40, "store pattern of three 16-channel tiles holding 40 channels");
  auto store_tiles = [&](f16* orow, const int g, const u32x2 (&oh)[DT]) {      // (g: from an opaque copy of the lane index)
    u32x2 a = oh[0], b = oh[1], c2 = oh[2], d2 = oh[2];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                 : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(c2[0]), "+v"(d2[0]), "+v"(c2[1]), "+v"(d2[1]));
    const u32x4 v01 = {a[0], a[1], b[0], b[1]};                  // even g: tile 0 channels 4 g .. + 7; odd g: tile 1 channels 4 (g - 1) ..
    const u32x4 v2 = {c2[0], c2[1], d2[0], d2[1]};               // g = 0: tile 2 channels 0 .. 7 (32 .. 39 of the head)
#ifdef I2V_MA_NOSTORE
    if (p.ldo < 0)
#endif
    {
      *reinterpret_cast<u32x4*>(orow + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g)) = v01;
      if (g == 0) *reinterpret_cast<u32x4*>(orow + 32) = v2;
    }
  };

  // ---- OUTP: the tile's o into the panel (every wave its head's D channels of all rows, at the panel's swizzled places; lane (g, l15)
  // holds channels 16 t + 4 g .. + 3 of row l15: one 8-byte write per tile), then the fourth pass: out^T[channel][row] = Wo (A operand,
  // the wave's 40 output channels in the fragment order of the q part) x o^T (B operand from the panel), accumulators starting from
  // the bias, the residual added at the end.
  static_assert(D % 8 == 0, "a head's channels are whole 16-byte chunks of a panel row");
  auto stash_o = [&](f16* pw, const int pix, const u32x2 (&oh)[DT]) {
    const int ln = ma_opaque(lane), g2 = ln >> 4, r15 = ln & 15;
    f16* dst = pw + (16 * pix + r15) * C + 4 * (g2 & 1);
#pragma unroll
    for (int t = 0; t < DT; ++t)        // (the padding channels D .. DP of the last tile belong to the next head)
      if (16 * t + 4 * g2 < D) *reinterpret_cast<u32x2*>(dst + (((wave * (D / 8) + 2 * t + (g2 >> 1)) ^ (r15 & 7)) * 8)) = oh[t];
  };
  auto out_project = [&](const f16* pr, const int tile) {
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_o), 0, H * DP * C * 2, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.b_o), 0, C * 4, 0x00020000);
    const f16* base = X + (int64_t)tile * (MA_PIX * MA_F) * p.ldx;                      // (wave-uniform)
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)(((MA_PIX * MA_F - 1) * p.ldx + C) * 2), 0x00020000);
    const int ln = ma_opaque(lane), sg = ln >> 4, sl15 = ln & 15;
    f32x4 bo[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
      bo[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs_b, 16 * t + 4 * sg < D ? (unsigned)((wave * D + 16 * t + 4 * sg) * 4) : 0x80000000u, 0, 0));
    // the residual: this lane's slice of its rows (8 bytes per row and tile), requested BEHIND the pass's first weight fragments
    // (loads return in order: in front of them the pass would start with an HBM round trip) and added to the finished accumulators
    u32x2 res[MA_PIX][DT];
    project(pr, rs_o, wave * (DP * C * 2), std::true_type{}, [&](const int, const int t) { return bo[t]; }, [&]() {
#pragma unroll
      for (int pix = 0; pix < MA_PIX; ++pix)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
#ifdef I2V_MA_NORES
          res[pix][t] = u32x2{0u, 0u};
#else
          const unsigned voff = 16 * t + 4 * sg < D ? (unsigned)((((16 * pix + sl15) * p.ldx) + wave * D + 16 * t + 4 * sg) * 2) : 0x80000000u;
          res[pix][t] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, voff, 0, 0));
#endif
        }
    });
    f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + (int64_t)tile * (MA_PIX * MA_F) * p.ldo + wave * D;
#pragma unroll
    for (int pix = 0; pix < MA_PIX; ++pix) {
      u32x2 oh[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const f16x4 r = __builtin_bit_cast(f16x4, res[pix][t]);
        const f32x4 v = {acc[pix][t][0] + (float)r[0], acc[pix][t][1] + (float)r[1], acc[pix][t][2] + (float)r[2], acc[pix][t][3] + (float)r[3]};
        oh[t] = __builtin_bit_cast(u32x2, to_half(v));
      }
      store_tiles(O + (int64_t)(16 * pix + sl15) * p.ldo, sg, oh);
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;        // (workgroup-uniform)
  MA_STAMP(0);
  fetch_rows(tile, panels);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  normalise_rows(panels);
  MA_STAMP(1);
  lds_barrier();
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    f16* panel = panels + (it & 1) * (MA_PIX * MA_F * C);
    const int next = tile + (int)gridDim.x;
    // (the other panel was last read in the previous iteration, which every wave has left through the barrier at its end)
    f16* other = panels + ((it + 1) & 1) * (MA_PIX * MA_F * C);
    MA_STAMP(2);
    if constexpr (CROSS) {      // one pass only: the next tile's rows leave HBM now and land under it (see below)
      if (next < ntiles) fetch_rows(next, other);
      __builtin_amdgcn_sched_barrier(0);
    }

    // q^T [channel][frame]
    f16x4 qh[MA_PIX][DT];
    project(panel, rs_w, w_wave, std::true_type{}, zero_init, no_hook);
    // (CROSS: softmax scale * log2 e goes into the fp16 q, as in i2v_attention_f16 -- 24 multiplies per tile instead of 160)
#pragma unroll
    for (int pix = 0; pix < MA_PIX; ++pix)
#pragma unroll
      for (int t = 0; t < DT; ++t) qh[pix][t] = to_half(CROSS ? acc[pix][t] * scale_log2 : acc[pix][t]);
    MA_STAMP(3);
    if constexpr (CROSS && OUTP) lds_barrier();      // the q pass was the panel's last reader: every wave has left it

    if constexpr (CROSS) {
      // ---- cross-attention against the resident context: key fragments K[key][channel] (A operand of S^T) and value
      // fragments V^T[channel][key] (A operand of O^T), 8 bytes per lane each
      const int ctx = tile / p.tiles_per_ctx;
      // (packed by the caller in FRAGMENT order, zero beyond the context's length and the head's width: 30 loads of 512
      // contiguous bytes per wave.  Gathered from row-major K / V^T -- 16 rows x 32 B per instruction -- they took 5.5k cycles
      // of a 48k-cycle tile.)
      const f16* cf = reinterpret_cast<const f16*>(p.ctx_frag) + ((int64_t)ctx * H + wave) * (2 * MA_KT * DT * 256) + lane * 4;
      f16x4 kf[MA_KT][DT], vf[DT][MA_KT];
#pragma unroll
      for (int kt = 0; kt < MA_KT; ++kt)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          kf[kt][t] = *reinterpret_cast<const f16x4*>(cf + (kt * DT + t) * 256);
          vf[t][kt] = *reinterpret_cast<const f16x4*>(cf + (MA_KT * DT + t * MA_KT + kt) * 256);
        }
      // decoupled image cross-attention of the IP-Adapter (softmax(q K_ip^T) V_ip over <= 16 image tokens, added with
      // ip_scale): the first key tile of a second fragment set in the same layout
      const bool has_ip = p.ip_len > 0;             // (workgroup-uniform)
      f16x4 kfi[DT], vfi[DT];
      if (has_ip) {
        const f16* ci = reinterpret_cast<const f16*>(p.ip_frag) + ((int64_t)ctx * H + wave) * (2 * MA_KT * DT * 256) + lane * 4;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          kfi[t] = *reinterpret_cast<const f16x4*>(ci + t * 256);
          vfi[t] = *reinterpret_cast<const f16x4*>(ci + (MA_KT * DT + t * MA_KT) * 256);
        }
      }
      MA_STAMP(4);
      MA_STAMP(5);
      f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + (int64_t)tile * (MA_PIX * MA_F) * p.ldo + wave * D;
      const int sln = ma_opaque(lane), sg = sln >> 4, sl15 = sln & 15;
#pragma unroll
      for (int pix = 0; pix < MA_PIX; ++pix) {
        float sv[MA_KT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MA_KT; ++kt) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < DT; ++t) sacc = mfma16x16x16(kf[kt][t], qh[pix][t], sacc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (contexts of 65 .. 80 tokens -- CLIP's 77 -- end in the last key tile: only its scores need the mask; a shorter
            // one is masked in every tile by the same test, resolved per tile at run time)
            const bool live = (kt < MA_KT - 1 && p.lt > 16 * (MA_KT - 1)) || 16 * kt + 4 * g + r < p.lt;
            sv[kt][r] = live ? sacc[r] : -INFINITY;
            mx = fmaxf(mx, sv[kt][r]);
          }
        }
        mx = lane_xor32_max(lane_xor16_max(mx));
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < MA_KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[kt][r] = __builtin_amdgcn_exp2f(sv[kt][r] - mx);
            ls += sv[kt][r];
          }
        ls = lane_xor32_sum(lane_xor16_sum(ls));
        const float inv = 1.0f / ls;                 // applied to O (12 values), not to P (20): P in (0, 1] as it is
        f16x4 pk[MA_KT];
#pragma unroll
        for (int kt = 0; kt < MA_KT; ++kt) pk[kt] = f16x4{(f16)sv[kt][0], (f16)sv[kt][1], (f16)sv[kt][2], (f16)sv[kt][3]};
        f32x4 ov[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < MA_KT; ++kt) o = mfma16x16x16(vf[t][kt], pk[kt], o);
          ov[t] = o * inv;
        }
        if (has_ip) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < DT; ++t) sacc = mfma16x16x16(kfi[t], qh[pix][t], sacc);
          float si[4];
          float mi = -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            si[r] = (4 * g + r < p.ip_len) ? sacc[r] : -INFINITY;
            mi = fmaxf(mi, si[r]);
          }
          mi = lane_xor32_max(lane_xor16_max(mi));
          float li = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            si[r] = __builtin_amdgcn_exp2f(si[r] - mi);
            li += si[r];
          }
          li = lane_xor32_sum(lane_xor16_sum(li));
          const float wi = p.ip_scale / li;
          const f16x4 pi = {(f16)si[0], (f16)si[1], (f16)si[2], (f16)si[3]};
#pragma unroll
          for (int t = 0; t < DT; ++t) ov[t] += mfma16x16x16(vfi[t], pi, f32x4{0.f, 0.f, 0.f, 0.f}) * wi;
        }
        u32x2 oh[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) oh[t] = __builtin_bit_cast(u32x2, to_half(ov[t]));
        if constexpr (OUTP)
          stash_o(panel, pix, oh);
        else
          store_tiles(O + (int64_t)(16 * pix + sl15) * p.ldo, sg, oh);
      }
      MA_STAMP(6);
      if (next < ntiles) normalise_rows(other);
      if constexpr (OUTP) {
        lds_barrier();                               // o of every head is in the panel
        out_project(panel, tile);
      }
    } else {
    // k^T, then S^T[key][query] and the softmax over the keys.  NB = 16-row tiles per pixel (F = 32: two -- the keys of a query
    // are the 8 accumulator rows of two score blocks and the 4 lane groups); ph[key tile][query tile of the same pixel]
    constexpr int NB = F == 32 ? 2 : 1;
    f16x4 ph[MA_PIX][NB];
    project(panel, rs_w, w_wave + DP * C * 2, std::true_type{}, zero_init, no_hook);
#pragma unroll
    for (int px = 0; px < MA_PIX; px += NB) {
#pragma unroll
      for (int jq = 0; jq < NB; ++jq) {
        float sv[NB][4];
        float mx = -INFINITY;
#pragma unroll
        for (int ik = 0; ik < NB; ++ik) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < DT; ++t) sacc = mfma16x16x16(to_half(acc[px + ik][t]), qh[px + jq][t], sacc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[ik][r] = sacc[r] * scale_log2;
            // F = 8: key row 4 g + r and query column l15 of a tile belong to the same pixel iff their bit 3 agrees
            if (F == 8 && (g >> 1) != (l15 >> 3)) sv[ik][r] = -INFINITY;
            mx = fmaxf(mx, sv[ik][r]);
          }
        }
        mx = lane_xor32_max(lane_xor16_max(mx));
        float ls = 0.f;
#pragma unroll
        for (int ik = 0; ik < NB; ++ik)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[ik][r] = __builtin_amdgcn_exp2f(sv[ik][r] - mx);
            ls += sv[ik][r];
          }
        ls = lane_xor32_sum(lane_xor16_sum(ls));
        const float inv = 1.0f / ls;
#pragma unroll
        for (int ik = 0; ik < NB; ++ik)
          ph[px + ik][jq] = f16x4{(f16)(sv[ik][0] * inv), (f16)(sv[ik][1] * inv), (f16)(sv[ik][2] * inv), (f16)(sv[ik][3] * inv)};
      }
    }
    MA_STAMP(4);

    // the next tile's rows leave HBM now and land (in the other panel, which nobody reads any more) under the v pass.  No wait
    // of their own: loads return in order, and the v pass below has waited for weight fragments it requested after them.
    if (next < ntiles) fetch_rows(next, other);
    __builtin_amdgcn_sched_barrier(0);

    // v [key][channel], O^T[channel][query], stored as o[query row][head channels]
    project(panel, rs_w, w_wave + 2 * (DP * C * 2), std::false_type{}, zero_init, no_hook);
    MA_STAMP(5);
    if constexpr (OUTP) lds_barrier();               // the v pass was the panel's last reader: every wave has left it
    f16* __restrict__ O = reinterpret_cast<f16*>(p.out) + (int64_t)tile * (MA_PIX * MA_F) * p.ldo + wave * D;
    const int sln = ma_opaque(lane), sg = sln >> 4, sl15 = sln & 15;
#pragma unroll
    for (int px = 0; px < MA_PIX; px += NB)
#pragma unroll
      for (int jq = 0; jq < NB; ++jq) {
        u32x2 oh[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ik = 0; ik < NB; ++ik) o = mfma16x16x16(to_half(acc[px + ik][t]), ph[px + ik][jq], o);
          oh[t] = __builtin_bit_cast(u32x2, to_half(o));
        }
        if constexpr (OUTP)
          stash_o(panel, px + jq, oh);
        else
          store_tiles(O + (int64_t)(16 * (px + jq) + sl15) * p.ldo, sg, oh);
      }
    MA_STAMP(6);
    if (next < ntiles) normalise_rows(other);
    if constexpr (OUTP) {
      lds_barrier();                                 // o of every head is in the panel
      out_project(panel, tile);
    }
    }
    MA_STAMP(7);
#ifdef I2V_MA_STAMPS
    if (stamps != nullptr && lane == 0 && it < 4) {
      long long* st = stamps + (((int64_t)blockIdx.x * 4 + it) * H + wave) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) st[k] = stamp[k];
    }
#endif
    lds_barrier();       // (LDS only: the output stores stay in flight across it)
  }
#undef MA_STAMP
}

template <int C, int D, int H, bool CROSS, int F = 16, bool OUTP = false>
int ma_cus() {       // CUs of the current device once it has granted this kernel its two panels of LDS; 0: refused (runtime.hip)
  return i2v_big_lds_kernel_cus(reinterpret_cast<const void*>(motion_attn_kernel<C, D, H, CROSS, F, OUTP>), 2 * (size_t)MA_PIX * MA_F * C * sizeof(f16));
}

template <int C, int D, int H, bool CROSS, int F = 16, bool OUTP = false>
int launch_ma(const ma_args& p, int64_t rows, float scale, hipStream_t s, const char* what) {
  const size_t lds = 2 * (size_t)MA_PIX * MA_F * C * sizeof(f16);
  const int cus = ma_cus<C, D, H, CROSS, F, OUTP>();
  if (cus <= 0) I2V_FAIL(I2V_ERR_UNSUPPORTED, "%s: %zu bytes of LDS refused by this device", what, lds);
  const int ntiles = (int)(rows / (MA_PIX * MA_F));
  // one workgroup per CU (160 KB of LDS each), every workgroup the same number of tiles where the count allows it
  const int grid = i2v_persistent_grid(ntiles, cus);
  long long* stamps = nullptr;
#ifdef I2V_MA_STAMPS
  stamps = getenv("I2V_MA_STAMP_PTR") ? reinterpret_cast<long long*>(strtoull(getenv("I2V_MA_STAMP_PTR"), nullptr, 0)) : nullptr;
#endif
  hipLaunchKernelGGL((motion_attn_kernel<C, D, H, CROSS, F, OUTP>), dim3((unsigned)grid), dim3(64 * H), lds, s, p, scale * 1.4426950408889634f,
                     ntiles, stamps);
  return i2v_check_launch(what);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int32_t i2v_motion_attn_supported(int64_t rows, int32_t channels, int32_t heads, int32_t head_dim, int32_t frames) {
  return rows > 0 && rows % (MA_PIX * MA_F) == 0 && rows / (MA_PIX * MA_F) < (1 << 24) && channels == 320 && heads == 8 &&
         head_dim == 40 &&
         ((frames == 16 && ma_cus<320, 40, 8, false, 16>() > 0) || (frames == 8 && ma_cus<320, 40, 8, false, 8>() > 0) ||
          (frames == 32 && ma_cus<320, 40, 8, false, 32>() > 0));
}

extern "C" int32_t i2v_motion_attn_pack_rows(int32_t heads, int32_t head_dim) { return heads * 3 * ((head_dim + 15) / 16) * 16; }

extern "C" int i2v_motion_attn_f16(const i2v_motion_attn_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_motion_attn_f16: null params");
  const i2v_motion_attn_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.shift && p.w_qkv && p.out, "i2v_motion_attn_f16: null pointer");
  I2V_CHECK_ARG(i2v_motion_attn_supported(p.rows, p.channels, p.heads, p.head_dim, p.frames),
                "i2v_motion_attn_f16: rows %lld channels %d heads %d head_dim %d frames %d is not a fused shape "
                "(i2v_motion_attn_supported)", (long long)p.rows, p.channels, p.heads, p.head_dim, p.frames);
  I2V_CHECK_ARG(p.ldx >= p.channels && p.ldx % 8 == 0 && p.ldo >= p.channels && p.ldo % 8 == 0 && p.ld_shift >= p.channels &&
                p.ld_shift % 4 == 0, "i2v_motion_attn_f16: row strides");
  I2V_CHECK_ARG(al16(p.x) && al16(p.gamma) && al16(p.shift) && al16(p.w_qkv) && al16(p.out),
                "i2v_motion_attn_f16: pointers must be 16-byte aligned");
  ma_args a = {};
  a.x = p.x; a.ldx = p.ldx; a.gamma = p.gamma; a.shift = p.shift; a.ld_shift = p.ld_shift; a.w = p.w_qkv; a.out = p.out;
  a.ldo = p.ldo; a.eps = p.eps;
  // (OUTP: the residual rows go through a buffer descriptor and per-lane offsets of 32 bits)
  I2V_CHECK_ARG(p.w_o == nullptr || p.ldx < (1 << 23), "i2v_motion_attn_f16: ldx (%lld) must be below 2^23 with the out-projection", (long long)p.ldx);
  I2V_CHECK_ARG((p.w_o == nullptr) == (p.b_o == nullptr) && al16(p.w_o) && al16(p.b_o),
                "i2v_motion_attn_f16: w_o and b_o come together, 16-byte aligned");
  a.w_o = p.w_o; a.b_o = p.b_o;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.w_o != nullptr) {
    if (p.frames == 8) return launch_ma<320, 40, 8, false, 8, true>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
    if (p.frames == 32) return launch_ma<320, 40, 8, false, 32, true>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
    return launch_ma<320, 40, 8, false, 16, true>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
  }
  if (p.frames == 8) return launch_ma<320, 40, 8, false, 8>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
  if (p.frames == 32) return launch_ma<320, 40, 8, false, 32>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
  return launch_ma<320, 40, 8, false, 16>(a, p.rows, p.scale, s, "i2v_motion_attn_f16");
}

extern "C" int32_t i2v_cross_attn_fused_supported(int64_t rows, int32_t channels, int32_t heads, int32_t head_dim, int32_t ctx_len,
                                                  int64_t rows_per_ctx) {
  return rows > 0 && rows % (MA_PIX * MA_F) == 0 && rows / (MA_PIX * MA_F) < (1 << 24) && channels == 320 && heads == 8 &&
         head_dim == 40 && ctx_len >= 1 && ctx_len <= 16 * MA_KT && rows_per_ctx > 0 && rows_per_ctx % (MA_PIX * MA_F) == 0 &&
         rows % rows_per_ctx == 0 && ma_cus<320, 40, 8, true>() > 0;
}

extern "C" int32_t i2v_cross_attn_fused_pack_rows(int32_t heads, int32_t head_dim) { return heads * ((head_dim + 15) / 16) * 16; }

extern "C" int64_t i2v_cross_attn_fused_ctx_elems(int32_t n_ctx, int32_t heads, int32_t head_dim) {
  return (int64_t)n_ctx * heads * 2 * MA_KT * ((head_dim + 15) / 16) * 256;
}

extern "C" int i2v_cross_attn_fused_f16(const i2v_cross_attn_fused_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_cross_attn_fused_f16: null params");
  const i2v_cross_attn_fused_params& p = *pp;
  I2V_CHECK_ARG(p.x && p.gamma && p.beta && p.w_q && p.ctx_frag && p.out, "i2v_cross_attn_fused_f16: null pointer");
  I2V_CHECK_ARG(i2v_cross_attn_fused_supported(p.rows, p.channels, p.heads, p.head_dim, p.ctx_len, p.rows_per_ctx),
                "i2v_cross_attn_fused_f16: rows %lld channels %d heads %d head_dim %d ctx_len %d rows_per_ctx %lld is not a fused shape "
                "(i2v_cross_attn_fused_supported)", (long long)p.rows, p.channels, p.heads, p.head_dim, p.ctx_len,
                (long long)p.rows_per_ctx);
  I2V_CHECK_ARG(p.ldx >= p.channels && p.ldx % 8 == 0 && p.ldo >= p.channels && p.ldo % 8 == 0, "i2v_cross_attn_fused_f16: row strides");
  I2V_CHECK_ARG(al16(p.x) && al16(p.gamma) && al16(p.beta) && al16(p.w_q) && al16(p.out) && al16(p.ctx_frag) && al16(p.ip_frag),
                "i2v_cross_attn_fused_f16: pointers must be 16-byte aligned");
  I2V_CHECK_ARG(p.ip_frag == nullptr || (p.ip_len >= 1 && p.ip_len <= 16), "i2v_cross_attn_fused_f16: ip_len (%d) must be in [1, 16]",
                p.ip_len);
  ma_args a = {};
  a.x = p.x; a.ldx = p.ldx; a.gamma = p.gamma; a.shift = p.beta; a.ld_shift = 0; a.w = p.w_q; a.out = p.out; a.ldo = p.ldo;
  a.eps = p.eps;
  a.ctx_frag = p.ctx_frag; a.lt = p.ctx_len;
  a.ip_frag = p.ip_frag; a.ip_len = p.ip_frag ? p.ip_len : 0; a.ip_scale = p.ip_scale;
  a.tiles_per_ctx = (int32_t)(p.rows_per_ctx / (MA_PIX * MA_F));
  // (OUTP: the residual rows go through a buffer descriptor and per-lane offsets of 32 bits)
  I2V_CHECK_ARG(p.w_o == nullptr || p.ldx < (1 << 23), "i2v_cross_attn_fused_f16: ldx (%lld) must be below 2^23 with the out-projection", (long long)p.ldx);
  I2V_CHECK_ARG((p.w_o == nullptr) == (p.b_o == nullptr) && al16(p.w_o) && al16(p.b_o),
                "i2v_cross_attn_fused_f16: w_o and b_o come together, 16-byte aligned");
  a.w_o = p.w_o; a.b_o = p.b_o;
  if (p.w_o != nullptr)
    return launch_ma<320, 40, 8, true, 16, true>(a, p.rows, p.scale, reinterpret_cast<hipStream_t>(stream), "i2v_cross_attn_fused_f16");
  return launch_ma<320, 40, 8, true>(a, p.rows, p.scale, reinterpret_cast<hipStream_t>(stream), "i2v_cross_attn_fused_f16");
}
