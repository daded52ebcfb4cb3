// Large-tile MFMA GEMM / implicit-GEMM conv for gfx950: BM x 320 x 64 tiles, 8 waves (2 x 4, per-wave (BM/2) x 80),
// operands staged by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS with no VGPR / ds_write pass), two LDS stages,
// the next K tile's DMA in flight under the current tile's MFMAs, one raw s_barrier per K tile, and the two waves of
// each SIMD running half a K tile out of phase so that one wave's LDS reads overlap the other's MFMAs.
//
// Why this shape (measured in round 1, profiles/r1_tile_sweep.txt): the 128-wide register-staged kernel of gemm.hip is
// bound by the LDS write path (ds_write_b128 ~79 B/clk/CU) and by L2->CU operand bandwidth (43-65 FLOP/B per tile vs
// the ~95 B/clk a CU's MFMA pipes consume); a 256 x 320 tile needs 146 FLOP per staged byte and LDS-DMA removes the
// ds_write pass.  Every channel count of the SD-1.5 topology is a multiple of 320 (320/640/960/1280/2560/5120/10240),
// so BN = 320 = 4 waves x 5 MFMA columns tiles N with no padding.
// LDS image: rows of 128 B, chunk c of row r at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments).  LDS-DMA
// writes lane-linearly (wave base + 16 * lane), so the swizzle is applied on the SOURCE side: lane l of the
// instruction for 8-row group j fetches logical chunk (l & 7) ^ ((row >> 1) & 7) of row 8 j + (l >> 3) (guide rule 21).
// Masked lanes (row >= M, conv halo) carry an out-of-range buffer offset: LDS-DMA cannot skip a lane, but the
// descriptor's range check makes it write zeros.
#include <cstdlib>
#include <utility>

#include "gemm_common.h"

namespace {

constexpr int BIG_BN = 320;

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane, global -> LDS at (wave-uniform LDS base) + 16 * lane; a lane
// whose offset fails the descriptor's range check writes zeros.  (A plain function on purpose: called directly from the
// kernel template, the builtin makes hipcc's host pass drop the kernel's launch stub without a diagnostic.)
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0,
                                           0);
}

// (Tried: an L2 prefetch of the A rows two / three K tiles ahead -- a 4-byte-per-lane LDS-DMA whose lanes touch one word
// of 64 different cache lines -- left in flight across the barrier with vmcnt(1).  The GEMM class got 3 % SLOWER
// (29.6 -> 30.5 ms per step): the cold-A projections are bound by HBM bandwidth, not by first-touch latency.)
// m / d for 0 <= m < 2^24 (row indices): one multiply by the precomputed reciprocal and a +-1 fix-up instead of the
// ~35-instruction integer division (the epilogue runs it once per 8-column task)
__device__ __forceinline__ int fast_div(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  const int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// x(l) + x(l ^ mask) through ds_bpermute with the partner's address passed in (derived by the caller from an opaque
// copy of the lane id at the point of use: __shfl_xor's own lane-id registers get hoisted in front of the tile loop and
// stay live through the accumulator-bound K loop)
__device__ __forceinline__ float xor_sum(float x, int partner_addr) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, x)));
}
// a wave-uniform float held in a scalar register (the compiler otherwise keeps such loop invariants in VGPRs)
__device__ __forceinline__ float uniform_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

#ifndef I2V_CONV_COLMAJOR
#define I2V_CONV_COLMAJOR 1
#endif
#ifndef I2V_GEMM_PANEL
#define I2V_GEMM_PANEL 8
#endif

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
//
// Pipeline depth: NS LDS stages of BK-deep tiles, NS - 1 tiles' DMA in flight.  (BK, NS) = (64, 2) keeps one 72 KiB
// tile in flight; (32, 4) keeps three 36 KiB tiles (1.5x the bytes, issued in finer steps) for the same 144 KiB.
// The LDS image is made of 128-byte units: one row of a BK = 64 tile, or two consecutive rows of a BK = 32 tile
// (chunks 0-3 = even row, 4-7 = odd row); unit u stores chunk c at c ^ ((u >> 1) & 7).
template <int BK>
__device__ __forceinline__ int big_lds_addr(int row, int kchunk) {
  const int u = BK == 64 ? row : row >> 1;
  const int c8 = BK == 64 ? kchunk : ((row & 1) << 2) + kchunk;
  return u * 128 + ((c8 ^ ((u >> 1) & 7)) << 4);
}

// LNF: LayerNorm folded into the epilogue (a template parameter: as a runtime branch the 160 accumulators of the two
// paths met in PHI nodes and the 256-row kernels spilled 170-290 registers).
// FAST: plain A from one source with M % BM == 0 -- no ragged rows, no second source, so a K tile's DMA is a fixed list of
// instructions with scalar offsets; these problems (every transformer GEMM of the step) run the persistent tile loop.
// Everything else (conv, split-K, ragged M, channel-concatenated A) runs one tile per workgroup.
// BNT: columns per tile, 320 (every channel count of the SD-1.5 UNet) or 256 / 128 (the VAE's 128 / 256 / 512 channels:
// 4 N-waves x 64 or 32 columns); the LayerNorm fold and split-K exist for 320 only.
// NW: waves per workgroup.  8 = 2 (M) x 4 (N) waves, one workgroup per CU (the two waves of a SIMD belong to the same tile
// and run staggered).  4 = 1 x 4 waves with a 128-row tile and 32-deep stages: 62 KiB of LDS and 256 registers per wave, so
// TWO workgroups share a CU -- the two waves of a SIMD then belong to DIFFERENT output tiles at unrelated phases, and one
// tile's prologue (first DMA wait) and epilogue (LayerNorm apply, GELU, transposes, stores: as long as the K loop itself
// when K = 320) run beside the other tile's MFMAs instead of stopping the CU.
template <int BM, int BK, int NS, int AMODE, int EPI, int STORE, bool SPLIT = false, bool STAGGER = true, bool LNF = false,
          bool FAST = false, int BNT = 320, int NW = 8, bool GNS = false, bool HILO = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_big_kernel(const i2v_gemm_params p, const int tiles_n,
                                                                            const int kps, const int ntiles) {
  static_assert(NS == 2 || (NS <= 4 && !FAST && !SPLIT && NW == 8),
                "the cross-tile prefetch of the persistent tile loop, the LayerNorm fold and split-K are written for two stages");
  static_assert(NW == 8 || (NW == 4 && !STAGGER && !SPLIT), "4-wave workgroups: one M-wave, no stagger partner, no split-K");
  constexpr int BN = BNT;
  constexpr int WNC = BN / 4;               // columns per N-wave: 80 / 64 / 32
  constexpr int WM = BM / (NW / 4), MI = WM / 16, NI = WNC / 16;
  static_assert(BN % 64 == 0 && (BN == 320 || (!LNF && !SPLIT)), "the LayerNorm fold and split-K are written for BN = 320");
  constexpr int AG = BM * BK / (512 * NW);  // 1 KiB (8-unit) A groups per wave: BM * BK * 2 / 1024 groups over NW waves
  constexpr int WGT = BN * BK * 2 / 1024;   // W groups per tile (40 or 20)
  constexpr int WG = (WGT + NW - 1) / NW;   // per wave (the last round may cover only the first waves)
  static_assert(AG >= 1 && (BM * BK) % (512 * NW) == 0, "the A tile must be whole 1 KiB groups per wave");
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int KSTEPS = BK / 32;
  // + 1 KiB that swallows the DMA of W groups past the tile (every wave issues the same count: one vmcnt for all)
  // LNF: + 4 KiB holding this tile's LayerNorm row statistics (BM x (mean, rstd)) and the 320 weight row sums.  The
  // statistics are computed IN the K loop from the A fragments every wave reads anyway (the loop streams whole rows of A
  // through the workgroup: K = the normalised width), so there is no statistics kernel and no extra pass over A; the row
  // sums are fetched before the first K tile's DMA, so the epilogue reads both from LDS instead of paying a global-load
  // round trip per tile (measured: +2.2 ms per step with the loads in the epilogue).
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE + 1024 + (LNF ? 2048 + 512 + 2 * 1280 : 0)];
  float2* const lds_st = reinterpret_cast<float2*>(smem + NS * STAGE + 1024);
  float* const lds_ws2 = reinterpret_cast<float*>(smem + NS * STAGE + 1024 + 2048 + 512);   // two buffers of BN sums

#if defined(I2V_PROBE) && I2V_PROBE == 5
  long long stamp[6], cstamp[6];
#define I2V_STAMP(i) stamp[i] = __builtin_amdgcn_s_memrealtime(); cstamp[i] = __builtin_amdgcn_s_memtime()
#else
#define I2V_STAMP(i)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, l15 = lane & 15;
  const int M = p.M, N = p.N, K = p.K;

  const f16* __restrict__ A = reinterpret_cast<const f16*>(p.a);
  const f16* __restrict__ A2 = reinterpret_cast<const f16*>(p.a2);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);
  const int ksp = (A2 != nullptr) ? p.k_split : K;

  // ---- DMA sources.  Every operand is addressed through a wave-uniform buffer descriptor with a 32-bit per-lane
  //      byte offset (slice extents < 2 GiB, K and k_split multiples of BK: checked on the host).  A lane that must
  //      contribute zeros (row >= M, conv halo) carries an out-of-range offset: the range check of buffer_load ... lds
  //      writes 0 to LDS for it (LDS-DMA cannot skip a lane).  Per K tile only the scalar offset changes, so the
  //      plain-GEMM prefetch costs no VALU work and no 64-bit address registers.
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSTEP = (BK == 64 ? 8 : 16) * NW;   // tile rows between a wave's consecutive 1 KiB groups
  const int lr = lane >> 3, lc = lane & 7;
  const int u0 = 8 * wave + lr;                // 128-byte unit of group 0; group i is unit u0 + 8 NW i, same swizzle
  const int c8 = lc ^ ((u0 >> 1) & 7);         // logical chunk this lane fetches (swizzle on the SOURCE side)
  const int r0 = BK == 64 ? u0 : 2 * u0 + (c8 >> 2);
  const int lane_k = (BK == 64 ? c8 : (c8 & 3)) * 8;   // k offset (halfs) of the chunk inside a K tile
  const int n_wbatch = p.rows_per_w > 0 ? M / p.rows_per_w : 1;
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(W), 0, (int)(((int64_t)(n_wbatch - 1) * p.w_batch_stride + (int64_t)(N - 1) * p.ldw + K) * 2), 0x00020000);
  const int64_t a_extent = AMODE == I2V_A_CONV3X3 ? ((int64_t)p.n_img * p.in_h * p.in_w - 1) * p.lda + p.cin
                                                  : (int64_t)(M - 1) * p.lda + ksp;
  const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, (int)(a_extent * 2), 0x00020000);
  const auto rs_a2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(A2 ? A2 : A), 0, A2 ? (int)(((int64_t)(M - 1) * p.lda2 + (K - ksp)) * 2) : 0, 0x00020000);
  // Plain A and W: the per-lane part of a DMA offset (row r0 of a 1 KiB group, chunk lane_k) does not depend on the
  // output tile; the tile's first row / column, the group's row step and the K tile all go into the instruction's SCALAR
  // offset.  So the K loop carries two address registers per lane whatever the tile, and the next tile's first K tile
  // can be issued with scalar arithmetic only.
  // a_perm (motion-module entry): output rows are (b, pixel, frame), A rows (b, frame, pixel).  A 1 KiB group holds 8
  // consecutive output rows; with frames | 64 the frame of a lane's row is r0 % frames for every group and tile, and its
  // pixel is (a scalar) + r0 / frames: the gather is a different per-lane constant plus a different scalar offset.
  const int pf = p.a_perm_frames;
  const int pf_shift = pf > 0 ? 31 - __builtin_clz(pf) : 0;
  const unsigned a_lane = pf > 0 ? (unsigned)((((r0 & (pf - 1)) * p.a_perm_hw + (r0 >> pf_shift)) * (int)p.lda + lane_k) * 2)
                                 : (unsigned)((r0 * (int)p.lda + lane_k) * 2);
  const unsigned w_lane = (unsigned)((r0 * (int)p.ldw + lane_k) * 2);
  // conv: output pixel of each of the lane's AG rows (first pixel of its image, oy, ox)
  auto conv_rows = [&](int tm0, int (&cp)[AG], int (&cy)[AG], int (&cx)[AG]) {
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int m = tm0 + r0 + RSTEP * i;
      const bool ok = m < M;
      const int ohw = p.out_h * p.out_w;
      const int mm = ok ? m : 0;
      const int img = mm / ohw, rem = mm - img * ohw;
      cp[i] = img * p.in_h * p.in_w;
      cy[i] = ok ? rem / p.out_w : -4;       // row >= M: every tap falls outside the image
      cx[i] = rem - (rem / p.out_w) * p.out_w;
    }
  };

  // DMA of K tile kt of the output tile at (tm0, tn0) into LDS stage `stage`.  conv: (cp, cy, cx) = conv_rows(tm0) and
  // (tap, ci) = the (tap, first channel) of that K tile, advanced here; cin % BK == 0, so a K tile never straddles taps.
  auto issue_at = [&](int kt, int stage, int tm0, int tn0, const int (&cp)[AG], const int (&cy)[AG], const int (&cx)[AG]) {
    char* sa = smem + stage * STAGE;
    char* sw = sa + BM * BK * 2;
    const int kb = kt * BK;
    if (AMODE == I2V_A_CONV3X3) {
      // (tap, first channel) of K tile kt, recomputed from kt (scalar arithmetic): carried as running state through the
      // lambdas it ended up in scratch memory -- two scratch loads + stores per issue inside the K loop, conv class
      // 12.5 -> 17.5 ms per step
      int tap, ci;
      if (p.conv_kblock) {   // channel-block-major: the 9 taps of a 64-channel block, then the next block
        tap = kt % 9;
        ci = (kt / 9) * BK;
      } else {               // tap-major; cin % BK == 0, so a K tile never straddles taps
        tap = kb / p.cin;
        ci = kb - tap * p.cin;
      }
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        int iy, ix;
        bool ok;
        if (p.upsample) {
          const int uy = cy[i] - 1 + dy, ux = cx[i] - 1 + dx;
          ok = (uy >= 0) && (ux >= 0) && (uy < p.out_h) && (ux < p.out_w);      // (out = 2 in, or 2 in - 1: forward_upsample_size)
          iy = uy >> 1;
          ix = ux >> 1;
        } else {
          iy = cy[i] * p.stride - (p.asym_pad ? 0 : 1) + dy;
          ix = cx[i] * p.stride - (p.asym_pad ? 0 : 1) + dx;
          ok = (iy >= 0) && (ix >= 0) && (iy < p.in_h) && (ix < p.in_w);
        }
        const unsigned voff = ok ? (unsigned)(((cp[i] + iy * p.in_w + ix) * (int)p.lda + ci + lane_k) * 2) : OOB;
        // (Round 3, timing experiment: with the A tile fetched for ONE tap in nine -- the traffic of a kernel that keeps an
        //  18 x 18 halo of the tile's input pixels in LDS and reads the nine taps from it -- the convolutions ran 4 - 10 %
        //  faster at 64 x 64, 12 - 18 % at 32 x 32, 4 - 13 % at 16 x 16 (tools/conv_ab.py): the A re-reads hit L2 and are not
        //  what bounds the K loop; fragment reads already keep LDS ~80 % busy at the MFMA rate.  The halo kernel was not built.)
        bdma16(rs_a, sa + (wave + NW * i) * 1024, voff, 0);
      }
    } else if (FAST || (kb < ksp && tm0 + BM <= M)) {
      int arow0 = tm0, astep = RSTEP;   // first A row of group 0, rows between groups
      if (pf > 0) {                     // (b, pixel, frame) rows tm0 .. : clip b, first pixel tp
        const int per = pf * p.a_perm_hw;
        const int tb = tm0 / per, tp = (tm0 - tb * per) >> pf_shift;
        arow0 = tb * per + tp;
        astep = RSTEP >> pf_shift;
      }
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        // (opaque scalar: otherwise the compiler folds the group's row step into AG per-lane offsets kept in registers
        // through the whole loop, which the accumulator-bound 256-row LayerNorm kernels cannot afford)
        int soff = ((arow0 + astep * i) * (int)p.lda + kb) * 2;
        asm volatile("" : "+s"(soff));
        bdma16(rs_a, sa + (wave + NW * i) * 1024, a_lane, soff);
      }
    } else {
      // last row tile of a ragged M (rows >= M carry the out-of-range offset: zero fill), or the second source of a
      // channel-concatenated A (skip connections).  The lane's row and chunk are re-derived here from an opaque copy of
      // the lane id, so that these rare forms keep no registers alive in the common loop.
      int ol = lane;
      asm volatile("" : "+v"(ol));
      const int ou0 = 8 * wave + (ol >> 3);
      const int oc8 = (ol & 7) ^ ((ou0 >> 1) & 7);
      const int orow = BK == 64 ? ou0 : 2 * ou0 + (oc8 >> 2);
      const int ok8 = (BK == 64 ? oc8 : (oc8 & 3)) * 8;
      const bool first = kb < ksp;
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const int m = tm0 + orow + RSTEP * i;
        const unsigned voff = m < M ? (unsigned)((m * (int)(first ? p.lda : p.lda2) + ok8) * 2) : OOB;
        if (first)
          bdma16(rs_a, sa + (wave + NW * i) * 1024, voff, kb * 2);
        else
          bdma16(rs_a2, sa + (wave + NW * i) * 1024, voff, (kb - ksp) * 2);
      }
    }
    const int wb_off = p.rows_per_w > 0 ? (int)((tm0 / p.rows_per_w) * p.w_batch_stride * 2) : 0;   // this tile's weights
#pragma unroll
    for (int i = 0; i < WG; ++i) {
      const bool in_tile = (WGT % NW == 0) || (wave + NW * i < WGT);   // wave-uniform
      bdma16(rs_w, in_tile ? sw + (wave + NW * i) * 1024 : smem + NS * STAGE, in_tile ? w_lane : OOB,
             ((tn0 + RSTEP * i) * (int)p.ldw + kb) * 2 + wb_off);
    }
  };

  // Tile id -> (row tile, column tile).  An XCD runs 32 consecutive ids at a time (xcd_remap): row-major ids make that
  // 32 / tiles_n row tiles x tiles_n column tiles, i.e. with the 16 / 32 column tiles of the feed-forward GEMMs one or two
  // A tiles against 16 / 32 different W tiles -- every CU of the XCD streams its own 820 KB of weights through an L2 that
  // holds 4 MB.  Ids are therefore laid out in column panels of 8 (I2V_GEMM_PANEL): 32 consecutive ids = 4 row tiles x 8
  // column tiles, each A tile shared by 8 CUs and each W tile by 4.
  const int tiles_m_all = ntiles / tiles_n;
  auto tile_coords = [&](int id, int& tm, int& tn) {
    if (I2V_CONV_COLMAJOR && AMODE == I2V_A_CONV3X3) {
      // conv: a W tile (320 x 9 Cin) is ~18x the input footprint of an A tile, and with 2-4 column tiles it does not stay
      // in L2: column-major ids give an XCD's 32 concurrent tiles ONE W tile (and 32 neighbouring pixel tiles).
      // Same box: conv class 15.03 -> 14.89 ms per step.
      tn = id / tiles_m_all;
      tm = id - tn * tiles_m_all;
    } else if (I2V_GEMM_PANEL > 0 && tiles_n > I2V_GEMM_PANEL && tiles_n % I2V_GEMM_PANEL == 0) {
      const int per_panel = tiles_m_all * I2V_GEMM_PANEL;
      const int panel = id / per_panel, r = id - panel * per_panel;
      tm = r / I2V_GEMM_PANEL;
      tn = panel * I2V_GEMM_PANEL + (r - tm * I2V_GEMM_PANEL);
    } else {
      tm = id / tiles_n;
      tn = id - tm * tiles_n;
    }
  };
  // ---- tile loop.  One workgroup per tile by default (a single trip).  In the persistent form (FAST kernels launched
  //      with min(tiles, CUs) workgroups, see persistent_grid) workgroup b takes tile b of every round of gridDim.x
  //      tiles (the XCD remap applied inside a round, so a round's tiles are laid out over the XCDs as a
  //      one-tile-per-workgroup launch would), and the DMA of a tile's FIRST K tile is issued from the previous tile's
  //      loop (behind its last barrier), landing under that tile's last MFMAs and its epilogue.
  const int nkt_all = (K + BK - 1) / BK;
  const int kt0 = SPLIT ? (int)blockIdx.y * kps : 0;
  const int nkt = SPLIT ? min(nkt_all, kt0 + kps) : nkt_all;
  static_assert(!FAST || (!SPLIT && AMODE == I2V_A_PLAIN && NW == 8), "FAST is a plain, un-split, 8-wave GEMM");
  constexpr bool PERSIST = FAST;
  int sbase = 0;             // LDS stage of the current tile's first K tile
  bool have_first = false;   // ... which the previous tile's loop has already issued
  int ws_buf = 0;            // LNF: which of the two weight-row-sum buffers this tile uses
  for (int round0 = 0; round0 < ntiles; round0 += (int)gridDim.x) {
  const int in_round = min((int)gridDim.x, ntiles - round0);
  if ((int)blockIdx.x >= in_round) break;
  const int tile = round0 + xcd_remap(blockIdx.x, in_round);
  int tile_m, tile_n;
  tile_coords(tile, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int next0 = round0 + (int)gridDim.x;
  const bool has_next = PERSIST && next0 < ntiles && (int)blockIdx.x < min((int)gridDim.x, ntiles - next0);
#if defined(I2V_PROBE) && I2V_PROBE == 5
  I2V_STAMP(0);
#endif
  int c_pix[AG], c_oy[AG], c_ox[AG];           // conv: first pixel of the row's image, output pixel coordinates
  if (AMODE == I2V_A_CONV3X3) {
    conv_rows(m0, c_pix, c_oy, c_ox);
  } else {
#pragma unroll
    for (int i = 0; i < AG; ++i) c_pix[i] = c_oy[i] = c_ox[i] = 0;
  }
  auto issue = [&](int kt, int stage) { issue_at(kt, stage, m0, n0, c_pix, c_oy, c_ox); };
  auto issue_next_first = [&](int stage) {   // K tile 0 of this workgroup's next output tile (plain A only)
    const int nt = next0 + xcd_remap(blockIdx.x, min((int)gridDim.x, ntiles - next0));
    int ntm, ntn;
    tile_coords(nt, ntm, ntn);
    issue_at(0, stage, ntm * BM, ntn * BN, c_pix, c_oy, c_ox);
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LNF: this column tile's 320 weight row sums into LDS.  Two buffers, alternating per output tile: a slow wave may still
  // be reading the previous tile's sums in its epilogue.  The wait is the one the first K tile's barrier would do anyway
  // (everything older -- the previous epilogue's stores, the first K tile's DMA -- is waited for there too); done here,
  // the value is not carried into the K loop.
  float* const lds_ws = lds_ws2 + ws_buf * BN;
  if (!have_first) {   // NS - 1 K tiles in flight before the loop
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (kt0 + st < nkt) issue(kt0 + st, (sbase + st) % NS);
  }
  if (LNF) {
    const float* lnws_g = reinterpret_cast<const float*>(p.ln_wsum);
    if (NW == 8) {
      if (tid < BN) {
        const float ln_col = lnws_g[n0 + tid];
        wait_vmcnt<0>();
        lds_ws[tid] = ln_col;
      }
    } else {   // 256 threads for 320 columns
      const float c0 = lnws_g[n0 + tid];
      const float c1 = tid < BN - 256 ? lnws_g[n0 + 256 + tid] : 0.f;
      wait_vmcnt<0>();
      lds_ws[tid] = c0;
      if (tid < BN - 256) lds_ws[256 + tid] = c1;
    }
    ws_buf ^= 1;
  }
  auto sync_tile = [&](int kt) {
    // K tile kt has landed.  Two stages: it is the only one in flight.  Deeper pipelines: the NS - 2 tiles issued after it
    // stay in flight (every wave issues exactly AG + WG DMA instructions per tile; vmcnt counts in issue order); the last
    // NS - 2 trips, with fewer tiles behind, wait for everything.
    if (NS == 2 || kt + NS - 2 >= nkt)
      wait_vmcnt<0>();
    else
      wait_vmcnt<(NS - 2) * (AG + WG)>();
    // one barrier per K tile: every wave's share of tile kt is in LDS, and every wave has finished reading tile
    // kt - 1 (or the previous output tile's epilogue slabs), whose stage is the one refilled next
    __builtin_amdgcn_s_barrier();
  };
  auto prefetch = [&](int kt) {
    const int stage = (sbase + kt - kt0 + NS - 1) % NS;   // the stage tile kt - 1 was read from (closed by this trip's barrier)
    if (kt + NS - 1 < nkt)
      issue(kt + NS - 1, stage);
    else if (NS == 2 && has_next)
      issue_next_first(stage);
  };
  // Fragment reads.  BK = 64: a row is 128 B and the 16-row steps of a wave's blocks leave the swizzle term
  // ((row >> 1) & 7) unchanged, so every fragment address is (one per-lane offset per operand) ^ (k-step << 6)
  // + (block * 2 KiB as the instruction's immediate) + the stage base.  The stage base enters as an opaque scalar: one
  // v_add per operand and k-step, instead of an address register per (stage, k-step, operand) held through the loop --
  // the nine registers the accumulator-bound LayerNorm kernels spilled to scratch INSIDE the K loop (a scratch reload
  // there waits on vmcnt, i.e. on the DMA it should overlap).
  const int fa_lane = big_lds_addr<BK>(wm * WM + l15, g);
  const int fw_lane = BM * BK * 2 + big_lds_addr<BK>(wn * WNC + l15, g);
  auto read_frags = [&](int cur, int ks, f16x8 (&wf)[NI], f16x8 (&af)[MI]) {
    if constexpr (BK == 64) {
      int so = cur * STAGE;
      asm volatile("" : "+s"(so));
      const char* pw = smem + ((fw_lane ^ (ks << 6)) + so);
      const char* pa = smem + ((fa_lane ^ (ks << 6)) + so);
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(pw + i * 2048);
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = *reinterpret_cast<const f16x8*>(pa + j * 2048);
    } else {
      const char* sa = smem + cur * STAGE;
      const char* sw = sa + BM * BK * 2;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        wf[i] = *reinterpret_cast<const f16x8*>(sw + big_lds_addr<BK>(wn * WNC + i * 16 + l15, ks * 4 + g));
#pragma unroll
      for (int j = 0; j < MI; ++j)
        af[j] = *reinterpret_cast<const f16x8*>(sa + big_lds_addr<BK>(wm * WM + j * 16 + l15, ks * 4 + g));
    }
  };
  // D = W_frag * A_frag: lane owns 4 consecutive n of one row m.  For the transposed V^T store the operands are
  // swapped (D = A_frag * W_frag): lane owns 4 consecutive m (keys) of one channel n = an 8-byte run of a V^T row.
  auto mma = [&](const f16x8 (&wf)[NI], const f16x8 (&af)[MI]) {
#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = (STORE == I2V_STORE_VT_T) ? mfma16x16x32(af[j], wf[i], acc[i][j])
                                              : mfma16x16x32(wf[i], af[j], acc[i][j]);
#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // LNF: row sums of x and x^2 from the A fragments.  A wave's MI row blocks are shared by the four N-waves that read
  // the same rows: wave wn accumulates blocks wn * RB .. wn * RB + RB - 1 (a wave-uniform select: a runtime index into
  // the fragment array would send it to scratch).  v_dot2_f32_f16: exact fp16 products, fp32 sums; 8 VALU per fragment.
  // Raw moments in fp32 are accurate to ~(mean / std)^2 x 1e-7, far below the quantisation of the fp16 rows themselves
  // (mean / std x 5e-4), so no shift is needed here (unlike GroupNorm's million-element sums).
  constexpr int RB = MI / 4 > 0 ? MI / 4 : 1;
  float ln_s[RB], ln_q[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) ln_s[b] = ln_q[b] = 0.f;
  auto ln_accum = [&](const f16x8 (&af)[MI]) {
    static_for<MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (wn == (j / RB) % 4) {
        const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f16x2 a2 = {af[j][2 * e], af[j][2 * e + 1]};
          ln_s[j % RB] = __builtin_amdgcn_fdot2(a2, one2, ln_s[j % RB], false);
          ln_q[j % RB] = __builtin_amdgcn_fdot2(a2, a2, ln_q[j % RB], false);
        }
      }
    });
  };

  // The two waves of a SIMD (wave w and w + 4: wm = 0 and wm = 1) run half a K tile out of phase: the wm = 1 group
  // reads the fragments of a tile's last k-step BEFORE the tile barrier but issues their MFMAs AFTER it, i.e. while the
  // wm = 0 group is reading the next tile's first fragments; from there on one wave's LDS reads coincide with the
  // other's MFMAs for the whole tile, instead of both waves reading, then both computing.
  if (!STAGGER || wm == 0) {
    for (int kt = kt0; kt < nkt; ++kt) {
      sync_tile(kt);
      if (kt == kt0) { I2V_STAMP(1); }
      prefetch(kt);
      const int cur = (sbase + kt - kt0) % NS;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        f16x8 wf[NI], af[MI];
        read_frags(cur, ks, wf, af);
        if (LNF) ln_accum(af);
        mma(wf, af);
      }
    }
  } else {
    // (initialised: left undefined on the zero-trip path they become values carried around the whole TILE loop, 52
    // registers live through the other branch and the epilogue)
    f16x8 pwf[NI], paf[MI];
#pragma unroll
    for (int i = 0; i < NI; ++i) pwf[i] = zero8();
#pragma unroll
    for (int j = 0; j < MI; ++j) paf[j] = zero8();
    for (int kt = kt0; kt < nkt; ++kt) {
      sync_tile(kt);
      const int cur = (sbase + kt - kt0) % NS;
      if (kt > kt0) mma(pwf, paf);   // before the DMA address arithmetic: the pending fragments die here
      prefetch(kt);
#pragma unroll
      for (int ks = 0; ks + 1 < KSTEPS; ++ks) {
        f16x8 wf[NI], af[MI];
        read_frags(cur, ks, wf, af);
        if (LNF) ln_accum(af);
        mma(wf, af);
      }
      __builtin_amdgcn_sched_barrier(0);   // strict read / MFMA phases: the overlap comes from the partner wave
      read_frags(cur, KSTEPS - 1, pwf, paf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads have left LDS before the stage can be refilled
      if (LNF) ln_accum(paf);
    }
    if (nkt > kt0) mma(pwf, paf);
  }
  I2V_STAMP(2);
  if (LNF) {
    // a row's 32 k of one step sit on the 4 lane groups: fold them, then lane group 0 publishes (mean, rstd) of the
    // rows of this wave's blocks; the epilogue's barrier (with the LDS write retired) makes them visible to every wave
    const float inv_k = uniform_f(1.0f / (float)K);
    int pl = lane;
    asm volatile("" : "+v"(pl));
    const int a16 = (pl ^ 16) << 2, a32 = (pl ^ 32) << 2;
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float sv = xor_sum(xor_sum(ln_s[b], a16), a32), qv = xor_sum(xor_sum(ln_q[b], a16), a32);
      const float mean = sv * inv_k;
      const float var = fmaxf(qv * inv_k - mean * mean, 0.f);
      if (g == 0) lds_st[wm * WM + ((MI >= 4 ? wn * RB : 0) + b) * 16 + l15] = float2{mean, rsqrtf(var + p.ln_eps)};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // the next output tile's first K tile is in flight into stage `sbase`; the stage the loop read last is free for the
  // epilogue's transpose slabs once every wave has left the loop (the epilogues' own barrier)
  sbase = NS == 2 ? (sbase + (nkt - kt0)) & 1 : 0;
  have_first = has_next;
  // (4-wave workgroups are never persistent: nothing is in flight after the loop and the slabs may span both stages --
  //  the V^T epilogue's four 8.4 KiB slabs exceed one 28 KiB stage)
  // (deeper pipelines are never persistent either: every stage is free after the loop)
  char* const slab_stage = (NW == 4 || NS > 2) ? smem : smem + (sbase ^ 1) * STAGE;

  // ---------------------------------------------------------------- epilogue (lane: row m, 4 consecutive n)
  // The (epilogue, store mode) pair is a template parameter and the accumulator indices are compile-time constants
  // (static_for): with the generic runtime-switched store the 8 x 5 loop was not unrolled, the 160 accumulators went
  // through scratch memory and every tile paid ~50 us for it.
  // (lane and what derives from it enter as arguments made opaque per tile: otherwise the compiler hoists the epilogue's
  // lane arithmetic -- slab and row addresses, a few dozen registers -- in front of the tile loop, where it stays live
  // across the accumulator-bound K loop: 70-290 spilled registers in the 256-row kernels)
  auto epilogue = [&](const int lane, const int g, const int l15, const int tid) {
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
  const f16* __restrict__ resid = reinterpret_cast<const f16*>(p.residual);
  const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(p.rowvec);
  f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
  const float oscale = p.out_scale;
  const float inv_rpv = uniform_f(1.0f / (float)(p.rows_per_vec > 0 ? p.rows_per_vec : 1));
  if (SPLIT) {
    float* __restrict__ ws = reinterpret_cast<float*>(p.workspace) + (int64_t)blockIdx.y * M * N;
    static_for<MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int m = m0 + wm * WM + j * 16 + l15;
      if (m < M) {
        static_for<NI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          *reinterpret_cast<f32x4*>(ws + (int64_t)m * N + n0 + wn * WNC + i * 16 + g * 4) = acc[i][j];
        });
      }
    });
    return;
  }
  if (STORE == I2V_STORE_VT_T && p.vt_len % 8 == 0 && M % 8 == 0 && p.vt_ld % 8 == 0 &&
      (reinterpret_cast<uintptr_t>(p.c) & 15) == 0) {
    // accumulator rows = m (4 g + r), column = n (l15): element (m, n) -> C[((m / L) * N + n) * ld + m % L].
    // Through LDS like the row-major stores: per 16-channel block the wave's [16 channels][WM keys] fp32 slab is
    // re-read so that a lane owns 8 consecutive keys of one channel: 16-byte stores, WM * 2 contiguous bytes per V^T row
    // (the direct form stores 16 rows x 32 bytes per instruction).
    constexpr int LDS_LD = WM + 4;   // floats: 16 rows start on 16 distinct bank groups
    constexpr int TPR = WM / 8, NT = 16 * TPR;
    __builtin_amdgcn_s_barrier();    // every wave has left the K loop: the stages are free
    // per wave: the [16 channels][WM keys] slab + its WNC bias values (fetched once, read back per task from LDS)
    float* stg = reinterpret_cast<float*>(slab_stage) + wave * (16 * LDS_LD + WNC);
    float* bslab = stg + 16 * LDS_LD;
    if (LNF) {
      // LayerNorm fold: accumulator rows are tokens m = .. + 4 g + r, the column is channel n (l15)
      static_for<MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = m0 + wm * WM + j * 16 + 4 * g;
        float mu[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 st = lds_st[m - m0 + r];
          mu[r] = st.x;
          rs[r] = st.y;
        }
        static_for<NI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const float ws = lds_ws[wn * WNC + i * 16 + l15];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = rs[r] * (acc[i][j][r] - mu[r] * ws);
        });
      });
    }
    // Global accesses as unconditional raw buffer instructions (see the row-major epilogue below): per 8-key task the old
    // form loaded ONE bias value and waited s_waitcnt vmcnt(0) for it -- 20 dependent memory round trips per wave and
    // tile, each also waiting for the previous task's store.  Now: the wave's bias values are fetched once up front, the
    // positional-table rows of block i + 1 are in flight under block i, lanes past M carry an out-of-range offset.
    constexpr unsigned VOOB = 0x80000000u;
    constexpr int QV = NT / 64;
    const bool has_pe = rowvec != nullptr;
    const int64_t vt_rows = (int64_t)(M / p.vt_len) * N;
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(((vt_rows - 1) * p.vt_ld + p.vt_len) * 2), 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(bias ? bias : C), 0, bias ? N * 2 : 0, 0x00020000);
    const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(has_pe ? rowvec : C), 0,
                                                        has_pe ? (int)((((int64_t)N - 1) * p.ld_rowvec + p.rowvec_period) * 2) : 0,
                                                        0x00020000);
#pragma unroll
    for (int e0 = 0; e0 < WNC; e0 += 64) {
      const int e = e0 + lane;
      const unsigned short raw = __builtin_amdgcn_raw_buffer_load_b16(rs_b, e < WNC ? (unsigned)((n0 + wn * WNC + e) * 2) : VOOB, 0, 0);
      if (e < WNC) bslab[e] = (float)__builtin_bit_cast(f16, raw);   // (an LDS write under the mask: no memory operation hides in the branch)
    }
    f16x8 pev[NI][QV];
    auto fetch_pe = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int q = 0; q < QV; ++q) {
        const int t = lane + 64 * q;
        const int row = t / TPR, c = t - row * TPR;
        const int n = n0 + wn * WNC + i * 16 + row;
        const int m = m0 + wm * WM + c * 8;
        const unsigned off = m < M ? (unsigned)((n * (int)p.ld_rowvec + (m & (p.rowvec_period - 1))) * 2) : VOOB;
        pev[i][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, 0, 0));
      }
    };
    static_for<NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      fetch_pe(ic);     // in flight under this block's LDS transpose (a small table that sits in L2)
      static_for<MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        *reinterpret_cast<f32x4*>(stg + l15 * LDS_LD + j * 16 + 4 * g) = acc[i][j];
      });
#pragma unroll
      for (int q = 0; q < QV; ++q) {
        const int t = lane + 64 * q;
        const int row = t / TPR, c = t - row * TPR;
        const int n = n0 + wn * WNC + i * 16 + row;
        const int m = m0 + wm * WM + c * 8;     // M % 8 == 0 and vt_len % 8 == 0: the 8 keys are in range and in one batch
        const float bn = bslab[i * 16 + row];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8 + 4);
        const int bt = m / p.vt_len, kk = m - bt * p.vt_len;
        float pe[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (has_pe) {   // transposed positional table [N][period]: the 8 keys are 8 consecutive positions
#pragma unroll
          for (int e = 0; e < 8; ++e) pe[e] = (float)pev[i][q][e];
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (f16)((lo[e] + bn + pe[e]) * oscale);
          o[4 + e] = (f16)((hi[e] + bn + pe[4 + e]) * oscale);
        }
        const unsigned off = m < M ? (unsigned)(((bt * N + n) * (int)p.vt_ld + kk) * 2) : VOOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, off, 0, 0);
      }
    });
    return;
  }
  if (STORE == I2V_STORE_VT_T) {
    static_for<NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int n = n0 + wn * WNC + i * 16 + l15;
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
  const int ncol0 = n0 + wn * WNC + g * 4;
  f16x4 b4[NI];
  static_for<NI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    b4[i] = bias ? *reinterpret_cast<const f16x4*>(bias + ncol0 + i * 16) : f16x4{0, 0, 0, 0};
  });
  if ((STORE == I2V_STORE_ROWMAJOR || STORE == I2V_STORE_ROWPERM) && (EPI != I2V_EPI_GEGLU || (!rowvec && !resid))) {
    // Row-contiguous stores through LDS.  In the accumulator layout a store (or residual load) instruction touches 16
    // rows x 32 bytes; with 40 of each per wave the tile's epilogue was bound by the address/line rate of the memory
    // pipe (a K = 320 tile spent ~2/3 of its time here: 3.5 TB/s for a GEMM that an elementwise add of the same
    // bytes does at 6.2).  Each wave therefore transposes its 16 x 80 fp32 block through a private LDS slab (no
    // barrier beyond the one that retires the K loop's stages; LDS executes one wave's operations in order) and
    // re-reads it so that a lane owns 8 consecutive columns of one row: 16-byte residual / time-embedding loads and
    // stores, 160 contiguous bytes per row, half the instructions.  Rounding is unchanged (fp32 until the final cast).
    constexpr int OC = EPI == I2V_EPI_GEGLU ? WNC / 2 : WNC;   // output columns per wave
    constexpr int LDS_LD = OC + 4;                        // floats; 84 / 44: 16 rows start on 16 distinct bank groups
    constexpr int TPR = OC / 8, NT = 16 * TPR;            // 8-column tasks per row / per 16-row block
    const int out_col0 = EPI == I2V_EPI_GEGLU ? (n0 >> 1) + wn * (WNC / 2) : n0 + wn * WNC;
    // One 8-column task of the 16-row block j: (valid, row m, destination row, first column)
    constexpr int QN = (NT + 63) / 64;
    auto task = [&](int j, int q, int& m, int& m_out, int& n) {
      const int t = lane + 64 * q;
      const int row = t / TPR, c = t - row * TPR;
      m = m0 + wm * WM + j * 16 + row;
      m_out = m;
      if (STORE == I2V_STORE_ROWPERM) {
        const int per = p.hw * p.frames;
        const int b = m / per, rem = m - b * per;
        const int pix = rem / p.frames, f = rem - pix * p.frames;
        m_out = (b * p.frames + f) * p.hw + pix;
      }
      n = out_col0 + c * 8;
      return t < NT && m < M;
    };
    // Every global access of this epilogue is an UNCONDITIONAL raw buffer instruction: a lane with no task (row >= M, the
    // partial last pass over a block's tasks) carries an out-of-range offset, so the descriptor's range check returns zeros
    // for its load and drops its store, and an absent operand (no residual, no row vector) is a descriptor of size 0.
    // Why: with the loads and stores under exec-mask branches (`if (task) ...`) the compiler cannot count how many memory
    // operations were issued after a given load, and waits s_waitcnt vmcnt(0) before every use -- on CDNA4 vmcnt counts
    // stores too, so each 8-column task waited for the previous task's STORE to be acknowledged and for the residual
    // prefetch issued a moment earlier: "stage + store" was 15.7 us of a 131072 x 320 x 320 + residual tile's 35 and 4.7 us
    // without a residual (tools/tile_timeline.py, round 3).  Straight-line buffer accesses get exact counted waits: the
    // residual / row-vector rows of block j + RES_AHEAD stay in flight under block j's stores.
    // (offsets are 32-bit: M * ldc, M * ldr < 2^30 elements, checked by big_plan)
    constexpr unsigned EOOB = 0x80000000u;
    // ONE added operand per problem: the residual (out-projections, conv2, proj_out) or the row-vector table (time
    // embedding, positional table, per-image bias of a folded GroupNorm) -- big_plan sends a problem with both to the
    // generic kernel -- so one set of prefetch registers serves either (two sets spilled 150-290 registers per lane)
    const bool has_res = !LNF && EPI != I2V_EPI_GEGLU && resid != nullptr;
    const bool has_rv = EPI != I2V_EPI_GEGLU && !has_res && rowvec != nullptr;
    const bool has_add = has_res || has_rv;
    const int n_out_cols = EPI == I2V_EPI_GEGLU ? N / 2 : N;
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((((int64_t)M - 1) * p.ldc + n_out_cols) * 2), 0x00020000);
    const int rv_rows = p.rowvec_period > 0 ? p.rowvec_period : (M - 1) / (p.rows_per_vec > 0 ? p.rows_per_vec : 1) + 1;
    const int add_ld = has_res ? (int)p.ldr : (int)p.ld_rowvec;
    const int64_t add_rows = has_res ? M : rv_rows;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(has_res ? resid : has_rv ? rowvec : C), 0,
                                                        has_add ? (int)(((add_rows - 1) * add_ld + N) * 2) : 0, 0x00020000);
#ifndef I2V_RES_AHEAD
#define I2V_RES_AHEAD 3
#endif
    // (behind a folded LayerNorm the added operand can only be a small row-vector table that sits in L2: fetched at the top
    //  of its own block, under the block's LDS transpose, so the accumulator-bound LayerNorm kernels keep their registers)
    // HILO (i2v_gemm_params.residual_lo / c_lo, the precise residual stream): the low halves of the residual rows travel with
    // the high ones (same offsets, a descriptor of size 0 when there is no residual_lo), one block ahead instead of three so that
    // the two sets of prefetch registers together are what the one set was
    constexpr int RES_AHEAD = LNF ? 0 : (HILO ? 1 : I2V_RES_AHEAD);
    static_assert(!HILO || (!LNF && !GNS && EPI == I2V_EPI_NONE), "the precise stream is a plain row-major / row-permuted store");
    const auto rs_xl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HILO && has_res && p.residual_lo ? p.residual_lo : p.c), 0,
                                                         (HILO && has_res && p.residual_lo) ? (int)(((add_rows - 1) * add_ld + N) * 2) : 0, 0x00020000);
    const auto rs_cl = __builtin_amdgcn_make_buffer_rsrc(HILO && p.c_lo ? p.c_lo : p.c, 0,
                                                         (HILO && p.c_lo) ? (int)((((int64_t)M - 1) * p.ldc + n_out_cols) * 2) : 0, 0x00020000);
    f16x8 xpre[MI][QN];
    f16x8 xlo[HILO ? MI : 1][QN];
    auto fetch_rows = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        int m, m_out, n;
        const bool ok = task(j, q, m, m_out, n);
        int xrow = m_out;
        if (!has_res) xrow = p.rowvec_period > 0 ? (m & (p.rowvec_period - 1)) : fast_div(m, p.rows_per_vec > 0 ? p.rows_per_vec : 1, inv_rpv);
        const unsigned off = ok ? (unsigned)((xrow * add_ld + n) * 2) : EOOB;
        xpre[j][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        if constexpr (HILO) xlo[j][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_xl, off, 0, 0));
      }
    };
    // (no residual behind a folded LayerNorm: those GEMMs feed q / k / v / the feed-forward; i2v_gemm_big_ln_ok refuses it)
    if (EPI != I2V_EPI_GEGLU) {
      static_for<(RES_AHEAD < MI ? RES_AHEAD : MI)>([&](auto jc) { fetch_rows(jc); });
    }
    __builtin_amdgcn_s_barrier();                         // every wave has left the K loop: the stages are free
    float* stg = reinterpret_cast<float*>(slab_stage) + wave * (16 * LDS_LD);
    if constexpr (GNS) {
      // GroupNorm statistics of the tile's output for the norm that reads it next (ResnetBlock2D: conv1 -> norm2, unet:203-214):
      // per statistics group of the tile's BN channels the partial (mean, M2) over the tile's BM rows, in the layout and with the
      // meaning of gn_stats_kernel's per-group partials (norm.hip), so that the consumer's statistics pass -- a launch that re-reads
      // the whole tensor -- disappears (i2v_gemm_params.gn_partial).  From the RAW accumulators a (the convolution without bias and
      // time-embedding row k_c, which are constants of a column inside a tile: mean_c = mean(a_c) + k_c, M2_c = sum a_c^2 -
      // (sum a_c)^2 / n is unchanged by them, and a carries no large common offset to cancel against), in fp32, before the fp16
      // rounding of the stored result.  Host-checked: M % BM == 0, a tile lies in one image and one row of the row vector.
      float* cst = reinterpret_cast<float*>(slab_stage) + NW * (16 * LDS_LD);      // [NW / 4][BN] (sum, sum of squares), then k_c [BN]
      float* kst = cst + (NW / 4) * BN * 2;
      auto row16_sum = [](float x) {          // over the 16 lanes of a DPP row (the rows l15 of one lane group)
        x = sum_lanes8(x);
        return x + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
      };
      static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
        static_for<MI>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          s4 += acc[i][j];
          q4 += acc[i][j] * acc[i][j];
        });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s4[r] = row16_sum(s4[r]);
          q4[r] = row16_sum(q4[r]);
        }
        if (l15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float2*>(cst + (wm * BN + wn * WNC + i * 16 + 4 * g + r) * 2) = float2{s4[r], q4[r]};
        }
      });
      if (tid < BN) {
        const int vrow = rowvec ? (p.rowvec_period > 0 ? (m0 & (p.rowvec_period - 1)) : fast_div(m0, p.rows_per_vec > 0 ? p.rows_per_vec : 1, inv_rpv)) : 0;
        kst[tid] = (bias ? (float)bias[n0 + tid] : 0.f) + (rowvec ? (float)rowvec[(int64_t)vrow * p.ld_rowvec + n0 + tid] : 0.f);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int cpg = N / p.gn_groups, gpt = BN / cpg;          // channels per group, groups of this tile's BN channels
      if (tid < gpt) {
        constexpr int NWM = NW / 4;
        const float nr = (float)WM, inv_nr = 1.0f / (float)WM;
        float mg = 0.f;
        for (int c = 0; c < cpg; ++c)
#pragma unroll
          for (int w = 0; w < NWM; ++w) mg += cst[(w * BN + tid * cpg + c) * 2] * inv_nr + kst[tid * cpg + c];
        mg /= (float)(cpg * NWM);
        float m2 = 0.f;
        for (int c = 0; c < cpg; ++c)
#pragma unroll
          for (int w = 0; w < NWM; ++w) {
            const float sc = cst[(w * BN + tid * cpg + c) * 2], qc = cst[(w * BN + tid * cpg + c) * 2 + 1];
            const float dm = sc * inv_nr + kst[tid * cpg + c] - mg;
            m2 += fmaxf(qc - sc * sc * inv_nr, 0.f) + nr * dm * dm;
          }
        const int hw = p.out_h * p.out_w, img = m0 / hw, chunk = (m0 - img * hw) / BM, nchunk = hw / BM;
        float* dst = reinterpret_cast<float*>(p.gn_partial) + (((int64_t)img * nchunk + chunk) * p.gn_groups + n0 / cpg + tid) * 2;
        dst[0] = mg;
        dst[1] = m2;
      }
      // (the slabs below and the statistics region are disjoint; the accumulators are untouched)
    }
    // bias (and the GEGLU gate, in place: registers 0 / 1 of each accumulator become the two outputs) first, as a
    // pure register pass: fused with the staging below, the GELU temporaries pushed accumulators into scratch
    if (LNF) {
      // LayerNorm fold: v = rstd_m (acc - mean_m wsum_n); the bias added below is W beta + b
      f32x4 ws4[NI];
      static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ws4[i] = *reinterpret_cast<const f32x4*>(lds_ws + wn * WNC + g * 4 + i * 16);
      });
      static_for<MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = m0 + wm * WM + j * 16 + l15;
        const float2 st = lds_st[m - m0];
        static_for<NI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = st.y * (acc[i][j][r] - st.x * ws4[i][r]);
        });
      });
    }
    static_for<MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b4[i][r];
        if (EPI == I2V_EPI_GEGLU) {
          acc[i][j][0] = acc[i][j][0] * gelu_erf(acc[i][j][1]) * oscale;
          acc[i][j][1] = acc[i][j][2] * gelu_erf(acc[i][j][3]) * oscale;
        }
      });
    });
    I2V_STAMP(3);
    static_for<MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + RES_AHEAD < MI && EPI != I2V_EPI_GEGLU) {
        fetch_rows(std::integral_constant<int, j + RES_AHEAD>{});
      }
      static_for<NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (EPI == I2V_EPI_GEGLU)
          *reinterpret_cast<float2*>(stg + l15 * LDS_LD + i * 8 + 2 * g) = float2{acc[i][j][0], acc[i][j][1]};
        else
          *reinterpret_cast<f32x4*>(stg + l15 * LDS_LD + i * 16 + 4 * g) = acc[i][j];
      });
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        int m, m_out, n;
        const bool ok = task(j, q, m, m_out, n);
        // (a lane without a task re-reads some row of the slab -- row index wrapped into its 16 rows -- and its store is
        //  dropped by the range check)
        const int t = lane + 64 * q;
        const int row = (t / TPR) & 15, c = t - (t / TPR) * TPR;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (EPI != I2V_EPI_GEGLU) {
          if (has_add) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)xpre[j][q][e];
            if constexpr (HILO) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)xlo[j][q][e];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= oscale;
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
        const unsigned off = ok ? (unsigned)((m_out * (int)p.ldc + n) * 2) : EOOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, off, 0, 0);
        if constexpr (HILO) {       // what the rounding to o dropped (a descriptor of size 0 without c_lo: the store is dropped)
          f16x8 ol;
#pragma unroll
          for (int e = 0; e < 8; ++e) ol[e] = (f16)(v[e] - (float)o[e]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ol), rs_cl, off, 0, 0);
        }
      }
    });
#if defined(I2V_PROBE) && I2V_PROBE == 5
    I2V_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    I2V_STAMP(5);
    if (tid == 0) {
      long long* dst = reinterpret_cast<long long*>(C + (int64_t)m0 * p.ldc + out_col0);
      for (int q = 0; q < 6; ++q) dst[q] = stamp[q];
      for (int q = 0; q < 6; ++q) dst[6 + q] = cstamp[q];
    }
#endif
    return;
  }
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
      const f16* rv = rowvec ? rowvec + (int64_t)(p.rowvec_period > 0 ? (m & (p.rowvec_period - 1)) : fast_div(m, p.rows_per_vec, inv_rpv)) * p.ld_rowvec
                             : nullptr;
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
  };
  int e_lane = lane;
  asm volatile("" : "+v"(e_lane));
  epilogue(e_lane, e_lane >> 4, e_lane & 15, wave * 64 + e_lane);
  if (!PERSIST) break;
  }   // persistent tile loop
}

// Persistent launch (I2V_GEMM_PERSIST=1; OFF by default): one workgroup per CU walks the tiles, the next tile's first K
// tile already in flight under the current epilogue.  Measured on the step's shapes (round 2, same box): the in-kernel
// timeline loses its 2.4-2.8 us first-tile wait, but a cache-warm 131072 x 2560 x 320 GEGLU gains only 2 %, the K = 640
// GEGLU loses 2.5 %, N = K = 320 gains 7 %, the whole step: 59.3 vs 59.3 ms.  The hardware's own workgroup turnover
// already hides what the loop hides; what bounds these kernels is the K loop itself (2.3-2.8 us per 64-deep K tile
// against 1.3 at the clock the chip holds under this load: one 72 KiB stage in flight does not cover the DMA's tail
// latency, and LDS has no room for a third stage).  Kept as an A/B switch.
int persist_mode() {   // 0 (default) never, 1 every eligible plain GEMM, 2 only the LayerNorm-folded row-major GEMMs
  static const int mode = getenv("I2V_GEMM_PERSIST") ? atoi(getenv("I2V_GEMM_PERSIST")) : 0;
  return mode;
}
int persistent_grid(int ntiles) {
  static const int cus = [] {
    if (persist_mode() == 0) return 0;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return 0;
    return n;
  }();
  return (cus > 0 && ntiles > cus) ? cus : ntiles;
}
// Round 3, per flavour (tools/gemm4_ab.py under I2V_GEMM_PERSIST=0 / 1 / 2): on one box the walk gained 3 - 14 % on the
// LayerNorm-folded GEMMs with plain row-major stores (131072 x 960 x 320: 156 -> 134 us) and lost on residual / GEGLU / V^T
// epilogues; on two other boxes mode 2 (those GEMMs only) measured 155.7 vs 159.2 us on that shape and 54.05 vs 54.09 ms
// on the whole step -- no reproducible gain, so the default stays 0.
bool want_persistent(const i2v_gemm_params& p) {
  const int mode = persist_mode();
  if (mode == 1) return true;
  return mode == 2 && p.ln_wsum != nullptr && p.epilogue == I2V_EPI_NONE && p.store_mode == I2V_STORE_ROWMAJOR;
}

template <int BM, int BK, int NS, int AMODE, bool FAST>
int launch_big_mode(const i2v_gemm_params& p, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = p.N / BIG_BN;
  const int ntiles = tiles_m * tiles_n;
  const dim3 grid(FAST ? persistent_grid(ntiles) : ntiles), block(512);
#define I2V_BIG_LAUNCH(EPI, STORE, LNF) \
  hipLaunchKernelGGL((gemm_big_kernel<BM, BK, NS, AMODE, EPI, STORE, false, true, LNF, FAST>), grid, block, 0, s, p, tiles_n, 0, ntiles)
#define I2V_BIG_LAUNCH_HILO(STORE) \
  hipLaunchKernelGGL((gemm_big_kernel<BM, BK, NS, AMODE, I2V_EPI_NONE, STORE, false, true, false, false, 320, 8, false, true>), grid, block, 0, s, \
                     p, tiles_n, 0, ntiles)
  const bool hilo = p.residual_lo != nullptr || p.c_lo != nullptr;      // (vetted by i2v_gemm_f16: EPI_NONE, row-major / row-permuted)
  if (p.ln_wsum != nullptr) {   // LayerNorm-folded epilogues (i2v_gemm_big_ln_ok has vetted the combination)
    if constexpr (AMODE == I2V_A_PLAIN) {
      if (p.epilogue == I2V_EPI_GEGLU)
        I2V_BIG_LAUNCH(I2V_EPI_GEGLU, I2V_STORE_ROWMAJOR, true);
      else if (p.store_mode == I2V_STORE_VT_T)
        I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT_T, true);
      else if (p.store_mode == I2V_STORE_ROWPERM)
        I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWPERM, true);
      else
        I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true);
    }
    const int rc = i2v_check_launch("i2v_gemm_f16(big, LayerNorm fold)");
    return rc < 0 ? rc : 1;
  }
  if (p.epilogue == I2V_EPI_GEGLU) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG_LAUNCH(I2V_EPI_GEGLU, I2V_STORE_ROWMAJOR, false);
  } else if (hilo) {               // the precise residual stream (residual_lo / c_lo): plain row-major / row-permuted stores
    if constexpr (!FAST) {
      if (p.store_mode == I2V_STORE_ROWPERM) {
        if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG_LAUNCH_HILO(I2V_STORE_ROWPERM);
      } else {
        I2V_BIG_LAUNCH_HILO(I2V_STORE_ROWMAJOR);
      }
    }
  } else if (p.store_mode == I2V_STORE_ROWPERM) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWPERM, false);
  } else if (p.store_mode == I2V_STORE_VT) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT, false);
  } else if (p.store_mode == I2V_STORE_VT_T) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT_T, false);
  } else if (AMODE == I2V_A_CONV3X3 && p.gn_partial != nullptr) {      // + GroupNorm partials of the result (vetted by i2v_gemm_gn_partial_rows)
    if constexpr (AMODE == I2V_A_CONV3X3 && !FAST)
      hipLaunchKernelGGL((gemm_big_kernel<BM, BK, NS, AMODE, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, false, true, false, false, 320, 8, true>), grid,
                         block, 0, s, p, tiles_n, 0, ntiles);
  } else {
    I2V_BIG_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWMAJOR, false);
  }
#undef I2V_BIG_LAUNCH
#undef I2V_BIG_LAUNCH_HILO
  const int rc = i2v_check_launch("i2v_gemm_f16(big)");
  return rc < 0 ? rc : 1;
}

// Two workgroups of 4 waves per CU: 128 x 320 x 32 tiles (see the NW template parameter).  Measured 5 - 40 % slower on every
// flavour (below), so the instantiations are compiled only into the A/B library (tools/build_variant.sh --variants).
#ifdef I2V_VARIANTS
template <int AMODE>
int launch_big4(const i2v_gemm_params& p, hipStream_t s) {
  constexpr int BM = 128;
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = p.N / BIG_BN;
  const int ntiles = tiles_m * tiles_n;
  const dim3 grid(ntiles), block(256);
#define I2V_BIG4_LAUNCH(EPI, STORE, LNF) \
  hipLaunchKernelGGL((gemm_big_kernel<BM, 32, 2, AMODE, EPI, STORE, false, false, LNF, false, 320, 4>), grid, block, 0, s, p, tiles_n, 0, ntiles)
  if (p.ln_wsum != nullptr) {
    if constexpr (AMODE == I2V_A_PLAIN) {
      if (p.epilogue == I2V_EPI_GEGLU)
        I2V_BIG4_LAUNCH(I2V_EPI_GEGLU, I2V_STORE_ROWMAJOR, true);
      else if (p.store_mode == I2V_STORE_VT_T)
        I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT_T, true);
      else if (p.store_mode == I2V_STORE_ROWPERM)
        I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWPERM, true);
      else
        I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true);
    }
  } else if (p.epilogue == I2V_EPI_GEGLU) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG4_LAUNCH(I2V_EPI_GEGLU, I2V_STORE_ROWMAJOR, false);
  } else if (p.store_mode == I2V_STORE_ROWPERM) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWPERM, false);
  } else if (p.store_mode == I2V_STORE_VT) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT, false);
  } else if (p.store_mode == I2V_STORE_VT_T) {
    if constexpr (AMODE == I2V_A_PLAIN) I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_VT_T, false);
  } else {
    I2V_BIG4_LAUNCH(I2V_EPI_NONE, I2V_STORE_ROWMAJOR, false);
  }
#undef I2V_BIG4_LAUNCH
  const int rc = i2v_check_launch("i2v_gemm_f16(big, 4-wave)");
  return rc < 0 ? rc : 1;
}

// Which un-split problems take the 4-wave / two-workgroups-per-CU form: NONE by default.  Measured in round 3 on every
// GEMM flavour of the step (tools/gemm4_ab.py, same box, profiles/r3_gemm_4wave_ab.txt): 5 - 40 % SLOWER on all of them
// (131072 x 2560 x 320 GEGLU + LayerNorm 368 -> 386 us, 8192 x 1280 x 5120 + residual 104 -> 145 us, whole step 58.2 ->
// 64.5 ms).  The overlap of one tile's epilogue with the other's MFMAs does not pay for what the 128 x 320 x 32 tile costs:
// 91 FLOP per staged byte instead of 146 against a CU's ~70 GB/s LDS-DMA intake, and a barrier every 40 MFMAs per wave.
// I2V_GEMM_4W=1 selects it for every eligible plain GEMM (kept as a tested A/B switch).
bool use_4wave(const i2v_gemm_params& p, int plan_rows) {
  static const int mode = getenv("I2V_GEMM_4W") ? atoi(getenv("I2V_GEMM_4W")) : 0;
  (void)plan_rows;
  return mode == 1 && p.a_mode == I2V_A_PLAIN && p.N % BIG_BN == 0 && p.K % 32 == 0;
}
#endif

// 3x3 convolutions whose channel count is not a multiple of 320 (the VAE: 128 / 256 / 512): column tiles of 256 or 128
template <int BM, int BN>
int launch_big_conv_bn(const i2v_gemm_params& p, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, BM), tiles_n = p.N / BN;
  const int ntiles = tiles_m * tiles_n;
  hipLaunchKernelGGL((gemm_big_kernel<BM, 64, 2, I2V_A_CONV3X3, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, false, true, false, false, BN>),
                     dim3(ntiles), dim3(512), 0, s, p, tiles_n, 0, ntiles);
  const int rc = i2v_check_launch("i2v_gemm_f16(big conv)");
  return rc < 0 ? rc : 1;
}

// Pipeline shape: two 64-deep stages.  Four 32-deep stages (three tiles in flight, <BM, 32, 4, ...>) measured 5-12 %
// slower on every shape of the step (profiles/r1_tile_sweep.txt), and again in round 3 on the full-chip 16 x 16 level
// (tools/l2_ab.py: 8192 x 1280 x 1280 40.6 -> 45.5 us, x 2560 70.7 -> 78.5): with every CU busy the extra barriers cost more
// than the third tile in flight gives.  Deeper pipelines pay only where the chip is NOT full (deep_plan below).
template <int BM>
int launch_big(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  (void)vec4;
  if (p.a_mode == I2V_A_CONV3X3) return launch_big_mode<BM, 64, 2, I2V_A_CONV3X3, false>(p, s);
  if (p.a2 == nullptr && p.M % BM == 0 && want_persistent(p) && persistent_grid(1 << 20) != (1 << 20) && !p.residual_lo && !p.c_lo)
    return launch_big_mode<BM, 64, 2, I2V_A_PLAIN, true>(p, s);
  return launch_big_mode<BM, 64, 2, I2V_A_PLAIN, false>(p, s);
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
  // (the weight gradients of the training step contract over every token: 320 x 320 x 65536 = 3 tiles with 1024 K tiles each.
  //  Eight splits left them on 24 workgroups, 138 us; plain GEMMs with >= 256 K tiles -- none on the inference path -- may take 64:
  //  54 us)
  const int cap = (p.a_mode == I2V_A_PLAIN && nkt >= 256) ? 64 : 8;
  if (splits > cap) splits = cap;
  if (splits > nkt / 8) splits = nkt / 8;
  if (splits < 2) return 0;
  const int kps = (int)i2v_cdiv(nkt, splits);
  splits = (int)i2v_cdiv(nkt, kps);          // no empty split
  if (kps_out) *kps_out = kps;
  return splits;
}

// 256-row tiles with K split in two to four: problems whose 256-row tiles fill only a quarter to half of the chip (the 16 x 16
// level: 8192 rows x 1280 columns = 128 tiles) with a long K.  As 128-row tiles they get one tile per CU, but a 128-row K tile
// takes 1.5 us for half the FLOPs of a 256-row one at 2.1 us (the K loop runs at the DMA round trip of its one stage in
// flight); splitting K over two workgroups of the full-height tile trades the fp32 partials' round trip
// (84 MB each way at 8192 x 1280) for a third fewer K-loop cycles.  Measured (round 3, tools/split256_ab.py, same box):
// conv 8192 x 1280 x 23040 425.7 -> 382.8 us, x 17280 329.4 -> 298.5, x 11520 + residual 232.7 -> 227.4, x 5760 130.2 -> 121.5;
// GEMM 8192 x 1280 x 5120 + residual 122.9 -> 129.9 (the partials cost more than its 80 K tiles give back).  Whole step, same
// box: 54.84 -> 54.13 ms with the 16 x 16 level's convolutions, 54.02 -> 53.76 with the 8 x 8 level's too (tiles >= 16),
// 52.12 -> 51.99 with the plain GEMMs as well.  Default (I2V_GEMM_SPLIT256 unset / 1): convolutions with >= 90 K tiles and the
// 8 x 8 level's plain GEMMs with >= 64; 2: every eligible GEMM; 0: off.
int splitk256_plan(const i2v_gemm_params& p, int vec4, int* kps_out) {
  static const int on = getenv("I2V_GEMM_SPLIT256") ? atoi(getenv("I2V_GEMM_SPLIT256")) : 1;
  if (!on) return 0;
  const int64_t t256_ = (int64_t)(p.M / 256 > 0 ? p.M / 256 : 1) * (p.N / BIG_BN);
  if (on == 1 && p.a_mode == I2V_A_CONV3X3 && (p.K + 63) / 64 < 90) return 0;
  if (on == 1 && p.a_mode != I2V_A_CONV3X3 && t256_ > 32) return 0;   // plain GEMMs: the 8 x 8 level only (52.12 -> 51.99 ms)
  if (p.N % BIG_BN != 0 || !vec4 || p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR || p.M % 256 != 0) return 0;
  if (p.a2 != nullptr || p.rows_per_w > 0 || p.a_perm_frames > 0 || p.ln_wsum != nullptr) return 0;
  const int nkt = (p.K + 63) / 64;
  const int64_t t256 = (int64_t)(p.M / 256) * (p.N / BIG_BN);
  static const int min_t = getenv("I2V_GEMM_SPLIT256_MINT") ? atoi(getenv("I2V_GEMM_SPLIT256_MINT")) : 16;
  if (nkt < 64 || t256 < min_t || t256 > 128) return 0;
  int splits = (int)(256 / t256);
  if (splits > 8) splits = 8;
  if (splits > nkt / 8) splits = nkt / 8;
  const int kps = (int)i2v_cdiv(nkt, splits);
  splits = (int)i2v_cdiv(nkt, kps);
  if (splits < 2) return 0;
  if (kps_out) *kps_out = kps;
  return splits;
}

int launch_split256(const i2v_gemm_params& p, int vec4, int splits, int kps, hipStream_t s) {
  const int tiles_m = p.M / 256, tiles_n = p.N / BIG_BN;
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  if (p.a_mode == I2V_A_CONV3X3)
    hipLaunchKernelGGL((gemm_big_kernel<256, 64, 2, I2V_A_CONV3X3, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps, tiles_m * tiles_n);
  else
    hipLaunchKernelGGL((gemm_big_kernel<256, 64, 2, I2V_A_PLAIN, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps, tiles_m * tiles_n);
  const int64_t groups = (int64_t)p.M * (p.N / 4);
  const int blocks = (int)(i2v_cdiv(groups, 256) < 2048 ? i2v_cdiv(groups, 256) : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, splits, vec4);
  const int rc = i2v_check_launch("i2v_gemm_f16(split-K, 256-row tiles)");
  return rc < 0 ? rc : 1;
}

int launch_split(const i2v_gemm_params& p, int vec4, int splits, int kps, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, 128), tiles_n = p.N / BIG_BN;
  const dim3 grid(tiles_m * tiles_n, splits), block(512);
  if (p.a_mode == I2V_A_CONV3X3)
    hipLaunchKernelGGL((gemm_big_kernel<128, 64, 2, I2V_A_CONV3X3, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps, tiles_m * tiles_n);
  else
    hipLaunchKernelGGL((gemm_big_kernel<128, 64, 2, I2V_A_PLAIN, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, true>), grid, block, 0, s, p,
                       tiles_n, kps, tiles_m * tiles_n);
  const int64_t groups = (int64_t)p.M * (p.N / 4);
  const int blocks = (int)(i2v_cdiv(groups, 256) < 2048 ? i2v_cdiv(groups, 256) : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, splits, vec4);
  const int rc = i2v_check_launch("i2v_gemm_f16(split-K)");
  return rc < 0 ? rc : 1;
}

// Deep-pipeline form for the problems that cannot fill the chip (the 8 x 8 level: 2048 rows): their K loop is a chain of
// nkt DMA round trips (1.3 - 1.5 us each with one tile in flight, whatever the tile size), so the tile is made SMALLER --
// 128 x 128 (32 KiB stages, four of them) or 128 x 256 (48 KiB, three) -- and the freed LDS holds two or three more K tiles
// in flight: the chain advances at round trip / 3 (or / 2), and the smaller tiles also spread the problem over more CUs.
// Returns the column tile (128 / 256) or 0.  I2V_GEMM_DEEP=0 turns it off.
int deep_plan(const i2v_gemm_params& p, int vec4) {
  static const int on = getenv("I2V_GEMM_DEEP") ? atoi(getenv("I2V_GEMM_DEEP")) : 1;
  if (!on || !vec4) return 0;
  if (p.a_mode != I2V_A_PLAIN || p.epilogue != I2V_EPI_NONE || p.store_mode == I2V_STORE_VT_T || p.ln_wsum != nullptr) return 0;
  if (p.rows_per_w > 0 || p.a_perm_frames > 0 || p.c_is_f32) return 0;
  static const int min_kt = getenv("I2V_GEMM_DEEP_MINKT") ? atoi(getenv("I2V_GEMM_DEEP_MINKT")) : 10;
  if (p.N % 128 != 0 || (p.K + 63) / 64 < min_kt) return 0;
  const int64_t tm = i2v_cdiv(p.M, 128);
  if (tm * (p.N / 128) <= 256) return 128;
  if (p.N % 256 == 0 && tm * (p.N / 256) <= 256) return 256;
  return 0;
}

template <int BN, int NS>
int launch_deep(const i2v_gemm_params& p, hipStream_t s) {
  const int tiles_m = (int)i2v_cdiv(p.M, 128), tiles_n = p.N / BN;
  const int ntiles = tiles_m * tiles_n;
#define I2V_DEEP_LAUNCH(STORE) \
  hipLaunchKernelGGL((gemm_big_kernel<128, 64, NS, I2V_A_PLAIN, I2V_EPI_NONE, STORE, false, true, false, false, BN>), \
                     dim3(ntiles), dim3(512), 0, s, p, tiles_n, 0, ntiles)
  if (p.residual_lo != nullptr || p.c_lo != nullptr) {
    if (p.store_mode == I2V_STORE_ROWPERM)
      hipLaunchKernelGGL((gemm_big_kernel<128, 64, NS, I2V_A_PLAIN, I2V_EPI_NONE, I2V_STORE_ROWPERM, false, true, false, false, BN, 8, false, true>),
                         dim3(ntiles), dim3(512), 0, s, p, tiles_n, 0, ntiles);
    else
      hipLaunchKernelGGL((gemm_big_kernel<128, 64, NS, I2V_A_PLAIN, I2V_EPI_NONE, I2V_STORE_ROWMAJOR, false, true, false, false, BN, 8, false, true>),
                         dim3(ntiles), dim3(512), 0, s, p, tiles_n, 0, ntiles);
  } else if (p.store_mode == I2V_STORE_ROWPERM)
    I2V_DEEP_LAUNCH(I2V_STORE_ROWPERM);
  else if (p.store_mode == I2V_STORE_VT)
    I2V_DEEP_LAUNCH(I2V_STORE_VT);
  else
    I2V_DEEP_LAUNCH(I2V_STORE_ROWMAJOR);
#undef I2V_DEEP_LAUNCH
  const int rc = i2v_check_launch("i2v_gemm_f16(big, deep pipeline)");
  return rc < 0 ? rc : 1;
}

}  // namespace

int64_t i2v_gemm_big_workspace_bytes(const i2v_gemm_params& p, int vec4) {
  static const int off = getenv("I2V_GEMM_SPLITK") ? (atoi(getenv("I2V_GEMM_SPLITK")) == 0) : 0;
  if (off) return 0;
  int splits = splitk_plan(p, vec4, nullptr);
  const int s256 = splitk256_plan(p, vec4, nullptr);
  if (s256 > splits) splits = s256;
  return splits ? (int64_t)splits * p.M * p.N * (int64_t)sizeof(float) : 0;
}

namespace {
// columns per tile for this problem: 320 (UNet), or for 3x3 convolutions 256 / 128 (VAE channel counts); 0 = none
int big_bn(const i2v_gemm_params& p) {
  if (p.c_is_f32) return 0;   // fp32 results exist in the generic kernel only (narrow outputs)
  if (p.N % BIG_BN == 0) return BIG_BN;
  if (p.a_mode == I2V_A_PLAIN) return deep_plan(p, 1) ? -1 : 0;   // -1: the deep-pipeline form (128 / 256 columns) or nothing
  static const int vae = getenv("I2V_GEMM_BIG_VAE") ? atoi(getenv("I2V_GEMM_BIG_VAE")) : 1;
  if (vae && p.a_mode == I2V_A_CONV3X3 && p.epilogue == I2V_EPI_NONE && p.store_mode == I2V_STORE_ROWMAJOR && !p.residual_lo && !p.c_lo) {
    if (p.N % 256 == 0) return 256;
    if (p.N % 128 == 0) return 128;
  }
  return 0;
}

// dispatch decision for one problem: 0 = not for this kernel, 256 / 128 = tile height, -1 = split-K (*splits, *kps set)
int big_plan(const i2v_gemm_params& p, int vec4, int* splits_out, int* kps_out) {
  static const int mode = getenv("I2V_GEMM_BIG") ? atoi(getenv("I2V_GEMM_BIG")) : -1;  // 0 off, 256 / 128 force
  if (mode == 0) return 0;
  const int bn = big_bn(p);
  if (bn == 0) return 0;
  // the kernel addresses each operand through a buffer descriptor with 32-bit byte offsets (< 2 GiB per operand) and
  // moves whole BK-deep K tiles (K, the concat split and the conv channel count must be multiples of 64)
  if (p.K % 64 != 0 || (p.a2 != nullptr && p.k_split % 64 != 0)) return 0;
  if ((int64_t)p.N * p.ldw >= (1ll << 30)) return 0;
  if (p.a_mode == I2V_A_CONV3X3) {
    if (p.cin % 64 != 0 || (int64_t)p.n_img * p.in_h * p.in_w * p.lda >= (1ll << 30)) return 0;
    if (p.conv_kblock != 0 && p.conv_kblock != 64) return 0;
  } else if ((int64_t)p.M * p.lda >= (1ll << 30) || (int64_t)p.M * p.lda2 >= (1ll << 30)) {
    return 0;
  }
  // the specialised epilogue handles the vector (8-byte) forms only; anything else stays on the generic kernel
  if (!vec4 || p.epilogue == I2V_EPI_GELU) return 0;
  if (p.a_mode == I2V_A_CONV3X3 && (p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR)) return 0;
  if (p.epilogue == I2V_EPI_GEGLU && p.store_mode != I2V_STORE_ROWMAJOR) return 0;
  if (p.store_mode == I2V_STORE_ROWMAJOR || p.store_mode == I2V_STORE_ROWPERM) {
    // the row-contiguous epilogue moves 16 bytes per lane
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (p.ldc % 8 != 0 || !a16(p.c)) return 0;
    if (p.residual && (p.ldr % 8 != 0 || !a16(p.residual))) return 0;
    if (!a16(p.residual_lo) || !a16(p.c_lo)) return 0;
    if (p.rowvec && (p.ld_rowvec % 8 != 0 || !a16(p.rowvec))) return 0;
  }
  if (p.rowvec && p.M >= (1 << 24)) return 0;   // the epilogue's reciprocal division of the row index is exact below 2^24
  // the row-contiguous epilogue prefetches ONE added operand (residual or row vector) and addresses C and it with 32-bit
  // byte offsets (buffer ops)
  if (p.residual && p.rowvec && p.epilogue != I2V_EPI_GEGLU) return 0;
  const int64_t rv_rows = p.rowvec_period > 0 ? p.rowvec_period : (p.M - 1) / (p.rows_per_vec > 0 ? p.rows_per_vec : 1) + 1;
  if ((int64_t)p.M * p.ldc >= (1ll << 30) || (p.residual && (int64_t)p.M * p.ldr >= (1ll << 30)) ||
      (p.rowvec && rv_rows * p.ld_rowvec >= (1ll << 30)))
    return 0;
  if (p.rows_per_w > 0 || p.a_perm_frames > 0) {
    // per-batch weights / the permuted A gather exist only in the full-tile DMA path of the un-split kernel
    if (p.a_mode != I2V_A_PLAIN || p.a2 != nullptr || p.M % 256 != 0) return 0;
    if (p.rows_per_w > 0 && (p.rows_per_w % 256 != 0 || p.M % p.rows_per_w != 0 ||
                             (int64_t)(p.M / p.rows_per_w) * p.w_batch_stride >= (1ll << 30)))
      return 0;
    if (p.a_perm_frames > 0) {
      const int f = p.a_perm_frames;
      if ((f & (f - 1)) != 0 || f > 64 || p.a_perm_hw <= 0 || p.a_perm_hw % (64 / f) != 0 ||
          ((int64_t)f * p.a_perm_hw) % 256 != 0 || p.M % (f * p.a_perm_hw) != 0)
        return 0;
    }
  }
  const int64_t tn = bn > 0 ? p.N / bn : i2v_cdiv(p.N, BIG_BN);
  const int64_t t256 = i2v_cdiv(p.M, 256) * tn, t128 = i2v_cdiv(p.M, 128) * tn;
  if (mode == 256 || mode == 128) return bn > 0 ? mode : 0;
  static const int min_k = getenv("I2V_GEMM_BIG_MINK") ? atoi(getenv("I2V_GEMM_BIG_MINK")) : 128;
  if (p.a_mode != I2V_A_CONV3X3 && p.K < min_k) return 0;   // a single K tile cannot hide its own DMA latency
  // one 8-wave block per CU: a tile count just above a multiple of 256 wastes most of the last round.  Pick the
  // tile height by (fill of the last round) x (measured relative rate: 256-row 1.0, 128-row 0.82,
  // profiles/r1_tile_sweep.txt); below 40 % the 3-blocks-per-CU kernel of gemm.hip is faster.
  {   // 256-row tiles with K split (returns -2)
    int kps256 = 0;
    const int s256 = splitk256_plan(p, vec4, &kps256);
    if (s256 && p.workspace && p.workspace_bytes >= (int64_t)s256 * p.M * p.N * (int64_t)sizeof(float) &&
        (reinterpret_cast<uintptr_t>(p.workspace) % 16) == 0) {
      if (splits_out) *splits_out = s256;
      if (kps_out) *kps_out = kps256;
      return -2;
    }
  }
  const double e256 = (double)t256 / (double)(i2v_cdiv(t256, 256) * 256);
  const double e128 = 0.82 * (double)t128 / (double)(i2v_cdiv(t128, 256) * 256);
  if (bn == BIG_BN || bn == -1) {   // a single, partial round of tiles and a chain of >= 10 K tiles: the deep-pipeline form
    static const int deep_mode = getenv("I2V_GEMM_DEEP") ? atoi(getenv("I2V_GEMM_DEEP")) : 2;
    static const int max_kt = getenv("I2V_GEMM_DEEP_MAXKT") ? atoi(getenv("I2V_GEMM_DEEP_MAXKT")) : 63;
    const int dbn = deep_plan(p, vec4);
    if (dbn && (p.K + 63) / 64 <= max_kt && (deep_mode == 2 || bn == -1 || (e256 < 0.40 && e128 < 0.40)))
      return -3 - (dbn == 256 ? 1 : 0);   // -3: 128-column tiles, -4: 256-column tiles
    if (bn == -1) return 0;
  }
  if (e256 >= e128 && e256 >= 0.40) return 256;
  if (e128 >= 0.40) return 128;
  if (bn != BIG_BN) return 0;   // (split-K exists for 320-column tiles only)

  // too few output tiles for the chip: split K when the caller supplied the fp32 scratch
  int kps = 0;
  const int splits = splitk_plan(p, vec4, &kps);
  if (splits && p.workspace && p.workspace_bytes >= (int64_t)splits * p.M * p.N * (int64_t)sizeof(float) &&
      (reinterpret_cast<uintptr_t>(p.workspace) % 16) == 0) {
    if (splits_out) *splits_out = splits;
    if (kps_out) *kps_out = kps;
    return -1;
  }
  return 0;
}
}  // namespace

// LayerNorm-folded problems (ln_wsum set) run only on the un-split 8-wave kernel, through the epilogues that
// implement the fold: the row-contiguous staged stores (row-major / row-permuted, plain or GEGLU) and the fast V^T form.
int i2v_gemm_big_ln_ok(const i2v_gemm_params& p, int vec4) {
  const int plan = big_plan(p, vec4, nullptr, nullptr);
  if (plan != 256 && plan != 128) return 0;
  if (p.a_mode != I2V_A_PLAIN || p.a2 != nullptr) return 0;   // the K loop must stream whole rows of ONE source
  if (p.rowvec && p.rowvec_period > 0 && (p.rowvec_period & (p.rowvec_period - 1)) != 0) return 0;
  if (p.residual) return 0;   // no consumer of a folded LayerNorm adds a residual; the LNF epilogues do not carry the path
  if (p.store_mode == I2V_STORE_ROWMAJOR || p.store_mode == I2V_STORE_ROWPERM)
    return (p.epilogue == I2V_EPI_NONE || (p.epilogue == I2V_EPI_GEGLU && !p.rowvec)) ? 1 : 0;
  if (p.store_mode == I2V_STORE_VT_T)
    return (p.vt_len % 8 == 0 && p.M % 8 == 0 && p.vt_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(p.c) & 15) == 0 &&
            (!p.rowvec || (p.rowvec_period >= 8 && (p.rowvec_period & (p.rowvec_period - 1)) == 0 && p.ld_rowvec % 8 == 0)))
               ? 1 : 0;
  return 0;
}

// rows per block of the GroupNorm partials the un-split convolution's epilogue writes (gn_partial), 0: not this problem
int i2v_gemm_big_gn_rows(const i2v_gemm_params& p, int vec4) {
  const int plan = big_plan(p, vec4, nullptr, nullptr);
  if (plan != 256 && plan != 128) return 0;
  if (p.a_mode != I2V_A_CONV3X3 || big_bn(p) != BIG_BN || p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR) return 0;
  if (p.residual || p.ln_wsum || p.c_is_f32 || p.rows_per_w > 0 || p.a_perm_frames > 0) return 0;
  // the partials describe acc + bias + rowvec: with an output scale the stored tensor is another one
  if (p.out_scale != 1.0f) return 0;
  if (p.gn_groups <= 0 || p.N % p.gn_groups != 0) return 0;
  const int cpg = p.N / p.gn_groups, hw = p.out_h * p.out_w;
  if (cpg <= 0 || BIG_BN % cpg != 0 || p.M % plan != 0 || hw % plan != 0) return 0;
  if (p.rowvec && (p.rowvec_period > 0 || p.rows_per_vec <= 0 || p.rows_per_vec % plan != 0)) return 0;
  return plan;
}

// 1 if the 8-wave kernel takes this problem un-split (per-batch weights and the permuted A gather exist only there)
int i2v_gemm_big_unsplit_ok(const i2v_gemm_params& p, int vec4) {
  const int plan = big_plan(p, vec4, nullptr, nullptr);
  return plan == 256 || plan == 128;
}

#ifdef I2V_VARIANTS
int i2v_gemm_alt_try(const i2v_gemm_params& p, hipStream_t s);   // variants/gemm_alt.hip: short-K problems, alternating wave groups
#endif

int i2v_gemm_big_try(const i2v_gemm_params& p, int vec4, hipStream_t s) {
  int splits = 0, kps = 0;
  const int plan = big_plan(p, vec4, &splits, &kps);
  const int bn = big_bn(p);
#ifdef I2V_VARIANTS
  if (bn == BIG_BN && (plan == 256 || plan == 128)) {
    const int rc = i2v_gemm_alt_try(p, s);
    if (rc != 0) return rc;
  }
#endif
  if (bn == 256 && plan == 256) return launch_big_conv_bn<256, 256>(p, s);
  if (bn == 256 && plan == 128) return launch_big_conv_bn<128, 256>(p, s);
  if (bn == 128 && plan == 256) return launch_big_conv_bn<256, 128>(p, s);
  if (bn == 128 && plan == 128) return launch_big_conv_bn<128, 128>(p, s);
#ifdef I2V_VARIANTS
  if ((plan == 256 || plan == 128) && use_4wave(p, plan)) return launch_big4<I2V_A_PLAIN>(p, s);
#endif
  if (plan == 256) return launch_big<256>(p, vec4, s);
  if (plan == 128) return launch_big<128>(p, vec4, s);
  if (plan == -1) return launch_split(p, vec4, splits, kps, s);
  if (plan == -2) return launch_split256(p, vec4, splits, kps, s);
  if (plan == -3) return launch_deep<128, 4>(p, s);
  if (plan == -4) return launch_deep<256, 3>(p, s);
  return 0;
}
