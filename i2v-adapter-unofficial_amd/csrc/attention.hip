// Flash-style attention forward for gfx950: K1 cross-frame adapter attention (all frames of a clip attend to the
// frame-0 K/V), K2 spatial self-attention, K3 text / IP-Adapter cross-attention.
//
// One workgroup = 4 waves; each wave owns QT 16-row query tiles of one (batch, head); the KVT-key K tile and V^T tile
// are staged once per workgroup in LDS and shared by the 4 waves (for K1 the same frame-0 K/V tile is additionally
// shared through L2 by the workgroups of all frames of the clip: kv batch = q batch / group).
//
// MFMA orientation (v_mfma_f32_16x16x32_f16, D = A * B):
//   S^T tile [16 keys x 16 queries] = K[16 x d] * Q^T[d x 16]        A = K fragment (LDS), B = Q fragment (registers)
//   O^T tile [16 d    x 16 queries] = V^T[16 x 32 keys] * P^T[32 x 16] A = V^T fragment (LDS), B = P (registers)
// With S^T in the accumulator, lane (g = lane >> 4, c = lane & 15) holds scores of query c for 4 keys: the softmax
// row reduction is in-register max3 chains plus two cross-lane VALU swaps (v_permlane16/32_swap), the per-query rescale
// of O^T is lane-local, and the accumulator of S^T *is* the B operand of the PV product (no LDS round trip, no lane
// movement): the row->key assignment of S^T tile `kt` is key = 32 (kt >> 1) + 8 g + 4 (kt & 1) + r so that the 8
// values a lane holds for k-step s are keys 32 s + 8 g + 0..7, i.e. one contiguous 16-byte read of a V^T row.
// V arrives already transposed ([channel][key], written by the projection GEMM's VT epilogue).
//
// Softmax cost (the d = 40 level is VALU-bound, not MFMA-bound; rocprof PMC: VALU 81 % busy, MFMA 24 %):
//   * no arithmetic between the QK^T MFMA and exp2: scale * log2 e is folded into the Q fragments once per workgroup,
//     and -m (the running max, taken BEFORE the tile: deferred max, guide T13) is the initial accumulator of the MFMA
//     chain, which therefore ends with s' = s * scale * log2 e - m.  The O / l rescale runs only when a tile's max
//     exceeds m by more than 2^8 (and on the first tile), so P <= 256 (exact in the fp32 accumulation, 11-bit
//     relative in the fp16 P operand as always);
//   * K / V^T tiles are prefetched with raw buffer loads (scalar tile offset, hardware range check instead of exec
//     masks): the hot loop's prefetch costs no VALU and nothing waits on it before the tile's MFMAs;
//   * the row sum l is not accumulated on the VALU when head_dim leaves a spare row in the 16-row V^T padding
//     (40 -> 48): that row of the LDS V^T tile is set to 1.0, so the PV MFMA itself produces sum(P~) with the SAME
//     fp16-rounded P~ that multiplies V (numerator and denominator consistent; P~ packed with one round-toward-zero
//     v_cvt_pkrtz per pair, whose bias cancels in the ratio);
//   * everything tile-invariant of the K / V^T staging is hoisted, the partial last key tile is peeled.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr float DEFER_THR = 8.0f;    // log2 units

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ float xor16_max(float x) { return lane_xor16_max(x); }   // common.h
__device__ __forceinline__ float xor32_max(float x) { return lane_xor32_max(x); }

__device__ __forceinline__ uint32_t pack_rtz(float a, float b) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);   // v_cvt_pkrtz_f16_f32: two floats -> packed half2, one instruction
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t pack_rn(float a, float b) {
  const f16x2 h = {(f16)a, (f16)b};
  return __builtin_bit_cast(uint32_t, h);
}

// KVT: keys per tile (64 or 128).  SPARE: head_dim < DPV, i.e. V^T row `head_dim` is free to hold the ones that make
// the MFMA compute the row sum.
// The head_dim 40 variant (the 64 x 64 level: 80 % of the attention time) is held to 128 VGPRs = 4 waves / SIMD (its
// few spills land outside the key loop): +5 % over the 154-VGPR / 3-wave build.  d = 64 spills inside the loop at 128.
#ifndef I2V_ATTN_XCD_REMAP
#define I2V_ATTN_XCD_REMAP 1
#endif
// WALK (single-key-tile problems only: the 77 text tokens): the workgroup stages K / V^T ONCE and walks `qbw` consecutive
// 64 QT-query blocks.  Those launches were 8192 workgroups of 42 MFMAs each behind ~800 VALU instructions of set-up per wave
// (descriptors, staging offsets, the tail mask, 64-bit addresses: 19.6 VALU per MFMA, a VALU-bound kernel,
// profiles/r4_pmc_summary.txt); the set-up is now paid once per qbw blocks.  A template parameter: the long-sequence kernels
// keep their single trip (and their register allocation).
// (r6 A/B, profiles/r6_attn_d80_occupancy.txt: waves per SIMD requested for the head_dim-80 kernel -- 1 lets the compiler keep its 244
//  registers = 2 waves; 3 caps it at 168)
#ifndef I2V_ATTN_OCC_D80
#define I2V_ATTN_OCC_D80 1
#endif
template <int DQK, int DPV, int QT, int KVT, bool SPARE, bool WALK = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(DQK == 64 && DPV == 48 && QT == 2 && KVT == 64 && !WALK ? 4 : (DQK == 96 && DPV == 80 && !WALK ? I2V_ATTN_OCC_D80 : 1))))
void attn_kernel(const i2v_attn_params p, const float scale_log2, const int qbw_arg) {
  const int qbw = WALK ? qbw_arg : 1;
  constexpr int KS = DQK + 8;          // K LDS row stride (halfs)
  constexpr int VS = KVT + 8;          // V^T LDS row stride (halfs)
  constexpr int KSTEPS = DQK / 32;     // k-steps of the QK^T product
  constexpr int DT = DPV / 16;         // d tiles of O^T
  constexpr int NKT = KVT / 16;        // 16-key S^T tiles per key tile
  constexpr int NS2 = KVT / 32;        // k-steps of the PV product
  constexpr int KCH = DQK / 8;         // 16-byte chunks per K row
  constexpr int VCH = KVT / 8;         // 16-byte chunks per V^T row
  static_assert((KVT * KCH) % 256 == 0, "the K tile must be whole 16-byte chunks per thread");
  constexpr int NKC = (KVT * KCH) / 256;                // K chunks per thread
  constexpr int NVC = (DPV * VCH + 255) / 256;          // V^T chunks per thread (last pass may be partial)
  // two stages: one barrier per key tile (WALK: the one tile there is needs one stage)
  constexpr int NSTG = WALK ? 1 : 2;
  __shared__ __attribute__((aligned(16))) f16 sKb[NSTG][KVT * KS];
  __shared__ __attribute__((aligned(16))) f16 sVb[NSTG][DPV * VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin in launch order, so with the natural order the
  // query blocks of one (batch, head) -- which all stream the same K / V^T -- land on all 8 L2s and each L2 holds a slice
  // of EVERY pair in flight.  Re-deal: XCD x takes pairs x, x + 8, ... and walks all query blocks of a pair before the
  // next, so a pair's K / V^T (655 KB at the 64 x 64 level) is fetched into one L2 once.  Speed only.
  int qb = blockIdx.x, h = blockIdx.y, bq = blockIdx.z;
  {
    const int nqb = gridDim.x, pairs = gridDim.y * gridDim.z;
    if (I2V_ATTN_XCD_REMAP && p.lk <= 128 && (nqb * (int)gridDim.z) % 8 == 0) {
      // short key sequences (the 77 text tokens): K / V^T are a few KB, the traffic is Q and O, whose rows interleave the
      // heads (80-byte slices of 640-byte rows at d = 40).  Here an XCD walks the HEADS of one query block back to back, so
      // that the slices of a row meet in one L2: each line of Q is fetched from HBM once and each line of O leaves whole.
      const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = lin & 7, slot = lin >> 3;
      h = slot % (int)gridDim.y;
      const int unit = (slot / (int)gridDim.y) * 8 + xcd;
      qb = unit % nqb;
      bq = unit / nqb;
    } else if (I2V_ATTN_XCD_REMAP && pairs % 8 == 0) {
      const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = lin & 7, slot = lin >> 3;
      const int pair = (slot / nqb) * 8 + xcd;
      qb = slot % nqb;
      h = pair % (int)gridDim.y;
      bq = pair / (int)gridDim.y;
    }
  }
  const int bkv = bq / p.kv_group;
  const int d = p.head_dim, lq = p.lq, lk = p.lk;
  const int qb_first = qb * qbw;
  int q0 = qb_first * (64 * QT) + wave * (16 * QT);     // first query row of this wave in the current block

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(p.q) + (int64_t)bq * p.q_batch_stride + h * d;
  const f16* __restrict__ Kg = reinterpret_cast<const f16*>(p.k) + (int64_t)bkv * p.k_batch_stride + h * d;
  const f16* __restrict__ Vg =
      reinterpret_cast<const f16*>(p.vt) + (int64_t)bkv * p.vt_batch_stride + (int64_t)h * d * p.vt_row_stride;

  // ---- Q fragments (B operand of QK^T), resident in registers for the whole key loop
  // (loading them BEHIND the first K / V^T tile's loads -- one round trip instead of two -- measured nothing on the 77-token
  //  text attention, 85 vs 86 us, and cost the d = 40 kernel 6 more spilled registers)
  f16x8 qf[QT][KSTEPS];
  f32x4 o[DT][QT];
  // negm[j] = -(running max of query column j) in all four registers: it is the INITIAL ACCUMULATOR of the QK^T MFMA
  // chain, so the chain ends with s - m and the softmax needs neither a multiply nor a subtraction per score
  f32x4 negm[QT];
  float lrow[QT];
  // WALK: the NEXT block's rows are requested (raw, qraw) before this block is computed, so their round trip runs under
  // its MFMAs, softmax and stores instead of in front of the next block (one block per ~7 us and workgroup without it)
  f16x8 qraw[QT][KSTEPS];
  auto fetch_q = [&](int q0f) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int row = q0f + qt * 16 + l15;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const int dd = 32 * s + 8 * g;
        qraw[qt][s] = (row < lq && dd < d) ? ld_global_16B(Q + (int64_t)row * p.q_row_stride + dd) : zero8();
      }
    }
  };
  auto load_q = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        f16x8 v = qraw[qt][s];
        // fold scale * log2(e) into Q once, so that the QK^T accumulator already holds base-2 logits
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] * scale_log2);
        qf[qt][s] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int j = 0; j < QT; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      negm[j] = f32x4{0.f, 0.f, 0.f, 0.f};   // finite: the first tile always takes the rescale path (sets the true max)
      lrow[j] = 0.f;
    }
  };
  fetch_q(q0);

  // ---- K / V^T staging: everything that does not depend on the tile index is hoisted (per-thread source pointers,
  //      LDS offsets, validity of the chunk inside head_dim); registers of chunks outside head_dim stay 0 (or 1.0
  //      for the row-sum row) for the whole kernel.  Only the last, partial key tile takes the masked loader.
  //      Loads are raw buffer loads through wave-uniform descriptors of this (batch, head)'s K and V^T slices: a
  //      32-bit per-lane byte offset computed once + a scalar tile offset, and the hardware range check instead of
  //      exec masks (a lane whose chunk lies outside head_dim carries an out-of-range offset and receives 0; K rows
  //      of keys >= lk are past the end of the slice and come back 0 too).  The hot loop therefore issues its
  //      prefetch with no VALU work and no branches, and nothing waits on it before the tile's MFMAs.
  constexpr int OOB = 0x40000000;      // > any slice extent (checked on the host), no wrap with the tile offset
  f16x8 rk[NKC], rv[NVC];
  int k_off[NKC], v_off[NVC], k_lds[NKC], v_lds[NVC];
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(Kg), 0, (int)(((int64_t)(lk - 1) * p.k_row_stride + d) * 2), 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(Vg), 0, (int)(((int64_t)(d - 1) * p.vt_row_stride + ((lk + 7) & ~7)) * 2), 0x00020000);
#pragma unroll
  for (int i = 0; i < NKC; ++i) {
    const int id = tid + 256 * i;
    const int row = id / KCH, c = id - row * KCH;
    k_off[i] = 8 * c < d ? (int)((row * p.k_row_stride + 8 * c) * 2) : OOB;
    // (r5) key `row` of the tile sits in LDS row 16 kt + l15 of the S^T tile (kt, l15) that reads it -- key = 32 (kt >> 1) + 8 (l15 >> 2)
    // + 4 (kt & 1) + (l15 & 3), i.e. bit 2 of the key moves to bit 4 of the LDS row -- so that the 16 rows of a fragment read are
    // CONSECUTIVE: with the natural order they were rows b .. b + 3, b + 8 .., b + 16 .., b + 24 .., and rows 16 apart share their
    // banks whatever the row stride (two-way conflicts on every K fragment read; SQ_LDS_BANK_CONFLICT 0.39 of the LDS cycles)
    const int lrow = (row & ~28) | ((row & 4) << 2) | ((row & 24) >> 1);
    k_lds[i] = lrow * KS + 8 * c;
  }
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    const int id = tid + 256 * i;
    const int row = id / VCH, c = id - row * VCH;
    v_off[i] = row < d ? (int)((row * p.vt_row_stride + 8 * c) * 2) : OOB;
    // rows d < row < DPV are rewritten with the zeros the range check returns; lanes past the tile, and the lanes of
    // the all-ones row-sum row (written once below), park their store in the 16-byte pad of some row
    const bool real = row < DPV && !(SPARE && row == d);
    v_lds[i] = real ? row * VS + 8 * c : (tid % DPV) * VS + KVT;
  }
  if (SPARE && tid < NSTG * VCH) {   // masked keys carry P = 0, so 1.0 for every key is right
    f16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (f16)1.f;
    *reinterpret_cast<f16x8*>(&sVb[tid / VCH][d * VS + 8 * (tid % VCH)]) = ones;
  }
  const int k_tile_bytes = (int)(KVT * p.k_row_stride * 2);

  auto issue = [&](int t) {
    const int ks = t * k_tile_bytes, vs = t * (KVT * 2);
#pragma unroll
    for (int i = 0; i < NKC; ++i)
      rk[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_off[i], ks, 0));
#pragma unroll
    for (int i = 0; i < NVC; ++i)
      rv[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, v_off[i], vs, 0));
  };
  // the last, partial key tile: V^T entries of keys >= lk continue into row padding / the next row, so they are
  // zeroed by hand (P = 0 for those keys, but 0 * garbage must not be NaN)
  auto mask_tail_v = [&](int t) {
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int key0 = t * KVT + 8 * ((tid + 256 * i) % VCH);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (key0 + e >= lk) rv[i][e] = (f16)0.f;
    }
  };
  auto commit = [&](int stage) {
    f16* sK = sKb[stage];
    f16* sV = sVb[stage];
#pragma unroll
    for (int i = 0; i < NKC; ++i) *reinterpret_cast<f16x8*>(&sK[k_lds[i]]) = rk[i];
#pragma unroll
    for (int i = 0; i < NVC; ++i) *reinterpret_cast<f16x8*>(&sV[v_lds[i]]) = rv[i];
  };

  const int ntiles = (lk + KVT - 1) / KVT;
  const bool partial = (lk % KVT) != 0;
  issue(0);
  if (ntiles == 1 && partial) mask_tail_v(0);
  commit(0);
  __syncthreads();

  // one key tile: TAIL = the last tile when lk % KVT != 0 (scores of keys >= lk are masked to -inf); MORE = a next
  // tile exists and is prefetched into registers while this one is computed
  auto process = [&](auto tail_c, auto more_c, const int t) {
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool more = decltype(more_c)::value;
    if (more) issue(t + 1);
    const f16* sK = sKb[t & 1];
    const f16* sV = sVb[t & 1];

    // ---- S^T = K Q^T
    f32x4 sacc[NKT][QT];
#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int krow = 16 * kt + l15;          // (the staging permutes the keys so that a tile's 16 keys are consecutive LDS rows)
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[krow * KS + 32 * s + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) sacc[kt][j] = mfma16x16x32(kf, qf[j][s], s == 0 ? negm[j] : sacc[kt][j]);
      }
    }

#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- online softmax with deferred max (per query column; keys over registers and the 4 lane groups)
    const int key_base = t * KVT;
    const bool first = t == 0;
    f16x8 pf[QT][NS2];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float sv[NKT][4];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[kt][j][r];   // = s * scale * log2(e) - m_prev
          if (TAIL) {
            const int key = key_base + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r;
            if (key >= lk) v = -INFINITY;
          }
          sv[kt][r] = v;
        }
      // the lane's own 4 NKT keys as a TREE of three-input maxima: (4 NKT - 1) / 2 instructions (8 for 16 keys) and a dependency
      // chain of 3 -- as a running maximum per key tile the compiler emitted 6 v_max3 + 6 v_max per query tile (r5: -8 VALU
      // instructions per 32-query x 64-key trip of a loop that is bound by its instruction count)
      float red[4 * NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[4 * kt + r] = sv[kt][r];
      {
        int n = 4 * NKT;
#pragma unroll
        for (int level = 0; level < 5 && n > 1; ++level) {
          int m = 0;
#pragma unroll
          for (int i = 0; i + 2 < n; i += 3) red[m++] = max3(red[i], red[i + 1], red[i + 2]);
          if (n % 3 == 2) {
            red[m] = fmaxf(red[n - 2], red[n - 1]);
            ++m;
          } else if (n % 3 == 1) {
            red[m] = red[n - 1];
            ++m;
          }
          n = m;
        }
      }
      float mx = red[0];
      // mx = the lane's own 16 keys.  The test needs no cross-lane traffic: if NO lane of the wave holds a score above the
      // threshold, every P of the tile is <= 2^8 against the running max as it stands.  Only when some lane does (the first
      // tile, then rarely) is the row maximum completed over the four lane groups of the query
      // (v_permlane16_swap / v_permlane32_swap: cross-lane on the VALU, no LDS round trip) and the running max moved.
      if (__any(first || mx > DEFER_THR)) {          // wave-uniform; rare after the first tile
        mx = xor16_max(mx);
        mx = xor32_max(mx);
        const float dlt = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 1.0f : fast_exp2(-dlt);   // O is still zero on the first tile
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[j][r] -= dlt;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[kt][r] -= dlt;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[i][j][r] *= alpha;
        if (!SPARE) lrow[j] *= alpha;
      }
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[kt][r] = fast_exp2(sv[kt][r]);
          if (!SPARE) ls += sv[kt][r];
        }
      if (!SPARE) lrow[j] += ls;
#pragma unroll
      for (int s2 = 0; s2 < NS2; ++s2) {
        u32x4 w;
        if (SPARE) {
          w[0] = pack_rtz(sv[2 * s2][0], sv[2 * s2][1]);
          w[1] = pack_rtz(sv[2 * s2][2], sv[2 * s2][3]);
          w[2] = pack_rtz(sv[2 * s2 + 1][0], sv[2 * s2 + 1][1]);
          w[3] = pack_rtz(sv[2 * s2 + 1][2], sv[2 * s2 + 1][3]);
        } else {
          w[0] = pack_rn(sv[2 * s2][0], sv[2 * s2][1]);
          w[1] = pack_rn(sv[2 * s2][2], sv[2 * s2][3]);
          w[2] = pack_rn(sv[2 * s2 + 1][0], sv[2 * s2 + 1][1]);
          w[3] = pack_rn(sv[2 * s2 + 1][2], sv[2 * s2 + 1][3]);
        }
        pf[j][s2] = __builtin_bit_cast(f16x8, w);
      }
    }

    // ---- O^T += V^T P^T   (with SPARE, row `d` of V^T is all ones: O^T[d][q] accumulates the row sum)
#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&sV[(i * 16 + l15) * VS + 32 * s2 + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) o[i][j] = mfma16x16x32(vf, pf[j][s2], o[i][j]);
      }

#ifdef I2V_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (more) {
      if (partial && t + 2 == ntiles) mask_tail_v(t + 1);
      commit((t + 1) & 1);   // the other stage was last read in iteration t - 1 (closed by its barrier)
      __syncthreads();
    }
  };

  for (int qi = 0; qi < qbw; ++qi) {
    if (WALK) {
      if ((qb_first + qi) * (64 * QT) >= lq) break;      // workgroup-uniform
      q0 = (qb_first + qi) * (64 * QT) + wave * (16 * QT);
    }
    load_q();
    if (WALK && qi + 1 < qbw) fetch_q(q0 + 64 * QT);     // (rows >= lq: zeros, never used)
    for (int t = 0; t + 1 < ntiles; ++t) process(std::false_type{}, std::true_type{}, t);
    if (partial)
      process(std::true_type{}, std::false_type{}, ntiles - 1);
    else
      process(std::false_type{}, std::false_type{}, ntiles - 1);

    // ---- normalise and store: lane holds O[query l15][d = 16 i + 4 g + r]
    f16* __restrict__ O = reinterpret_cast<f16*>(p.o) + (int64_t)bq * p.o_batch_stride + h * d;
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float lt;
      if (SPARE) {
        // the sum sits in O^T row d: tile d >> 4, lane group (d & 15) >> 2, register d & 3
        float cand = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (i == (d >> 4) && r == (d & 3)) cand = o[i][j][r];
        lt = __shfl(cand, (((d & 15) >> 2) << 4) | l15, 64);
      } else {
        lt = lrow[j];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
      }
      const float inv = 1.0f / lt;
      const int row = q0 + j * 16 + l15;
      if (row >= lq) continue;
      // training forward: log2-sum-exp of the row = running reference + log2(sum of P against it) (v_log_f32 is log2); the four
      // lane groups of a query hold the same reference, lane group 0 writes
      if (p.lse != nullptr && g == 0)
        p.lse[((int64_t)bq * p.heads + h) * lq + row] = __builtin_amdgcn_logf(lt) - negm[j][0];
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const int dd = i * 16 + 4 * g;
        if (dd >= d) continue;
        f16* dst = O + (int64_t)row * p.o_row_stride + dd;
        f16x4 ov;
        if (p.accumulate) {
          const f16x4 prev = *reinterpret_cast<const f16x4*>(dst);
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (f16)((float)prev[r] + p.acc_scale * o[i][j][r] * inv);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (f16)(o[i][j][r] * inv);
        }
        *reinterpret_cast<f16x4*>(dst) = ov;
      }
    }
  }
}

inline bool al(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <int DQK, int DPV, bool SPARE>
int launch_q(const i2v_attn_params& p, hipStream_t s) {
  const float scale_log2 = p.scale * 1.4426950408889634f;
  // measured (profiles/r1_tile_sweep.txt): 2 query tiles per wave keep 2 waves / SIMD resident, so one wave's
  // softmax VALU overlaps the other's MFMAs; 4 tiles drop to 1 wave / SIMD and serialise the two pipes.
  // head_dim 160 (round 3, tools/attn_only.py): two tiles per wave win there too, although they leave one wave per SIMD
  // (256 + 52 registers): 256 x 256 keys 37 -> 32 us, 1024 x 1024 409 -> 330 us; 128 left at one tile (not on the UNet's path).
  int qt = 1;
  if (p.lq >= 128 && (DQK <= 96 || DQK == 160)) qt = 2;
  static const int qt_env = getenv("I2V_ATTN_QT") ? atoi(getenv("I2V_ATTN_QT")) : 0;   // tuning overrides
  static const int kvt_env = getenv("I2V_ATTN_KVT") ? atoi(getenv("I2V_ATTN_KVT")) : 0;
  if (qt_env == 1 || qt_env == 2) qt = qt_env;
  int kvt = 64;   // 128-key tiles (half the barriers per key) measured 2-4 % slower: the loop is VALU-bound, not sync-bound
  // the 77 text tokens (+ padding) fit ONE 96-key tile: a single trip through the key loop -- no second barrier round trip --
  // and 42 MFMAs / 24 exponentials per 32 queries instead of the 56 / 32 of two 64-key tiles (these launches are 8192
  // workgroups of almost nothing but prologue: 99.6 us at 64 x 64 for 168 MB of Q + O)
  // (head_dim 40 .. 64 only: 96 keys x DQK / 8 chunks must be a multiple of the 256 threads)
  if (DQK == 64 && p.lk > 64 && p.lk <= 96) kvt = 96;
  if (kvt_env == 64 || (kvt_env == 128 && DQK <= 64) || (kvt_env == 96 && DQK == 64)) kvt = kvt_env;
  const dim3 block(256);
  const int nqb = (int)i2v_cdiv(p.lq, 64 * qt);
  // one key tile covers the whole sequence (the text / image-prompt tokens): walk several query blocks per workgroup so that
  // about 1024 workgroups remain (see WALK)
  static const int walk_off = getenv("I2V_ATTN_WALK") ? (atoi(getenv("I2V_ATTN_WALK")) == 0) : 0;
  int qbw = 1;
  if (!walk_off && p.lk <= kvt) {
    const int64_t wgs = (int64_t)nqb * p.heads * p.batch_q;
    qbw = (int)(wgs / 1024);
    if (qbw > 8) qbw = 8;
    if (qbw > nqb) qbw = nqb;
    if (qbw < 1) qbw = 1;
  }
  // (measured and removed: O staged through a per-wave LDS slab and stored as 16-byte row chunks instead of the accumulator
  //  layout's 8-byte pieces -- 77.7 vs 73.6 us; a head-major layout of q / o (80-byte rows contiguous) -- 72.7 vs 73.6 us;
  //  with 4 keys instead of 77 the launch still takes 56 us against 24 us for a copy of Q to O: what is left is the per-block
  //  instruction count of the single-tile path, tools/attn77_probe.py)
  const bool walk = qbw > 1;
  const dim3 grid((unsigned)i2v_cdiv(nqb, qbw), (unsigned)p.heads, (unsigned)p.batch_q);
#define I2V_ATTN_LAUNCH(QTV, KVTV)                                                                                       \
  do {                                                                                                                   \
    if (walk)                                                                                                            \
      hipLaunchKernelGGL((attn_kernel<DQK, DPV, QTV, KVTV, SPARE, true>), grid, block, 0, s, p, scale_log2, qbw);        \
    else                                                                                                                 \
      hipLaunchKernelGGL((attn_kernel<DQK, DPV, QTV, KVTV, SPARE, false>), grid, block, 0, s, p, scale_log2, 1);         \
  } while (0)
  if constexpr (DQK == 64) {
    if (kvt == 96) {
      if (qt == 2) I2V_ATTN_LAUNCH(2, 96); else I2V_ATTN_LAUNCH(1, 96);
      return i2v_check_launch("i2v_attention_f16");
    }
  }
  if constexpr (DQK <= 64) {
    if (kvt == 128) {
      if (qt == 2) I2V_ATTN_LAUNCH(2, 128); else I2V_ATTN_LAUNCH(1, 128);
      return i2v_check_launch("i2v_attention_f16");
    }
  }
  if (qt == 2) I2V_ATTN_LAUNCH(2, 64); else I2V_ATTN_LAUNCH(1, 64);
#undef I2V_ATTN_LAUNCH
  return i2v_check_launch("i2v_attention_f16");
}

template <int DQK, int DPV>
int launch_d(const i2v_attn_params& p, hipStream_t s) {
  return p.head_dim < DPV ? launch_q<DQK, DPV, true>(p, s) : launch_q<DQK, DPV, false>(p, s);
}

}  // namespace

// Measured-and-rejected forms of the head_dim-40 kernel (profiles/r3_attn_pipe_ab.txt, r2_attn_fragment_layout_pmc.txt) live in
// csrc/variants/ and are compiled only into an A/B library (tools/build_variant.sh --variants: -DI2V_VARIANTS), where the
// I2V_ATTN32 / I2V_ATTN_PIPE switches select them.
#ifdef I2V_VARIANTS
int i2v_attention32_try(const i2v_attn_params& p, hipStream_t s);   // variants/attention32.hip: head_dim 40 / 48 on 32x32x16 MFMAs
int i2v_attention_pipe_try(const i2v_attn_params& p, hipStream_t s); // variants/attention_pipe.hip: software-pipelined key loop
#endif

extern "C" int i2v_attention_f16(const i2v_attn_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_attention_f16: null params");
  const i2v_attn_params& p = *pp;
  I2V_CHECK_ARG(p.q && p.k && p.vt && p.o, "i2v_attention_f16: null pointer");
  I2V_CHECK_ARG(p.batch_q > 0 && p.kv_group > 0 && p.batch_q % p.kv_group == 0,
                "i2v_attention_f16: batch_q (%d) must be a positive multiple of kv_group (%d)", p.batch_q, p.kv_group);
  I2V_CHECK_ARG(p.heads > 0 && p.lq > 0 && p.lk > 0, "i2v_attention_f16: heads, lq, lk must be positive");
  I2V_CHECK_ARG(!(p.lse != nullptr && p.accumulate), "i2v_attention_f16: lse output is not combined with accumulate");
  I2V_CHECK_ARG(p.head_dim > 0 && p.head_dim % 8 == 0 && p.head_dim <= 160,
                "i2v_attention_f16: head_dim (%d) must be a multiple of 8 and <= 160", p.head_dim);
  I2V_CHECK_ARG(p.q_row_stride % 8 == 0 && p.k_row_stride % 8 == 0 && p.q_batch_stride % 8 == 0 &&
                    p.k_batch_stride % 8 == 0,
                "i2v_attention_f16: q / k strides must be multiples of 8 elements");
  I2V_CHECK_ARG(p.vt_row_stride % 8 == 0 && p.vt_batch_stride % 8 == 0 && p.vt_row_stride >= ((p.lk + 7) / 8) * 8,
                "i2v_attention_f16: vt_row_stride must be a multiple of 8 and >= lk rounded up to 8");
  I2V_CHECK_ARG(p.o_row_stride % 4 == 0 && p.o_batch_stride % 4 == 0, "i2v_attention_f16: o strides must be multiples of 4");
  I2V_CHECK_ARG(al(p.q, 16) && al(p.k, 16) && al(p.vt, 16) && al(p.o, 8), "i2v_attention_f16: pointer alignment");
  I2V_CHECK_ARG(p.heads <= 65535 && p.batch_q <= 65535, "i2v_attention_f16: heads / batch_q exceed the grid limits");
  // the kernel addresses one (batch, head) slice of K and of V^T with 32-bit byte offsets (buffer loads)
  I2V_CHECK_ARG(p.k_row_stride > 0 && ((int64_t)p.lk + 128) * p.k_row_stride * 2 < (1ll << 30) &&
                    p.vt_row_stride > 0 && ((int64_t)p.head_dim * p.vt_row_stride + 256) * 2 < (1ll << 30),
                "i2v_attention_f16: one (batch, head) slice of K / V^T must span less than 1 GiB");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int d = p.head_dim;
  if (d <= 16) return launch_d<32, 16>(p, s);
  if (d <= 32) return launch_d<32, 32>(p, s);
#ifdef I2V_VARIANTS
  if (d > 32 && d < 48) {
    const int rc = i2v_attention_pipe_try(p, s);
    if (rc != 0) return rc < 0 ? rc : I2V_OK;
  }
  if (d > 32 && d <= 48) {
    const int rc = i2v_attention32_try(p, s);
    if (rc != 0) return rc < 0 ? rc : I2V_OK;
  }
#endif
  if (d <= 48) return launch_d<64, 48>(p, s);
  if (d <= 64) return launch_d<64, 64>(p, s);
  if (d <= 80) return launch_d<96, 80>(p, s);
  if (d <= 96) return launch_d<96, 96>(p, s);
  if (d <= 128) return launch_d<128, 128>(p, s);
  return launch_d<160, 160>(p, s);
}
