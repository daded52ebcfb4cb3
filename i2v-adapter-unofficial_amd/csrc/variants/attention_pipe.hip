// Flash attention forward for head_dim 40 (the 64 x 64 level of the UNet), three-stage software pipeline inside a wave.
// Same contract and the same arithmetic per element as attn_kernel<64, 48, 2, 64, true> (attention.hip): S^T = K Q^T with
// scale * log2 e folded into Q and -m as the initial accumulator, deferred max, P packed round-toward-zero, the row sum from
// the ones row of V^T; only the ORDER of independent work differs.
//
// attn_kernel runs, per key tile and wave, QK^T (16 MFMAs) -> softmax (VALU) -> PV (12 MFMAs) back to back and leaves the
// overlap of one wave's VALU with another's MFMAs to the four waves of a SIMD (mfma_busy 0.465,
// issue stall 45 % of wave cycles).  Here trip t of the key loop holds three INDEPENDENT pieces of work in one wave,
//     A: S^T(t + 1) = K(t + 1) Q^T      B: P(t) = softmax(S^T(t))      C: O^T += V^T(t - 1) P^T(t - 1)
// so that a wave's own instruction stream alternates MFMAs and VALU; two score tiles and two P tiles are live (about 190
// registers: 2 waves / SIMD, 2 workgroups / CU).  LDS: K in two stages (K(t + 2) lands in the stage K(t) was read from one
// trip earlier), V^T in four (V^T(t - 1) is read while (t), (t + 1) wait and (t + 2) lands); one barrier per trip.
// A change of the running max at tile t (rare after the first tiles: deferred by 2^8) rescales O AFTER trip t's C (whose P
// still refers to the old max) and reaches S^T(t + 1) through the initial accumulator, which A reads after the decision.
// Opt-in / default: see i2v_attention_pipe_try.
#include <cstdlib>
#include <type_traits>

#include "../common.h"

namespace {

constexpr float PIPE_DEFER_THR = 8.0f;   // log2 units, as attention.hip

__device__ __forceinline__ float pexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float pmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float pxor16_max(float x) { return lane_xor16_max(x); }   // common.h
__device__ __forceinline__ float pxor32_max(float x) { return lane_xor32_max(x); }
__device__ __forceinline__ uint32_t ppack_rtz(float a, float b) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
  return __builtin_bit_cast(uint32_t, h);
}

#ifndef I2V_PIPE_WAVES
#define I2V_PIPE_WAVES 2
#endif

// head_dim in (32, 48), lk % 64 == 0, lk >= 192 (checked by the launcher)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(I2V_PIPE_WAVES, I2V_PIPE_WAVES)))
void attn_pipe_kernel(const i2v_attn_params p, const float scale_log2) {
  constexpr int DQK = 64, DPV = 48, QT = 2, KVT = 64;
  constexpr int KS = DQK + 8, VS = KVT + 8;
  constexpr int KSTEPS = DQK / 32, DT = DPV / 16, NKT = KVT / 16, NS2 = KVT / 32;
  constexpr int KCH = DQK / 8, VCH = KVT / 8;
  constexpr int NKC = (KVT * KCH) / 256, NVC = (DPV * VCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) f16 sKb[2][KVT * KS];
  __shared__ __attribute__((aligned(16))) f16 sVb[4][DPV * VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  int qb = blockIdx.x, h = blockIdx.y, bq = blockIdx.z;
  {   // an XCD walks all query blocks of a (batch, head) pair before the next pair (as attn_kernel)
    const int nqb = gridDim.x, pairs = gridDim.y * gridDim.z;
    if (pairs % 8 == 0) {
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
  const int q0 = qb * (64 * QT) + wave * (16 * QT);

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(p.q) + (int64_t)bq * p.q_batch_stride + h * d;
  const f16* __restrict__ Kg = reinterpret_cast<const f16*>(p.k) + (int64_t)bkv * p.k_batch_stride + h * d;
  const f16* __restrict__ Vg =
      reinterpret_cast<const f16*>(p.vt) + (int64_t)bkv * p.vt_batch_stride + (int64_t)h * d * p.vt_row_stride;

  f16x8 qf[QT][KSTEPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int row = q0 + qt * 16 + l15;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int dd = 32 * s + 8 * g;
      f16x8 v = zero8();
      if (row < lq && dd < d) v = ld_global_16B(Q + (int64_t)row * p.q_row_stride + dd);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] * scale_log2);
      qf[qt][s] = v;
    }
  }

  f32x4 o[DT][QT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 negm[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) negm[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- K / V^T staging (as attn_kernel: raw buffer loads, out-of-range offsets for the padding chunks)
  constexpr int OOB = 0x40000000;
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
    k_lds[i] = row * KS + 8 * c;
  }
#pragma unroll
  for (int i = 0; i < NVC; ++i) {
    const int id = tid + 256 * i;
    const int row = id / VCH, c = id - row * VCH;
    v_off[i] = row < d ? (int)((row * p.vt_row_stride + 8 * c) * 2) : OOB;
    const bool real = row < DPV && row != d;   // row d: the all-ones row-sum row, written once below
    v_lds[i] = real ? row * VS + 8 * c : (tid % DPV) * VS + KVT;
  }
  if (tid < 4 * VCH) {
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
  auto commit = [&](int t) {
    f16* sK = sKb[t & 1];
    f16* sV = sVb[t & 3];
#pragma unroll
    for (int i = 0; i < NKC; ++i) *reinterpret_cast<f16x8*>(&sK[k_lds[i]]) = rk[i];
#pragma unroll
    for (int i = 0; i < NVC; ++i) *reinterpret_cast<f16x8*>(&sV[v_lds[i]]) = rv[i];
  };

  // ---- the three pieces of work
  auto qk = [&](int t, f32x4 (&sacc)[NKT][QT]) {   // A: S^T(t) - m
    const f16* sK = sKb[t & 1];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int krow = 32 * (kt >> 1) + 8 * (l15 >> 2) + 4 * (kt & 1) + (l15 & 3);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[krow * KS + 32 * s + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) sacc[kt][j] = mfma16x16x32(kf, qf[j][s], s == 0 ? negm[j] : sacc[kt][j]);
      }
    }
  };
  auto pv = [&](int t, const f16x8 (&pf)[QT][NS2]) {   // C: O^T += V^T(t) P^T(t)
    const f16* sV = sVb[t & 3];
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2)
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&sV[(i * 16 + l15) * VS + 32 * s2 + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) o[i][j] = mfma16x16x32(vf, pf[j][s2], o[i][j]);
      }
  };
  // B, first half: row max of tile t and the (rare) change of the running max.  Returns the O rescale factor per query
  // tile (1 = none); the scores of tile t are shifted here, tile t + 1 is shifted through negm (its QK^T comes after).
  auto decide = [&](bool first, f32x4 (&sacc)[NKT][QT], float (&alpha)[QT]) -> bool {
    bool any_rescale = false;
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float mx = pmax3(sacc[0][j][0], sacc[0][j][1], fmaxf(sacc[0][j][2], sacc[0][j][3]));
#pragma unroll
      for (int kt = 1; kt < NKT; ++kt)
        mx = fmaxf(pmax3(mx, sacc[kt][j][0], sacc[kt][j][1]), fmaxf(sacc[kt][j][2], sacc[kt][j][3]));
      alpha[j] = 1.0f;
      if (__any(first || mx > PIPE_DEFER_THR)) {   // wave-uniform; the row max is completed over the lane groups only here
        mx = pxor16_max(mx);
        mx = pxor32_max(mx);
        const float dlt = first ? mx : fmaxf(mx, 0.f);
        alpha[j] = first ? 1.0f : pexp2(-dlt);
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[j][r] -= dlt;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[kt][j][r] -= dlt;
        any_rescale = any_rescale || !first;
      }
    }
    return any_rescale;
  };
  auto rescale_o = [&](const float (&alpha)[QT]) {
#pragma unroll
    for (int j = 0; j < QT; ++j)
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[i][j][r] *= alpha[j];
  };
  // B, second half: P = exp2(S' - m), packed to fp16 in the B-operand layout of the PV product
  auto softmax_pack = [&](const f32x4 (&sacc)[NKT][QT], f16x8 (&pf)[QT][NS2]) {
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float e[NKT][4];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) e[kt][r] = pexp2(sacc[kt][j][r]);
#pragma unroll
      for (int s2 = 0; s2 < NS2; ++s2) {
        u32x4 w;
        w[0] = ppack_rtz(e[2 * s2][0], e[2 * s2][1]);
        w[1] = ppack_rtz(e[2 * s2][2], e[2 * s2][3]);
        w[2] = ppack_rtz(e[2 * s2 + 1][0], e[2 * s2 + 1][1]);
        w[3] = ppack_rtz(e[2 * s2 + 1][2], e[2 * s2 + 1][3]);
        pf[j][s2] = __builtin_bit_cast(f16x8, w);
      }
    }
  };

  const int ntiles = lk / KVT;
  // trip t: scores of tile t in `cur` (computed one trip earlier), P of tile t - 1 in `pin`
  auto trip = [&](int t, f32x4 (&cur)[NKT][QT], f32x4 (&nxt)[NKT][QT], f16x8 (&pout)[QT][NS2], const f16x8 (&pin)[QT][NS2]) {
    if (t + 2 < ntiles) issue(t + 2);
    float alpha[QT];
    const bool resc = decide(t == 0, cur, alpha);
    if (resc) {   // rare: P(t - 1) still refers to the old max, so its product goes in before O is rescaled
      if (t >= 1) pv(t - 1, pin);
      rescale_o(alpha);
      if (t + 1 < ntiles) qk(t + 1, nxt);
      softmax_pack(cur, pout);
    } else {      // the common trip: three independent streams for the scheduler to interleave
      if (t + 1 < ntiles) qk(t + 1, nxt);
      softmax_pack(cur, pout);
      if (t >= 1) pv(t - 1, pin);
    }
    if (t + 2 < ntiles) commit(t + 2);
    __syncthreads();
  };

  f32x4 sa[NKT][QT], sb[NKT][QT];
  f16x8 pa[QT][NS2], pb[QT][NS2];
#pragma unroll
  for (int j = 0; j < QT; ++j)
#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) pa[j][s2] = pb[j][s2] = zero8();
  issue(0);
  commit(0);
  issue(1);
  commit(1);
  __syncthreads();
  qk(0, sa);
  __syncthreads();   // every wave has read K(0) before trip 0 lets K(2) land in its stage
  int t = 0;
  for (; t + 1 < ntiles; t += 2) {
    trip(t, sa, sb, pa, pb);
    trip(t + 1, sb, sa, pb, pa);
  }
  if (t < ntiles) {   // odd tile count: the last tile's scores are in sa, P(t - 1) in pb
    trip(t, sa, sb, pa, pb);
    pv(ntiles - 1, pa);
  } else {
    pv(ntiles - 1, pb);
  }

  // ---- normalise and store: lane holds O[query l15][d = 16 i + 4 g + r]; the row sum sits in O^T row d
  f16* __restrict__ O = reinterpret_cast<f16*>(p.o) + (int64_t)bq * p.o_batch_stride + h * d;
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    float cand = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (i == (d >> 4) && r == (d & 3)) cand = o[i][j][r];
    const float lt = __shfl(cand, (((d & 15) >> 2) << 4) | l15, 64);
    const float inv = 1.0f / lt;
    const int row = q0 + j * 16 + l15;
    if (row >= lq) continue;
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

}  // namespace

// 0: not taken (the caller falls through to attn_kernel); > 0 launched; < 0 launch error.  I2V_ATTN_PIPE=1 selects it for the
// head_dim-40 problems whose keys are whole 64-key tiles (the 64 x 64 level's self and cross-frame attention).
int i2v_attention_pipe_try(const i2v_attn_params& p, hipStream_t s) {
  if (p.lse != nullptr) return 0;   // the log-sum-exp output exists in attn_kernel only
  static const int on = getenv("I2V_ATTN_PIPE") ? atoi(getenv("I2V_ATTN_PIPE")) : 0;
  if (!on) return 0;
  if (p.head_dim <= 32 || p.head_dim >= 48 || p.lk % 64 != 0 || p.lk < 192 || p.lq < 128) return 0;
  const float scale_log2 = p.scale * 1.4426950408889634f;
  const dim3 grid((unsigned)i2v_cdiv(p.lq, 128), (unsigned)p.heads, (unsigned)p.batch_q), block(256);
  hipLaunchKernelGGL(attn_pipe_kernel, grid, block, 0, s, p, scale_log2);
  const int rc = i2v_check_launch("i2v_attention_f16(pipelined)");
  return rc < 0 ? rc : 1;
}
