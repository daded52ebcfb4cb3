// Flash attention forward for head_dim 40 on v_mfma_f32_32x32x16_f16 (the 64 x 64 level of the UNet: 80 % of the
// attention time).  Same contract as attn_kernel (attention.hip): K1 / K2 / K3 forms, V passed transposed.
//
// Why a second formulation.  At head_dim 40 the 16x16x32 kernel is bound by the SIMD's vector issue port, not by the
// matrix pipe: per 16-query x 64-key tile it issues 14 MFMAs (each holds the port for 8 of its 16 cycles), 16 v_exp_f32
// (8 cycles each) and ~19 other VALU (4 each) = 316 issue cycles against 224 MFMA cycles.  A 32x32x16 MFMA holds the port
// for the same 8 cycles but runs 32, and contracts 16 deep, so
//   S^T [64 keys x 32 queries] = K Q^T : 2 key blocks x 3 k-steps (d padded 40 -> 48, not 64)      =  6 MFMAs
//   O^T [64 d    x 32 queries] = V^T P^T: 2 d blocks   x 4 k-steps (d padded to 64)                 =  8 MFMAs
// = 14 MFMAs per 32 queries instead of 28: the same 448 matrix-pipe cycles, half the MFMA issue cycles.
//
// Layouts (guide: D col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); A[row = lane & 31][k = 8 h + j],
// B[k = 8 h + j][col = lane & 31], h = lane >> 5): a lane owns ONE query (both halves h own the same 32 queries) and
// 16 keys of each 32-key block: keys (reg & 3) + 8 (reg >> 2) + 4 h.  The row max is an in-register chain over 32
// values + one v_permlane32_swap.  The S^T accumulator is the B operand of the PV product as it stands: registers
// 8 s2 .. 8 s2 + 7 of key block kb are k-slots 0..7 of k-step (kb, s2), i.e. keys 32 kb + 16 s2 + {0..3, 8..11} + 4 h;
// the V^T tile is stored in LDS with its keys in exactly that order, so the A fragment is one 16-byte read.
// V^T row `head_dim` is all ones: O^T[head_dim][q] is the row sum of the fp16-rounded P that multiplies V.
#include <cstdlib>
#include <type_traits>

#include "../common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float DEFER_THR32 = 8.0f;   // log2 units: O / l are rescaled only when a tile's max exceeds m by more

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float xor32_maxf(float x) { return lane_xor32_max(x); }   // common.h
__device__ __forceinline__ uint32_t pack_rtz2(float a, float b) {
  const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int A32_DQK = 48;            // contraction depth of QK^T (3 k-steps of 16)
constexpr int A32_KVT = 64;            // keys per tile
constexpr int A32_KS = A32_DQK + 8;    // K LDS row stride (halfs): 112 B
constexpr int A32_VS = A32_KVT + 8;    // V^T LDS row stride (halfs): 144 B

// (held to 128 VGPRs = 4 waves / SIMD: 16 spilled registers, 944 us against 964 us at 147 VGPRs / 3 waves)
#ifndef I2V_A32_WAVES
#define I2V_A32_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(I2V_A32_WAVES))) void attn32_kernel(const i2v_attn_params p, const float scale_log2) {
  __shared__ __attribute__((aligned(16))) f16 sKb[2][A32_KVT * A32_KS];
  __shared__ __attribute__((aligned(16))) f16 sVb[2][64 * A32_VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r31 = lane & 31, hh = lane >> 5;
  // XCD-aware order (see attention.hip): the query blocks of one (batch, head) stay on one XCD's L2
  int qb = blockIdx.x, h = blockIdx.y, bq = blockIdx.z;
  {
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
  const int q0 = qb * 128 + wave * 32;

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(p.q) + (int64_t)bq * p.q_batch_stride + h * d;
  const f16* __restrict__ Kg = reinterpret_cast<const f16*>(p.k) + (int64_t)bkv * p.k_batch_stride + h * d;
  const f16* __restrict__ Vg =
      reinterpret_cast<const f16*>(p.vt) + (int64_t)bkv * p.vt_batch_stride + (int64_t)h * d * p.vt_row_stride;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (query r31, half hh) holds Q[q][16 s + 8 hh .. + 7], pre-scaled
  f16x8 qf[3];
  {
    const int row = q0 + r31;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int dd = 16 * s + 8 * hh;
      f16x8 v = zero8();
      if (row < lq && dd < d) v = ld_global_16B(Q + (int64_t)row * p.q_row_stride + dd);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] * scale_log2);
      qf[s] = v;
    }
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  // -(running max) enters through the contraction itself: head_dim 40 leaves slots 40..47 of the 48-deep QK^T free, so
  // K carries 1.0 in slot 40 (every key) and the query's B fragment -m there: the MFMA chain ends with s - m at no cost,
  // without a 16-register initial accumulator.  m is kept as the fp16 VALUE that sits in the fragment (softmax is
  // invariant to the constant subtracted per query; what matters is that O's rescale uses the same value).
  float negm = 0.f;   // always exactly representable in fp16
  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;

  // ---- K / V^T staging through registers (raw buffer loads: per-lane offset once + scalar tile offset; lanes outside
  //      head_dim / past the tile carry an out-of-range offset and receive zeros)
  constexpr int OOB = 0x40000000;
  constexpr int KCH = A32_DQK / 8;     // 6 chunks per K row (the 6th is zero for head_dim 40)
  f16x8 rk[2], rv[2];
  int k_off[2], k_lds[2], v_off[2], v_lds[2], v_lds2[2];
  bool k_one[2];
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(Kg), 0, (int)(((int64_t)(lk - 1) * p.k_row_stride + d) * 2), 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(Vg), 0, (int)(((int64_t)(d - 1) * p.vt_row_stride + ((lk + 7) & ~7)) * 2), 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int row = id / KCH, c = id - row * KCH;
    const bool in = id < A32_KVT * KCH;
    k_off[i] = (in && 8 * c < d) ? (int)((row * p.k_row_stride + 8 * c) * 2) : OOB;
    k_one[i] = in && c == 5;   // chunk 5 = slots 40..47: slot 40 carries the 1.0 that multiplies -m
    k_lds[i] = in ? row * A32_KS + 8 * c : (tid & 63) * A32_KS + A32_DQK;   // parked in a row's 16-byte pad
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    const int row = id >> 3, c = id & 7;   // V^T row (channel), 8-key chunk of the tile
    const bool in = row < d;
    v_off[i] = in ? (int)((row * p.vt_row_stride + 8 * c) * 2) : OOB;
    // keys 8 c + 0..3 -> slot ((kb 2 + s2) 2 + 0), keys 8 c + 4..7 -> the next slot; position run * 4 inside the slot
    const int kb = c >> 2, s2 = (c >> 1) & 1, run = c & 1;
    v_lds[i] = in ? row * A32_VS + ((kb * 2 + s2) * 2) * 8 + run * 4 : (tid & 63) * A32_VS + A32_KVT;
    v_lds2[i] = in ? v_lds[i] + 8 : v_lds[i] + 4;   // lanes with nothing to store park both halves in a row's 16-byte pad
  }
  // rows head_dim + 1 .. 63 of V^T stay zero, row head_dim is all ones (the row sum); written once per stage
  for (int i = tid; i < 2 * 64 * (A32_VS / 8); i += 256) {
    const int st = i / (64 * (A32_VS / 8)), rem = i - st * (64 * (A32_VS / 8));
    const int row = rem / (A32_VS / 8), c = rem - row * (A32_VS / 8);
    if (row >= d) {
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)((row == d && c < 8) ? 1.f : 0.f);
      *reinterpret_cast<f16x8*>(&sVb[st][row * A32_VS + 8 * c]) = v;
    }
  }
  const int k_tile_bytes = (int)(A32_KVT * p.k_row_stride * 2);

  auto issue = [&](int t) {
    const int ks = t * k_tile_bytes, vs = t * (A32_KVT * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rk[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_off[i], ks, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rv[i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, v_off[i], vs, 0));
  };
  auto mask_tail_v = [&](int t) {   // V^T entries of keys >= lk run into row padding / the next row: zero them
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key0 = t * A32_KVT + 8 * ((tid + 256 * i) & 7);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (key0 + e >= lk) rv[i][e] = (f16)0.f;
    }
  };
  auto commit = [&](int stage) {
    f16* sK = sKb[stage];
    f16* sV = sVb[stage];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x8 kv = rk[i];
      if (k_one[i]) kv[0] = (f16)1.f;
      *reinterpret_cast<f16x8*>(&sK[k_lds[i]]) = kv;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f16x4 lo = {rv[i][0], rv[i][1], rv[i][2], rv[i][3]};
      const f16x4 hi = {rv[i][4], rv[i][5], rv[i][6], rv[i][7]};
      *reinterpret_cast<f16x4*>(&sV[v_lds[i]]) = lo;
      *reinterpret_cast<f16x4*>(&sV[v_lds2[i]]) = hi;
    }
  };

  const int ntiles = (lk + A32_KVT - 1) / A32_KVT;
  const bool partial = (lk % A32_KVT) != 0;
  issue(0);
  if (ntiles == 1 && partial) mask_tail_v(0);
  __syncthreads();   // the ones / zero rows above
  commit(0);
  __syncthreads();

  auto process = [&](auto tail_c, auto more_c, const int t) {
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool more = decltype(more_c)::value;
    if (more) issue(t + 1);
    const f16* sK = sKb[t & 1];
    const f16* sV = sVb[t & 1];

    // ---- S^T = K Q^T - m
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const f16* kr = sK + (kb * 32 + r31) * A32_KS + 8 * hh;
      sacc[kb] = mfma32(*reinterpret_cast<const f16x8*>(kr), qf[0], zero16);
      sacc[kb] = mfma32(*reinterpret_cast<const f16x8*>(kr + 16), qf[1], sacc[kb]);
      sacc[kb] = mfma32(*reinterpret_cast<const f16x8*>(kr + 32), qf[2], sacc[kb]);
    }
    if (TAIL) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int key = t * A32_KVT + 32 * kb + (v & 3) + 8 * (v >> 2) + 4 * hh;
          if (key >= lk) sacc[kb][v] = -INFINITY;
        }
    }
    // ---- online softmax with deferred max: the lane's 32 values belong to ONE query
    float mx = max3f(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
    for (int v = 3; v < 15; v += 2) mx = max3f(mx, sacc[0][v], sacc[0][v + 1]);
    mx = fmaxf(mx, sacc[0][15]);
#pragma unroll
    for (int v = 0; v < 16; v += 2) mx = max3f(mx, sacc[1][v], sacc[1][v + 1]);
    const bool first = t == 0;
    if (__any(first || mx > DEFER_THR32)) {   // wave-uniform; rare after the first tile (the lane's own 32 keys decide)
      mx = xor32_maxf(mx);                    // the row max completed over the two halves only here
      const float want = first ? mx : fmaxf(mx, 0.f);
      const float negm_new = (float)(f16)(negm - want);   // the value the fragment will hold
      const float dlt = negm - negm_new;                  // what was really subtracted
      const float alpha = first ? 1.0f : exp2_fast(-dlt);
      negm = negm_new;
      if (hh == 1) qf[2][0] = (f16)negm;                  // slot 40 = element 0 of the upper half's third fragment
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[kb][e] -= dlt;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    // P = exp2(s - m), packed (round toward zero: the bias cancels in the ratio with the MFMA-made row sum)
    f16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[j] = pack_rtz2(exp2_fast(sacc[kb][8 * s2 + 2 * j]), exp2_fast(sacc[kb][8 * s2 + 2 * j + 1]));
        pf[kb][s2] = __builtin_bit_cast(f16x8, w);
      }
    // ---- O^T += V^T P^T (V^T row head_dim is all ones: that row of O^T is the row sum)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const f16x8 vf =
              *reinterpret_cast<const f16x8*>(&sV[(db * 32 + r31) * A32_VS + (((kb * 2 + s2) * 2) + hh) * 8]);
          oacc[db] = mfma32(vf, pf[kb][s2], oacc[db]);
        }
    if (more) {
      if (partial && t + 2 == ntiles) mask_tail_v(t + 1);
      commit((t + 1) & 1);
      __syncthreads();
    }
  };

  for (int t = 0; t + 1 < ntiles; ++t) process(std::false_type{}, std::true_type{}, t);
  if (partial)
    process(std::true_type{}, std::false_type{}, ntiles - 1);
  else
    process(std::false_type{}, std::false_type{}, ntiles - 1);

  // ---- normalise and store.  Lane (query r31, half hh) holds O[q][d = 32 db + 8 (v >> 2) + 4 hh + (v & 3)]; the row
  //      sum is O^T row head_dim: block db_l, register v_l, half h_l
  const int db_l = d >> 5, w_l = d & 31;
  const int v_l = ((w_l >> 3) << 2) | (w_l & 3), h_l = (w_l >> 2) & 1;
  float lsum = 0.f;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int v = 0; v < 16; ++v)
      if (db == db_l && v == v_l) lsum = oacc[db][v];
  lsum = __shfl(lsum, r31 + 32 * h_l, 64);
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  const int row = q0 + r31;
  if (row < lq) {
    f16* __restrict__ O = reinterpret_cast<f16*>(p.o) + (int64_t)bq * p.o_batch_stride + h * d + (int64_t)row * p.o_row_stride;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dd = 32 * db + 8 * g4 + 4 * hh;
        if (dd >= d) continue;
        f16x4 ov;
        if (p.accumulate) {
          const f16x4 prev = *reinterpret_cast<const f16x4*>(O + dd);
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (f16)((float)prev[r] + p.acc_scale * oacc[db][4 * g4 + r] * inv);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (f16)(oacc[db][4 * g4 + r] * inv);
        }
        *reinterpret_cast<f16x4*>(O + dd) = ov;
      }
  }
}

}  // namespace

// head_dim 40 with enough queries to fill 128-query workgroups.  OPT-IN (I2V_ATTN32=1): measured on the 64 x 64 level's
// self-attention (32 x 8 heads x 4096 x 4096, same box) this formulation runs 944-964 us against 954 us of the 16x16x32
// kernel -- the halved MFMA issue share buys nothing, so the issue port is not what bounds the kernel (the softmax of a
// tile cannot start before its S^T chain ends nor the PV chain before the softmax: per wave the three phases are serial,
// and three or four waves per SIMD do not cover each other completely).  Kept as a tested alternative.
int i2v_attention32_try(const i2v_attn_params& p, hipStream_t s) {
  if (p.lse != nullptr) return 0;   // the log-sum-exp output exists in attn_kernel only
  static const int on = getenv("I2V_ATTN32") ? atoi(getenv("I2V_ATTN32")) : 0;
  if (!on || p.head_dim != 40 || p.lq < 128) return 0;   // head_dim 40: slot 40 of the 48-deep contraction is free
  const float scale_log2 = p.scale * 1.4426950408889634f;
  const dim3 grid((unsigned)i2v_cdiv(p.lq, 128), (unsigned)p.heads, (unsigned)p.batch_q), block(256);
  hipLaunchKernelGGL(attn32_kernel, grid, block, 0, s, p, scale_log2);
  const int rc = i2v_check_launch("i2v_attention_f16(32x32)");
  return rc < 0 ? rc : 1;
}
