// Flash-style attention forward for gfx950: K1 cross-frame adapter attention (all frames of a clip attend to the
// frame-0 K/V), K2 spatial self-attention, K3 text / IP-Adapter cross-attention.
//
// One workgroup = 4 waves; each wave owns QT 16-row query tiles of one (batch, head); the 64-key K tile and the
// 64-key V^T tile are staged once per workgroup in LDS and shared by the 4 waves (for K1 the same frame-0 K/V
// tile is additionally shared through L2 by the workgroups of all frames of the clip: kv batch = q batch / group).
//
// MFMA orientation (v_mfma_f32_16x16x32_f16, D = A * B):
//   S^T tile [16 keys x 16 queries] = K[16 x d] * Q^T[d x 16]        A = K fragment (LDS), B = Q fragment (registers)
//   O^T tile [16 d    x 16 queries] = V^T[16 x 32 keys] * P^T[32 x 16] A = V^T fragment (LDS), B = P (registers)
// With S^T in the accumulator, lane (g = lane >> 4, c = lane & 15) holds scores of query c for 4 keys: the softmax
// row reduction is 15 in-register max/adds plus two wavefront shuffles (xor 16, 32), the per-query rescale of O^T is
// lane-local, and the accumulator of S^T *is* the B operand of the PV product (no LDS round trip, no lane movement):
// the row->key assignment of S^T tile `kt` is chosen as key = 32 (kt >> 1) + 8 g + 4 (kt & 1) + r so that the 8 values
// a lane holds for k-step s are keys 32 s + 8 g + 0..7, i.e. one contiguous 16-byte read of a V^T row.
// V arrives already transposed ([channel][key], written by the projection GEMM's I2V_STORE_VT epilogue).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int VS = KV_TILE + 8;  // V^T LDS row stride (halfs)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DQK, int DPV, int QT>
__global__ __launch_bounds__(256) void attn_kernel(const i2v_attn_params p, const float scale_log2) {
  constexpr int KS = DQK + 8;          // K LDS row stride (halfs)
  constexpr int KSTEPS = DQK / 32;     // k-steps of the QK^T product
  constexpr int DT = DPV / 16;         // d tiles of O^T
  constexpr int KCH = DQK / 8;         // 16-byte chunks per K row
  constexpr int NKC = (KV_TILE * KCH) / 256;            // K chunks per thread
  constexpr int NVC = (DPV * 8 + 255) / 256;            // V^T chunks per thread (last pass may be partial)
  __shared__ __attribute__((aligned(16))) f16 sK[KV_TILE * KS];
  __shared__ __attribute__((aligned(16))) f16 sV[DPV * VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int bq = blockIdx.z, h = blockIdx.y;
  const int bkv = bq / p.kv_group;
  const int d = p.head_dim, lq = p.lq, lk = p.lk;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);

  const f16* __restrict__ Q = reinterpret_cast<const f16*>(p.q) + (int64_t)bq * p.q_batch_stride + h * d;
  const f16* __restrict__ Kg = reinterpret_cast<const f16*>(p.k) + (int64_t)bkv * p.k_batch_stride + h * d;
  const f16* __restrict__ Vg =
      reinterpret_cast<const f16*>(p.vt) + (int64_t)bkv * p.vt_batch_stride + (int64_t)h * d * p.vt_row_stride;

  // ---- Q fragments (B operand of QK^T), resident in registers for the whole key loop
  f16x8 qf[QT][KSTEPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int row = q0 + qt * 16 + l15;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int dd = 32 * s + 8 * g;
      f16x8 v = zero8();
      if (row < lq && dd < d) v = ld_global_16B(Q + (int64_t)row * p.q_row_stride + dd);
      qf[qt][s] = v;
    }
  }

  f32x4 o[DT][QT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[QT], lrow[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    mrow[j] = -INFINITY;
    lrow[j] = 0.f;
  }

  f16x8 rk[NKC], rv[NVC];

  auto prefetch = [&](int t) {
    const int key_base = t * KV_TILE;
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + 256 * i;
      const int row = id / KCH, c = id - row * KCH;
      const int key = key_base + row;
      f16x8 v = zero8();
      if (key < lk && 8 * c < d) v = ld_global_16B(Kg + (int64_t)key * p.k_row_stride + 8 * c);
      rk[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + 256 * i;
      const int row = id >> 3, c = id & 7;
      const int key0 = key_base + 8 * c;
      f16x8 v = zero8();
      if (row < DPV && row < d && key0 < lk) {
        v = ld_global_16B(Vg + (int64_t)row * p.vt_row_stride + key0);
        if (key0 + 8 > lk) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (key0 + e >= lk) v[e] = (f16)0.f;
        }
      }
      rv[i] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + 256 * i;
      const int row = id / KCH, c = id - row * KCH;
      *reinterpret_cast<f16x8*>(&sK[row * KS + 8 * c]) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + 256 * i;
      const int row = id >> 3, c = id & 7;
      if (row < DPV) *reinterpret_cast<f16x8*>(&sV[row * VS + 8 * c]) = rv[i];
    }
  };

  const int ntiles = (lk + KV_TILE - 1) / KV_TILE;
  prefetch(0);
  commit();
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) prefetch(t + 1);

    // ---- S^T = K Q^T
    f32x4 sacc[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < QT; ++j) sacc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int krow = 32 * (kt >> 1) + 8 * (l15 >> 2) + 4 * (kt & 1) + (l15 & 3);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[krow * KS + 32 * s + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) sacc[kt][j] = mfma16x16x32(kf, qf[j][s], sacc[kt][j]);
      }
    }

    // ---- online softmax (per query column; keys spread over registers and the 4 lane groups)
    const int key_base = t * KV_TILE;
    const bool tail = key_base + KV_TILE > lk;
    f16x8 pf[QT][2];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      float sv[4][4];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sacc[kt][j][r] * scale_log2;
          if (tail) {
            const int key = key_base + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r;
            if (key >= lk) v = -INFINITY;
          }
          sv[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrow[j], mx);
      const float alpha = fast_exp2(mrow[j] - mnew);
      mrow[j] = mnew;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(sv[kt][r] - mnew);
          sv[kt][r] = e;
          ls += e;
        }
      lrow[j] = lrow[j] * alpha + ls;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[i][j][r] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pk[r] = (f16)sv[2 * s2][r];
          pk[4 + r] = (f16)sv[2 * s2 + 1][r];
        }
        pf[j][s2] = pk;
      }
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&sV[(i * 16 + l15) * VS + 32 * s2 + 8 * g]);
#pragma unroll
        for (int j = 0; j < QT; ++j) o[i][j] = mfma16x16x32(vf, pf[j][s2], o[i][j]);
      }

    __syncthreads();
    if (more) commit();
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[query l15][d = 16 i + 4 g + r]
  f16* __restrict__ O = reinterpret_cast<f16*>(p.o) + (int64_t)bq * p.o_batch_stride + h * d;
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    float lt = lrow[j];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
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

template <int DQK, int DPV>
int launch_d(const i2v_attn_params& p, hipStream_t s) {
  const float scale_log2 = p.scale * 1.4426950408889634f;
  // measured (profiles/r1_tile_sweep.txt): 2 query tiles per wave keep 2 waves / SIMD resident, so one wave's
  // softmax VALU overlaps the other's MFMAs; 4 tiles drop to 1 wave / SIMD and serialise the two pipes.
  int qt = 1;
  if (p.lq >= 128 && DQK <= 96) qt = 2;
  static const int qt_env = getenv("I2V_ATTN_QT") ? atoi(getenv("I2V_ATTN_QT")) : 0;  // tuning override
  if (qt_env == 1 || qt_env == 2 || (qt_env == 4 && DQK <= 96)) qt = qt_env;
  const dim3 block(256);
  const dim3 grid((unsigned)i2v_cdiv(p.lq, 64 * qt), (unsigned)p.heads, (unsigned)p.batch_q);
  if (qt == 4) {
    if constexpr (DQK <= 96) hipLaunchKernelGGL((attn_kernel<DQK, DPV, 4>), grid, block, 0, s, p, scale_log2);
  } else if (qt == 2) {
    hipLaunchKernelGGL((attn_kernel<DQK, DPV, 2>), grid, block, 0, s, p, scale_log2);
  } else {
    hipLaunchKernelGGL((attn_kernel<DQK, DPV, 1>), grid, block, 0, s, p, scale_log2);
  }
  return i2v_check_launch("i2v_attention_f16");
}

inline bool al(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" int i2v_attention_f16(const i2v_attn_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_attention_f16: null params");
  const i2v_attn_params& p = *pp;
  I2V_CHECK_ARG(p.q && p.k && p.vt && p.o, "i2v_attention_f16: null pointer");
  I2V_CHECK_ARG(p.batch_q > 0 && p.kv_group > 0 && p.batch_q % p.kv_group == 0,
                "i2v_attention_f16: batch_q (%d) must be a positive multiple of kv_group (%d)", p.batch_q, p.kv_group);
  I2V_CHECK_ARG(p.heads > 0 && p.lq > 0 && p.lk > 0, "i2v_attention_f16: heads, lq, lk must be positive");
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
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int d = p.head_dim;
  if (d <= 16) return launch_d<32, 16>(p, s);
  if (d <= 32) return launch_d<32, 32>(p, s);
  if (d <= 48) return launch_d<64, 48>(p, s);
  if (d <= 64) return launch_d<64, 64>(p, s);
  if (d <= 80) return launch_d<96, 80>(p, s);
  if (d <= 96) return launch_d<96, 96>(p, s);
  if (d <= 128) return launch_d<128, 128>(p, s);
  return launch_d<160, 160>(p, s);
}
