// Temporal (motion-module) self-attention for gfx950: sequence = the <= 32 frames of ONE pixel, batch = B * H * W.
//
// Inside a motion module the token order is (b, pixel, frame) (the GroupNorm entry kernel writes it, the proj_out
// GEMM epilogue undoes it), so a pixel's q / k rows are `frames` consecutive rows and its V^T block
// [channel][frame] is contiguous.  Every q / k / v element is used by exactly one (pixel, head) problem, so
// there is nothing to share: fragments go straight from global memory to registers (no LDS) and the kernel is
// bound by HBM traffic (read q, k, v^T once, write o once).
// One wave computes one (pixel, head): S^T[32 keys x 16 NQT queries] with 16x16x32 MFMAs over the head dim,
// softmax over the keys with two wavefront shuffles, O^T = V^T P^T with one 16x16x32 MFMA per 16 channels
// (the S^T accumulator is the B operand; key rows are permuted as in attention.hip so that a lane's 8 keys are
// one contiguous 16-byte run of a V^T row).
#include <cstdlib>

#include "common.h"

namespace {

template <int DQK, int DPV, int NQT>
__global__ __launch_bounds__(256) void tattn_kernel(const i2v_tattn_params p, const float scale_log2, const int n_items) {
  constexpr int KSTEPS = DQK / 32, DT = DPV / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int F = p.frames, d = p.head_dim, C = p.heads * p.head_dim;
  const f16* __restrict__ Qb = reinterpret_cast<const f16*>(p.q);
  const f16* __restrict__ Kb = reinterpret_cast<const f16*>(p.k);
  const f16* __restrict__ Vb = reinterpret_cast<const f16*>(p.vt);
  f16* __restrict__ Ob = reinterpret_cast<f16*>(p.o);

  for (int item = blockIdx.x * 4 + wave; item < n_items; item += gridDim.x * 4) {
    const int pix = item / p.heads, h = item - pix * p.heads;
    const f16* Q = Qb + (int64_t)pix * F * p.q_row_stride + h * d;
    const f16* K = Kb + (int64_t)pix * F * p.k_row_stride + h * d;
    const f16* Vt = Vb + ((int64_t)pix * C + h * d) * p.vt_ld;

    // V^T does not depend on the softmax: its loads are issued with the Q / K loads, one memory round trip per item
    f16x8 vfr[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int ch = i * 16 + l15;
      vfr[i] = (ch < d && 8 * g < F) ? ld_global_16B(Vt + (int64_t)ch * p.vt_ld + 8 * g) : zero8();
    }

    f32x4 sacc[2][NQT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < NQT; ++j) sacc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int dd = 32 * s + 8 * g;
      f16x8 qf[NQT], kf[2];
#pragma unroll
      for (int j = 0; j < NQT; ++j) {
        const int f = j * 16 + l15;
        qf[j] = (f < F && dd < d) ? ld_global_16B(Q + (int64_t)f * p.q_row_stride + dd) : zero8();
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int f = 8 * (l15 >> 2) + 4 * kt + (l15 & 3);
        kf[kt] = (f < F && dd < d) ? ld_global_16B(K + (int64_t)f * p.k_row_stride + dd) : zero8();
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int j = 0; j < NQT; ++j) sacc[kt][j] = mfma16x16x32(kf[kt], qf[j], sacc[kt][j]);
    }

    f16x8 pf[NQT];
#pragma unroll
    for (int j = 0; j < NQT; ++j) {
      float sv[2][4];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 8 * g + 4 * kt + r;
          const float v = key < F ? sacc[kt][j][r] * scale_log2 : -INFINITY;
          sv[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[kt][r] = __builtin_amdgcn_exp2f(sv[kt][r] - mx);
          ls += sv[kt][r];
        }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      const float inv = 1.0f / ls;
      f16x8 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk[r] = (f16)(sv[0][r] * inv);
        pk[4 + r] = (f16)(sv[1][r] * inv);
      }
      pf[j] = pk;
    }

#pragma unroll
    for (int i = 0; i < DT; ++i) {
      f16x8 vf = vfr[i];
      if (8 * g + 8 > F) {   // frames past F inside the last 16-byte run: padding, not data
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * g + e >= F) vf[e] = (f16)0.f;
      }
#pragma unroll
      for (int j = 0; j < NQT; ++j) {
        const f32x4 oacc = mfma16x16x32(vf, pf[j], f32x4{0.f, 0.f, 0.f, 0.f});
        const int f = j * 16 + l15, dd = i * 16 + 4 * g;
        if (f < F && dd < d) {
          f16x4 ov = {(f16)oacc[0], (f16)oacc[1], (f16)oacc[2], (f16)oacc[3]};
          *reinterpret_cast<f16x4*>(Ob + ((int64_t)pix * F + f) * p.o_row_stride + h * d + dd) = ov;
        }
      }
    }
  }
}

// Narrow-channel variant (frames <= 16 and a pixel's q / k / o rows + V^T block within 64 KiB of LDS: C <= 320, i.e. the
// 64 x 64 level; at C = 640 the block is 93 KB = one workgroup per CU, and the register kernel above already streams
// that level at 4.3-4.6 TB/s, so it stays there).  With head_dim 40 a head's slice of
// a token row is 80 bytes, so the per-wave fragment loads above touch 16 rows x 64-80 bytes per instruction and the
// kernel ran at 3.4 TB/s.  Here one workgroup owns ONE pixel with all its heads: the pixel's q / k rows ([F][C],
// whole rows) and V^T block ([C][vt_ld], contiguous) are copied to LDS with 16 bytes per lane over whole rows, the
// waves take their heads' fragments from LDS (row strides padded so the 16 rows of a fragment read start on distinct
// bank groups), and O goes back through LDS for whole-row 16-byte stores.
template <int DQK, int DPV, int PF>
__global__ __launch_bounds__(256) void tattn_lds_kernel(const i2v_tattn_params p, const float scale_log2) {
  constexpr int KSTEPS = DQK / 32, DT = DPV / 16;
  extern __shared__ __attribute__((aligned(16))) f16 tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int F = p.frames, d = p.head_dim, C = p.heads * p.head_dim;
  const int RS = C + 8, VS = p.vt_ld + 8;          // padded LDS row strides (halfs)
  f16* sq = tsm;
  f16* sk = sq + 16 * RS;
  f16* so = sk + 16 * RS;
  f16* sv = so + 16 * RS;
  const int cpr = C / 8, vpr = p.vt_ld / 8;         // 16-byte chunks per row
  const f16* __restrict__ Qb = reinterpret_cast<const f16*>(p.q);
  const f16* __restrict__ Kb = reinterpret_cast<const f16*>(p.k);
  const f16* __restrict__ Vb = reinterpret_cast<const f16*>(p.vt);
  f16* __restrict__ Ob = reinterpret_cast<f16*>(p.o);

  // The NEXT pixel's q / k / V^T chunks are fetched into registers while this pixel is computed and stored (one pixel's
  // phases -- load, compute, store -- used to run in series per workgroup: 3.4 TB/s); PF = chunks per thread and operand,
  // sized by the host for this shape (F * C / 8 and C * vt_ld / 8 over 256 threads).
  f16x8 rq[PF], rk[PF], rv[PF];
  auto fetch = [&](int pix) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int t = tid + 256 * i;
      rq[i] = rk[i] = rv[i] = zero8();
      if (t < F * cpr) {
        const int f = t / cpr, c = t - f * cpr;
        rq[i] = ld_global_16B(Qb + ((int64_t)pix * F + f) * p.q_row_stride + c * 8);
        rk[i] = ld_global_16B(Kb + ((int64_t)pix * F + f) * p.k_row_stride + c * 8);
      }
      if (t < C * vpr) rv[i] = ld_global_16B(Vb + ((int64_t)pix * C + t / vpr) * p.vt_ld + (t % vpr) * 8);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int t = tid + 256 * i;
      if (t < F * cpr) {
        const int f = t / cpr, c = t - f * cpr;
        *reinterpret_cast<f16x8*>(sq + f * RS + c * 8) = rq[i];
        *reinterpret_cast<f16x8*>(sk + f * RS + c * 8) = rk[i];
      }
      if (t < C * vpr) *reinterpret_cast<f16x8*>(sv + (t / vpr) * VS + (t % vpr) * 8) = rv[i];
    }
  };
  if ((int)blockIdx.x < p.n_pixels) {
    fetch(blockIdx.x);
    commit();
  }
  __syncthreads();
  for (int pix = blockIdx.x; pix < p.n_pixels; pix += gridDim.x) {
    const int next = pix + (int)gridDim.x;
    if (next < p.n_pixels) fetch(next);

    for (int h = wave; h < p.heads; h += 4) {
      f32x4 sacc[2];
      sacc[0] = sacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const int dd = 32 * s + 8 * g;
        const bool in = dd < d;
        const f16x8 qf = (l15 < F && in) ? *reinterpret_cast<const f16x8*>(sq + l15 * RS + h * d + dd) : zero8();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int f = 8 * (l15 >> 2) + 4 * kt + (l15 & 3);
          const f16x8 kf = (f < F && in) ? *reinterpret_cast<const f16x8*>(sk + f * RS + h * d + dd) : zero8();
          sacc[kt] = mfma16x16x32(kf, qf, sacc[kt]);
        }
      }
      float sv8[2][4];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 8 * g + 4 * kt + r;
          const float v = key < F ? sacc[kt][r] * scale_log2 : -INFINITY;
          sv8[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv8[kt][r] = __builtin_amdgcn_exp2f(sv8[kt][r] - mx);
          ls += sv8[kt][r];
        }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      const float inv = 1.0f / ls;
      f16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (f16)(sv8[0][r] * inv);
        pf[4 + r] = (f16)(sv8[1][r] * inv);
      }
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const int ch = i * 16 + l15;
        f16x8 vf = (ch < d && 8 * g < F) ? *reinterpret_cast<const f16x8*>(sv + (h * d + ch) * VS + 8 * g) : zero8();
        if (8 * g + 8 > F) {   // frames past F inside the last 16-byte run: padding, not data
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (8 * g + e >= F) vf[e] = (f16)0.f;
        }
        const f32x4 oacc = mfma16x16x32(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f});
        const int dd = i * 16 + 4 * g;
        if (l15 < F && dd < d) {
          const f16x4 ov = {(f16)oacc[0], (f16)oacc[1], (f16)oacc[2], (f16)oacc[3]};
          *reinterpret_cast<f16x4*>(so + l15 * RS + h * d + dd) = ov;
        }
      }
    }
    __syncthreads();      // every wave has read this pixel's q / k / V^T and written its part of `so`
    for (int t = tid; t < F * cpr; t += 256) {
      const int f = t / cpr, c = t - f * cpr;
      *reinterpret_cast<f16x8*>(Ob + ((int64_t)pix * F + f) * p.o_row_stride + c * 8) =
          *reinterpret_cast<const f16x8*>(so + f * RS + c * 8);
    }
    if (next < p.n_pixels) commit();
    __syncthreads();      // the next pixel's operands are in LDS; `so` has been read
  }
}

template <int DQK, int DPV>
int launch_t(const i2v_tattn_params& p, hipStream_t s) {
  const float scale_log2 = p.scale * 1.4426950408889634f;
  const int C = p.heads * p.head_dim;
  static const int lds_off = getenv("I2V_TATTN_LDS") ? (atoi(getenv("I2V_TATTN_LDS")) == 0) : 0;
  // 64 KiB = the dynamic-LDS size a kernel may request without hipFuncSetAttribute, and two workgroups per CU
  const size_t lds = (size_t)(3 * 16 * (C + 8) + C * (p.vt_ld + 8)) * sizeof(f16);
  if (!lds_off && p.frames <= 16 && lds <= 64 * 1024 && p.o_row_stride % 8 == 0 &&
      (reinterpret_cast<uintptr_t>(p.o) & 15) == 0) {
    int64_t blocks = p.n_pixels < 256 * 8 ? p.n_pixels : 256 * 8;
    const int chunks = (C / 8) * (p.frames > p.vt_ld ? p.frames : p.vt_ld);   // max(F * C / 8, C * vt_ld / 8)
    if (chunks <= 256)
      hipLaunchKernelGGL((tattn_lds_kernel<DQK, DPV, 1>), dim3((unsigned)blocks), dim3(256), lds, s, p, scale_log2);
    else if (chunks <= 512)
      hipLaunchKernelGGL((tattn_lds_kernel<DQK, DPV, 2>), dim3((unsigned)blocks), dim3(256), lds, s, p, scale_log2);
    else if (chunks <= 768)
      hipLaunchKernelGGL((tattn_lds_kernel<DQK, DPV, 3>), dim3((unsigned)blocks), dim3(256), lds, s, p, scale_log2);
    else
      hipLaunchKernelGGL((tattn_lds_kernel<DQK, DPV, 4>), dim3((unsigned)blocks), dim3(256), lds, s, p, scale_log2);
    return i2v_check_launch("i2v_temporal_attention_f16");
  }
  const int64_t items64 = (int64_t)p.n_pixels * p.heads;
  const int n_items = (int)items64;
  int64_t blocks = i2v_cdiv(items64, 4);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (p.frames <= 16)
    hipLaunchKernelGGL((tattn_kernel<DQK, DPV, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p, scale_log2, n_items);
  else
    hipLaunchKernelGGL((tattn_kernel<DQK, DPV, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p, scale_log2, n_items);
  return i2v_check_launch("i2v_temporal_attention_f16");
}

inline bool al(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" int i2v_temporal_attention_f16(const i2v_tattn_params* pp, i2v_stream_t stream) {
  I2V_CHECK_ARG(pp != nullptr, "i2v_temporal_attention_f16: null params");
  const i2v_tattn_params& p = *pp;
  I2V_CHECK_ARG(p.q && p.k && p.vt && p.o, "i2v_temporal_attention_f16: null pointer");
  I2V_CHECK_ARG(p.n_pixels > 0 && p.heads > 0, "i2v_temporal_attention_f16: n_pixels / heads must be positive");
  I2V_CHECK_ARG((int64_t)p.n_pixels * p.heads < (1ll << 31), "i2v_temporal_attention_f16: too many items");
  I2V_CHECK_ARG(p.frames >= 1 && p.frames <= 32, "i2v_temporal_attention_f16: frames (%d) must be in [1, 32]", p.frames);
  I2V_CHECK_ARG(p.head_dim > 0 && p.head_dim % 8 == 0 && p.head_dim <= 160,
                "i2v_temporal_attention_f16: head_dim (%d) must be a multiple of 8 and <= 160", p.head_dim);
  I2V_CHECK_ARG(p.q_row_stride % 8 == 0 && p.k_row_stride % 8 == 0 && p.o_row_stride % 4 == 0,
                "i2v_temporal_attention_f16: row strides must be multiples of 8 (q, k) / 4 (o)");
  I2V_CHECK_ARG(p.vt_ld % 8 == 0 && p.vt_ld >= ((p.frames + 7) / 8) * 8,
                "i2v_temporal_attention_f16: vt_ld must be a multiple of 8 and >= frames rounded up to 8");
  I2V_CHECK_ARG(al(p.q, 16) && al(p.k, 16) && al(p.vt, 16) && al(p.o, 8), "i2v_temporal_attention_f16: pointer alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int d = p.head_dim;
  if (d <= 16) return launch_t<32, 16>(p, s);
  if (d <= 32) return launch_t<32, 32>(p, s);
  if (d <= 48) return launch_t<64, 48>(p, s);
  if (d <= 64) return launch_t<64, 64>(p, s);
  if (d <= 80) return launch_t<96, 80>(p, s);
  if (d <= 96) return launch_t<96, 96>(p, s);
  if (d <= 128) return launch_t<128, 128>(p, s);
  return launch_t<160, 160>(p, s);
}
