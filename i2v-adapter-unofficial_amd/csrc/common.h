// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of libi2v_hip.so.
// wave = 64 lanes; MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/i2v_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define I2V_WAVE 64

// error plumbing (thread-local message, never throws / aborts across the ABI)
void i2v_set_error(const char* fmt, ...);
#define I2V_FAIL(code, ...)     \
  do {                          \
    i2v_set_error(__VA_ARGS__); \
    return (code);              \
  } while (0)
#define I2V_CHECK_ARG(cond, ...) \
  do {                           \
    if (!(cond)) I2V_FAIL(I2V_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

int i2v_check_launch(const char* what);
// CUs of the current device after opting `func` in to `lds_bytes` of dynamic LDS there (cached per kernel and device), 0 when the
// device refuses (runtime.hip)
int i2v_big_lds_kernel_cus(const void* func, size_t lds_bytes);
// grid of a persistent 128-row-tile kernel: by default every workgroup (one per CU) walks ceil(ntiles / CUs) tiles;
// I2V_FUSED_TILES_PER_WG=n (tuning) gives every workgroup n tiles instead (1: one workgroup per tile, the hardware's workgroup
// turnover instead of the walk)
int i2v_persistent_grid(int ntiles, int cus);

static inline int64_t i2v_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// D(16x16, f32) += A(16x32, f16) * B(32x16, f16)
//   A fragment: lane l holds A[row = l & 15][k = 8 * (l >> 4) + j], j = 0..7
//   B fragment: lane l holds B[k = 8 * (l >> 4) + j][col = l & 15]
//   D fragment: lane l holds D[row = 4 * (l >> 4) + r][col = l & 15], r = 0..3
__device__ __forceinline__ f32x4 mfma16x16x32(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// erf GELU: 0.5 x (1 + erf(x / sqrt 2)), arranged as gelu(x) = max(x, 0) - f(|x|) with f(t) = t (1 - Phi(t)) (|x| and the
// negations are free source modifiers, no sign select).  Two forms of f:
//  * default since round 4: transcendental-free -- f(min(t, 4.5)) as a degree-10 polynomial (weighted
//    least squares iterated to near-minimax, tools/fit_gelu.py), max |error| 5.3e-5 over all x in fp32 Horner form (below the
//    fp16 rounding of any result above 0.06; f(t > 4.5) < 1.6e-5): 9 FMAs + mul + min + max + sub.  A degree low enough to be
//    much cheaper (<= 7) leaves > 5e-4.  Same-box A/B of the whole step: 51.49 -> 51.22 ms (profiles/r4_ab_runs.txt).
//    (r5) refitted with f(0) = 0 PINNED (p(t) = t q(t)): gelu(0) = 0 exactly -- the round-4 fit wrote -2.4e-5 into zero
//    and padded columns -- and the relative error is bounded for small |x| (< 9e-4 below 0.05).
//  * -DI2V_GELU_ERF: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): 0.5 erfc(t / sqrt 2) = 0.5 poly(1 / (1 + p t))
//    exp(-t^2 / 2) with the constants folded -- 11 plain VALU + v_rcp + v_exp per value (a transcendental issues in 7.5
//    cycles against 2 - 2.5 for an f32 FMA, tools/valu_rate.hip).
#ifndef I2V_GELU_ERF
__device__ __forceinline__ float gelu_erf(float x) {
  const float t = fminf(fabsf(x), 4.5f);
  float p = -1.259170971e-05f;
  p = fmaf(p, t, 2.979045903e-04f);
  p = fmaf(p, t, -2.898241399e-03f);
  p = fmaf(p, t, 1.448444588e-02f);
  p = fmaf(p, t, -3.614101523e-02f);
  p = fmaf(p, t, 2.524543465e-02f);
  p = fmaf(p, t, 5.604827519e-02f);
  p = fmaf(p, t, -1.281169221e-03f);
  p = fmaf(p, t, -3.966752556e-01f);
  p = fmaf(p, t, 4.995719716e-01f);
  p *= t;                                   // f(0) = 0 pinned: no constant term
  return fmaxf(x, 0.0f) - p;
}
#else
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
  const float y = x * 0.84932180028801904272f;            // sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-y^2)
  const float h = poly * __builtin_amdgcn_exp2f(-(y * y));
  return fmaf(-ax, h, fmaxf(x, 0.0f));
}
#endif
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ f16x8 ld_global_16B(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ f16x8 zero8() {
  f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return z;
}

// max over lanes l, l ^ 16 (resp. l ^ 32) on the VALU (no LDS round trip).  v_permlane16_swap exchanges the odd 16-lane rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of
// the second; with both registers holding x they end as (rows 0,0,2,2 / rows 1,1,3,3) resp. (lo, lo / hi, hi) of x, and the
// max of the two is the reduction.
// Written as inline assembly on purpose.  Through __builtin_amdgcn_permlane16_swap / 32_swap this compiler (ROCm 7.2 clang)
// hands back the FIRST result for both elements of the returned pair: the max that follows folds away, the swaps stay, and
// the "reduction" silently becomes the value of lane group 0.  Every attention kernel of rounds 1-2 therefore took its
// running max from 16 of a tile's 64 keys -- exact in exact arithmetic (any shift common to a row is), invisible to tests
// with unit-variance inputs, and wrong once logits spread enough for exp2(s - m) to leave fp16 (found in round 3:
// tools/attn_amp_check.py, tests/test_kernels_gpu.py::test_attention_large_logits).
// (s_nop 1: a VALU write of an operand needs two wait states before the swap reads it; the compiler cannot see into the asm.)
__device__ __forceinline__ float lane_xor16_max(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float lane_xor32_max(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

// sums over lanes l ^ 16 / l ^ 32 the same way (see lane_xor16_max for why these are inline assembly)
__device__ __forceinline__ float lane_xor16_sum(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float lane_xor32_sum(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// sum over the 8 lanes l ^ {0..7} on the VALU (DPP: two quad permutes and the mirror of a half row), not three LDS round trips
__device__ __forceinline__ float sum_lanes8(float x) {
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
  x += dpp(x, std::integral_constant<int, 0x141>{});      // row_half_mirror: lane i <-> 7 - i of each 8
  return x;
}

// Workgroup barrier for LDS traffic only: this wave's LDS reads / writes are complete (lgkmcnt), every wave has arrived.  Unlike
// __syncthreads() -- whose release fence makes the compiler wait vmcnt(0) first -- global loads, stores and LDS-DMA already issued
// STAY IN FLIGHT across it (r5: the fused feed-forward's end-of-tile __syncthreads() waited for the epilogue's stores and for the
// next tile's prefetch: 20k cycles of a 150k-cycle tile).  Data that arrives in LDS by DMA is published by an explicit
// `s_waitcnt vmcnt` in front of the barrier, where a barrier has to publish it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// XCD-aware block remap (guide T1, bijective form): blocks b and b+8 share an XCD/L2, so give each XCD a
// contiguous run of logical tile ids.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  if (nwg < nx * 2) return bid;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
