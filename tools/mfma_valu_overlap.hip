// Microbenchmark (r5): do a wave's MFMAs and its own independent VALU work overlap on gfx950, and do those of the two waves of a SIMD?
// Per iteration: 4 independent v_mfma_f32_16x16x32_f16 and NV independent v_fma_f32 (or v_exp_f32), interleaved in program order.
// Whole chip (256 CUs), 1 or 2 waves per SIMD, wall time by events.  Prints ns per iteration and wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, bool EXP>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  h8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {1, 1, 1, 1, 1, 1, 1, 1};
  f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.001f * (threadIdx.x + j);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < NM) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 4; ++j) {
        const int q = (m * (NV / 4) + j) & 7;
        if (EXP) v[q] = __builtin_amdgcn_exp2f(v[q]); else v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][0] + c[2][0] + c[3][0] + s;
}
template <int NM, int NV, bool EXP>
void run(const char* what, float* out, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<NM, NV, EXP><<<256, threads>>>(out, 100);
  hipEventRecord(e0);
  k<NM, NV, EXP><<<256, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %d waves/SIMD: %7.1f ns per iteration\n", what, threads / 256, ms * 1e6 / iters);
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<4, 0, false>("4 MFMA", out, threads);
    run<0, 16, false>("16 v_fma_f32", out, threads);
    run<4, 16, false>("4 MFMA + 16 v_fma_f32 interleaved", out, threads);
    run<0, 32, false>("32 v_fma_f32", out, threads);
    run<4, 32, false>("4 MFMA + 32 v_fma_f32 interleaved", out, threads);
    run<0, 8, true>("8 v_exp_f32", out, threads);
    run<4, 8, true>("4 MFMA + 8 v_exp_f32 interleaved", out, threads);
    run<0, 16, true>("16 v_exp_f32", out, threads);
    run<4, 16, true>("4 MFMA + 16 v_exp_f32 interleaved", out, threads);
  }
  return 0;
}
