// Microbenchmark: cycles per MFMA for the f16 shapes on gfx950 (one wave per SIMD, 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  h8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {1, 1, 1, 1, 1, 1, 1, 1};
  h4 a4 = {1, 2, 3, 4}, b4 = {1, 1, 1, 1};
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f16v d0 = {0}, d1 = d0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
                     c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0); }
    if (MODE == 1) { c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
                     c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0); }
    if (MODE == 2) { d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d1, 0, 0, 0);
                     d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d1, 0, 0, 0); }
    if (MODE == 3) { d0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d1, 0, 0, 0);
                     d0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d1, 0, 0, 0); }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + d0[0] + d1[0];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  const char* names[4] = {"16x16x32_f16", "16x16x16_f16", "32x32x16_f16", "32x32x8_f16"};
  for (int m = 0; m < 4; ++m) {
    long long h = 0; int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) k<0><<<1, 64>>>(out, cyc, iters); if (m == 1) k<1><<<1, 64>>>(out, cyc, iters);
      if (m == 2) k<2><<<1, 64>>>(out, cyc, iters); if (m == 3) k<3><<<1, 64>>>(out, cyc, iters);
      hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%s: %.2f cycles per MFMA (s_memtime ticks)\n", names[m], (double)h / (4.0 * iters));
  }
  return 0;
}
