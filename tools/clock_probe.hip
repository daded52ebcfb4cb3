// Does the chip slow down when every CU computes at once?  One 8-wave workgroup per CU (150 KiB of LDS keeps a second
// one out) runs a fixed loop of MFMAs (mode 0: operands with live, pseudo-random bits; mode 1: all-zero operands),
// LDS fragment reads (mode 2) or both (mode 3) on 32 ... 256 of the CUs.  Reported: the slowest workgroup's duration on
// the constant 100 MHz counter (wall_clock64), the shader-clock ticks it saw (clock64) and the launch's wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* dur, long long* clk, int iters, int zero) {
  __shared__ __attribute__((aligned(16))) char lds[150 * 1024];
  const int tid = threadIdx.x;
  // fill the LDS with pseudo-random fp16 in [-1, 1)
  for (int i = tid; i < 150 * 1024 / 2; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<_Float16*>(lds)[i] = zero ? (_Float16)0.f : (_Float16)(((int)(h & 0xffff) - 32768) / 32768.0f);
  }
  __syncthreads();
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const h8*>(lds + ((tid * 16 + i * 8192) % (140 * 1024)));
    b[i] = *reinterpret_cast<const h8*>(lds + ((tid * 16 + i * 8192 + 4096) % (140 * 1024)));
  }
  f4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4{(float)i, 0, 0, 0};
  const long long t0 = wall_clock64();
  const long long s0 = clock64();
  int off = tid * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const h8*>(lds + off); off = (off + 8192 + 16) & 0x1fff0;
        b[i] = *reinterpret_cast<const h8*>(lds + off); off = (off + 8192 + 16) & 0x1fff0;
      }
    }
    if (MODE != 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i][0] += (float)a[i][0] + (float)b[i][7];
    }
  }
  const long long s1 = clock64();
  const long long t1 = wall_clock64();
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) { dur[blockIdx.x] = t1 - t0; clk[blockIdx.x] = s1 - s0; }
}
template <int MODE>
void run(const char* name, int zero, float* out, long long* dur, long long* clk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {300, 3000})
    for (int grid : {64, 128, 192, 256}) {
      float ms = 0; std::vector<long long> h(grid), hc(grid);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<grid, 512>>>(out, dur, clk, iters, zero);
        hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(h.data(), dur, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), clk, grid * 8, hipMemcpyDeviceToHost);
      const long long mx = *std::max_element(h.begin(), h.end()), mn = *std::min_element(h.begin(), h.end());
      const long long cx = *std::max_element(hc.begin(), hc.end());
      // per SIMD and iteration: 2 waves x 8 MFMAs
      printf("%-22s iters %5d grid %3d: workgroup %7.1f .. %7.1f us  %6.2f ns per SIMD-MFMA-slot  clock64 ticks/us %.0f  launch %.1f us\n",
             name, iters, grid, mn / 100.0, mx / 100.0, mx * 10.0 / (16.0 * iters), (double)cx / (mx / 100.0), ms * 1e3);
    }
}
int main() {
  float* out; long long *dur, *clk;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&dur, 256 * 8); (void)hipMalloc(&clk, 256 * 8);
  run<0>("MFMA random operands", 0, out, dur, clk);
  run<0>("MFMA zero operands", 1, out, dur, clk);
  run<2>("LDS reads only", 0, out, dur, clk);
  run<3>("MFMA + LDS random", 0, out, dur, clk);
  run<3>("MFMA + LDS zero", 1, out, dur, clk);
  return 0;
}
