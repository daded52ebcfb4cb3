// Measured ceilings of the chip this runs on ("re-measure both with a stream-copy and an MFMA-saturating
// microbenchmark ... quote fractions of the measured ceilings too").  Prints ONE JSON object:
//   hbm_copy_gbps   : float4 copy of 1 GiB (read + write bytes / time), median of 5 after warm-up
//   mfma_f16_tflops : v_mfma_f32_16x16x32_f16 back to back on RANDOM fp16 operands held in registers, 4 independent
//                     accumulators per wave, 1 / 2 waves per SIMD on every CU, >= 0.2 s of sustained issue (the clock
//                     the chip holds under this load is what sets the number: zeros would read ~20 % high)
//   mfma_clock_ghz  : shader clock inside that loop (delta s_memtime / delta s_memrealtime x 100 MHz, median workgroup)
//   mfma_lds_f16_tflops : the same MFMA stream with every operand re-read from LDS by ds_read_b128 (conflict-free
//                     image), i.e. the ceiling of an LDS-fed MFMA loop with no global traffic at all
// build: hipcc --offload-arch=gfx950 -O3 tools/ceilings.hip -o tools/build/ceilings ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// LDSFED = 0: operands in registers.  LDSFED = 1: 2 A + 2 B fragments re-read from LDS per 4 MFMAs (the per-MFMA LDS
// traffic of a 128 x 128 per-wave tile... far below the GEMM kernels'), so this is an upper bound for an LDS-fed loop.
template <int LDSFED>
__global__ __launch_bounds__(512) void mfma_kernel(const _Float16* __restrict__ seed, float* __restrict__ out,
                                                   long long* __restrict__ stamps, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[8 * 4 * 64 * 8];   // per wave: 4 fragments x 64 lanes x 8 halfs
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 a0, a1, b0, b1;
  for (int e = 0; e < 8; ++e) {
    a0[e] = seed[(lane * 8 + e) & 4095];
    a1[e] = seed[(lane * 8 + e + 512) & 4095];
    b0[e] = seed[(lane * 8 + e + 1024) & 4095];
    b1[e] = seed[(lane * 8 + e + 1536) & 4095];
  }
  _Float16* mine = lds + wave * 4 * 512;
  *reinterpret_cast<h8*>(mine + 0 * 512 + lane * 8) = a0;
  *reinterpret_cast<h8*>(mine + 1 * 512 + lane * 8) = a1;
  *reinterpret_cast<h8*>(mine + 2 * 512 + lane * 8) = b0;
  *reinterpret_cast<h8*>(mine + 3 * 512 + lane * 8) = b1;
  __syncthreads();
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (LDSFED) {
      a0 = *reinterpret_cast<volatile h8*>(mine + 0 * 512 + lane * 8);
      a1 = *reinterpret_cast<volatile h8*>(mine + 1 * 512 + lane * 8);
      b0 = *reinterpret_cast<volatile h8*>(mine + 2 * 512 + lane * 8);
      b1 = *reinterpret_cast<volatile h8*>(mine + 3 * 512 + lane * 8);
    }
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  // accumulators grow without bound on random data; only finiteness matters here (keep the chains live)
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int LDSFED>
static void run_mfma(int waves_per_simd, const _Float16* seed, float* out, long long* stamps, int cus, double* tflops,
                     double* ghz) {
  const int threads = 256 * waves_per_simd, blocks = cus, iters = 4000000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto launch = [&](int it) {
    if (threads == 256)
      hipLaunchKernelGGL((mfma_kernel<LDSFED>), dim3(blocks), dim3(256), 0, 0, seed, out, stamps, it);
    else
      hipLaunchKernelGGL((mfma_kernel<LDSFED>), dim3(blocks), dim3(512), 0, 0, seed, out, stamps, it);
  };
  launch(iters);   // warm-up: the clock settles under load
  hipDeviceSynchronize();
  std::vector<double> rates;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    launch(iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * (threads / 64) * iters * 4.0 * 2.0 * 16 * 16 * 32;
    rates.push_back(flops / (ms * 1e-3) / 1e12);
  }
  std::sort(rates.begin(), rates.end());
  *tflops = rates[1];
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);   // 100 MHz reference
  std::sort(clk.begin(), clk.end());
  *ghz = clk[clk.size() / 2];
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // ---- stream copy
  const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(float4);
  float4 *src, *dst;
  hipMalloc(&src, bytes);
  hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<double> bw;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(copy_kernel, dim3(cus * 32), dim3(256), 0, 0, src, dst, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep >= 3) bw.push_back(2.0 * bytes / (ms * 1e-3) / 1e9);
  }
  std::sort(bw.begin(), bw.end());
  // ---- MFMA
  std::vector<_Float16> hs(4096);
  srand(12345);
  for (auto& v : hs) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
  _Float16* seed;
  float* out;
  long long* stamps;
  hipMalloc(&seed, 4096 * 2);
  hipMalloc(&out, (size_t)cus * 512 * 4);
  hipMalloc(&stamps, (size_t)cus * 16);
  hipMemcpy(seed, hs.data(), 4096 * 2, hipMemcpyHostToDevice);
  double t1, g1, t2, g2, tl, gl;
  run_mfma<0>(1, seed, out, stamps, cus, &t1, &g1);
  run_mfma<0>(2, seed, out, stamps, cus, &t2, &g2);
  run_mfma<1>(2, seed, out, stamps, cus, &tl, &gl);
  printf("{\"cus\": %d, \"hbm_copy_gbps\": %.1f, \"hbm_copy_gbps_min_max\": [%.1f, %.1f], "
         "\"mfma_f16_tflops\": %.1f, \"mfma_clock_ghz\": %.3f, \"mfma_f16_tflops_1wave_per_simd\": %.1f, "
         "\"mfma_clock_ghz_1wave_per_simd\": %.3f, \"mfma_lds_f16_tflops\": %.1f, \"mfma_lds_clock_ghz\": %.3f, "
         "\"vendor_peak_tflops\": 2500.0, \"vendor_peak_hbm_gbps\": 8000.0}\n",
         cus, bw[bw.size() / 2], bw.front(), bw.back(), std::max(t1, t2), t2 >= t1 ? g2 : g1, t1, g1, tl, gl);
  return 0;
}
