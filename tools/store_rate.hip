// Peak store rate of one CU's vector memory path (to price the GEMM epilogue: 164 KB per tile per CU).
// Each workgroup (8 waves, one per CU with 144 KB of LDS) stores `reps` x 1 KiB per wave in one of three patterns:
//   0: fully contiguous (64 lanes x 16 B = 1 KiB run)          1: the epilogue's pattern (6.4 rows x 160 B, row stride 5120 B)
//   2: whole 640-byte rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void store_kernel(_Float16* out, int reps, int pattern, long ld) {
  __shared__ char pad[147456];
  if (threadIdx.x == 9999) pad[0] = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 v;
  for (int e = 0; e < 8; ++e) v[e] = (_Float16)(lane + e);
  _Float16* base = out + (long)blockIdx.x * 256 * ld;   // this block's 256 rows
  for (int r = 0; r < reps; ++r) {
    long off;
    if (pattern == 0) off = ((long)(wave * reps + r) * 64 + lane) * 8;
    else if (pattern == 1) { const int t = lane + 64 * (r % 3); const int row = (wave >> 2) * 128 + (r / 3) * 16 % 128 + t / 10; off = (long)row * ld + (wave & 3) * 80 + (t % 10) * 8; if (t >= 160) continue; }
    else { const int t = lane + 64 * (r % 5); const int row = wave * 32 + (r / 5) * 8 % 32 + t / 40; off = (long)row * ld + (t % 40) * 8; }
    *reinterpret_cast<f16x8*>(base + off) = v;
  }
}
int main(int argc, char** argv) {
  const int pattern = argc > 1 ? atoi(argv[1]) : 0, reps = 20, blocks = 256 * 16;
  const long ld = 2560;
  _Float16* out;
  hipMalloc(&out, (size_t)blocks * 256 * ld * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) store_kernel<<<blocks, 512>>>(out, reps, pattern, ld);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) store_kernel<<<blocks, 512>>>(out, reps, pattern, ld);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  const double bytes = (double)blocks * 8 * reps * 1024;
  printf("pattern %d: %.1f us per launch, %.2f TB/s, %.1f us per workgroup-round of %d KB\n", pattern, ms * 1e3, bytes / ms / 1e9,
         ms * 1e3 / 16, 8 * reps);
  return 0;
}
