// Microbenchmark: issue cost of the VALU instructions of the attention softmax on gfx950, one wave on one SIMD,
// 8 independent destination registers per loop trip (s_memtime ticks; compare against v_fma_f32 = 4 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(OP)                                                                                                      \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                       \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])     \
               : "v"(x), "v"(y))
#define EXP32(i) "v_exp_f32 %" #i ", %8\n"
#define EXP16(i) "v_exp_f16 %" #i ", %8\n"
#define EXP16HI(i) "v_exp_f16_sdwa %" #i ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
#define FMA32(i) "v_fma_f32 %" #i ", %8, %9, %9\n"
#define MAX3(i) "v_max3_f32 %" #i ", %8, %9, %9\n"
#define PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %8, %9\n"
#define PKMUL16(i) "v_pk_mul_f16 %" #i ", %8, %9\n"
#define PKMAX16(i) "v_pk_max_f16 %" #i ", %8, %9\n"
#define PKADD16(i) "v_pk_add_f16 %" #i ", %8, %9\n"
#define LDEXP(i) "v_ldexp_f32 %" #i ", %8, %9\n"
#define RCP32(i) "v_rcp_f32 %" #i ", %8\n"
#define CVT16(i) "v_cvt_f16_f32 %" #i ", %8\n"
#define SUB32(i) "v_sub_f32 %" #i ", %8, %9\n"
#define MAX32(i) "v_max_f32 %" #i ", %8, %9\n"
#define MUL32(i) "v_mul_f32 %" #i ", %8, %9\n"
#define PKFMA32(i) "v_pk_fma_f32 %" #i ", %8, %9, %9\n"
#define PKMUL32(i) "v_pk_mul_f32 %" #i ", %8, %9\n"
#define MED3(i) "v_med3_f32 %" #i ", %8, %9, %9\n"
#define PERM16(i) "v_permlane16_swap_b32 %" #i ", %8\n"
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float x = -1.25f + threadIdx.x * 1e-3f, y = 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) REP8(FMA32);
    if (MODE == 1) REP8(EXP32);
    if (MODE == 2) REP8(EXP16);
    if (MODE == 3) REP8(EXP16HI);
    if (MODE == 4) REP8(MAX3);
    if (MODE == 5) REP8(PKRTZ);
    if (MODE == 6) REP8(PKMUL16);
    if (MODE == 7) REP8(PKMAX16);
    if (MODE == 8) REP8(PKADD16);
    if (MODE == 9) REP8(LDEXP);
    if (MODE == 10) REP8(RCP32);
    if (MODE == 11) REP8(CVT16);
    if (MODE == 12) REP8(SUB32);
    if (MODE == 13) REP8(MAX32);
    if (MODE == 14) REP8(MUL32);
    if (MODE == 17) REP8(MED3);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
static int THREADS = 64;
template <int M>
void run(const char* name, float* out, long long* cyc) {
  long long h = 0;
  const int iters = 20000;
  if (THREADS < 0) {   // whole chip, 2 x 1024-thread workgroups per CU = 8 waves per SIMD: wall clock
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<M><<<512, 1024>>>(out, cyc, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<M><<<512, 1024>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %.3f ns per instruction per SIMD (8 waves / SIMD, whole chip, wall clock)\n", name, ms * 1e6 / (8.0 * iters * 8));
    return;
  }
  for (int rep = 0; rep < 2; ++rep) {
    k<M><<<1, THREADS>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  const int wps = THREADS >= 256 ? THREADS / 256 : 1;
  printf("%-22s %.2f ticks per instruction per SIMD (%d waves / SIMD)\n", name, (double)h / (8.0 * iters * wps), wps);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  for (int t : {64, 512, 1024, -1}) {
  THREADS = t;
  run<0>("v_fma_f32", out, cyc); run<1>("v_exp_f32", out, cyc); run<2>("v_exp_f16", out, cyc);
  run<3>("v_exp_f16 op_sel hi", out, cyc); run<4>("v_max3_f32", out, cyc); run<5>("v_cvt_pkrtz_f16_f32", out, cyc);
  run<6>("v_pk_mul_f16", out, cyc); run<7>("v_pk_max_f16", out, cyc); run<8>("v_pk_add_f16", out, cyc);
  run<9>("v_ldexp_f32", out, cyc); run<10>("v_rcp_f32", out, cyc); run<11>("v_cvt_f16_f32", out, cyc);
  run<12>("v_sub_f32", out, cyc); run<13>("v_max_f32", out, cyc); run<14>("v_mul_f32", out, cyc);
  run<17>("v_med3_f32", out, cyc);
  }
  return 0;
}
