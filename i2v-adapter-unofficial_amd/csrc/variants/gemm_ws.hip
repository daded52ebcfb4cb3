// Weight-stationary GEMM for the K = 320 projections of the 64 x 64 level (gfx950): C[M, N] = epi(A[M, 320] W[N, 320]^T).
//
// Why a second GEMM kernel.  At K = 320 the 256 x 320 x 64 tile kernel of gemm_big.hip spends 45 % of a tile outside its K
// loop (first-DMA wait, bias / GELU pass, staged stores: DESIGN section 8 item 2) with one workgroup per CU and nothing to
// overlap it with, and inside the loop it waits on the ONE 72 KiB stage it can keep in flight: the 131072-row projections of
// the step (q | k | v, out-projections, GEGLU: 8.5 ms of a 53 ms step) run at 2 - 2.5x their HBM / MFMA bound.  With K this
// short the WEIGHTS are the small operand: an 80-column slice of W is 80 x 320 halfs = 200 VGPRs per lane as MFMA
// fragments.  So here
//   * each of a workgroup's 4 waves (one per SIMD, the whole 512-register file) keeps its 80-column slice of W in
//     REGISTERS for the whole launch and walks 64-row blocks of A: no W staging, no W fragment reads from LDS;
//   * only A goes through LDS: a 64 x 320 block is 40 KiB, so THREE blocks fit (two in flight by LDS-DMA while one is
//     multiplied) -- the bytes-in-flight limit of the big tile (one stage) does not apply;
//   * one s_barrier per 64-row block = per 200 MFMAs of a wave (the big tile: per 80), and the epilogue of block b (private
//     LDS slab, row-contiguous 16-byte stores, residual rows prefetched at the top of the block) runs under the DMA of
//     blocks b + 1 and b + 2 instead of stopping the CU;
//   * the LayerNorm fold costs no cross-wave traffic: every wave reads every A fragment of the block anyway, so each
//     computes the row moments itself (v_dot2 on the fragments) and they come out on the lanes that own the rows.
// Work split: 256 workgroups; the 32 that share an XCD (blockIdx & 7) take a slice of the rows and all N / 320 column tiles of
// it, walking the rows in step, so a block of A is fetched from HBM once per XCD and W (<= 1.6 MB) lives in every L2.
//
// RESULT (round 4, profiles/r4_gemm_ws_ab.txt): correct on every flavour (tests/test_kernels_gpu.py::test_gemm_weight_
// stationary_k320) and NOT faster -- 46 / 58 / 59 / 117 / 163 / 465 us against 44 / 50 / 48 / 91 / 146 / 384 us of the 8-wave
// tile kernel (N = 320 plain / + residual / + LayerNorm, N = 640 + table, N = 960, N = 2560 GEGLU).  With ONE wave per SIMD
// every stall is exposed: the MFMA loop alone runs at 4170 cycles per 64-row block (3200 of MFMA), the epilogue costs
// another 3030 ... 6040, and cutting the epilogue into slices placed between the MFMAs (sched_group_barrier: the ISA shows
// 2 - 8 fillers per gap) did not overlap them: 7240 cycles per block either way.  PMC: 3.7 - 8.3 VALU instructions per MFMA
// (register copies out of the AGPR half of the file, address arithmetic), 25 - 33 % of wave cycles in s_waitcnt / barrier.
// Kept out of the default build (csrc/variants/, -DI2V_VARIANTS, I2V_GEMM_WS=1).
//
// MFMA orientation as in gemm_big.hip: D = W_frag * A_frag, lane (g, l15) owns row m = l15 of the 16-row block and the 4
// consecutive columns n = 16 i + 4 g + r -- the GEGLU (value, gate) pair sits in one lane.
#include <cstdlib>
#include <utility>

#include "../gemm_common.h"

namespace {

constexpr int WS_K = 320, WS_BN = 320, WS_BM = 64, WS_NS = 3;
constexpr int WS_KSTEPS = WS_K / 32;          // 10
constexpr int WS_ATILE = WS_BM * WS_K * 2;    // 40960 bytes
constexpr int WS_DMA = WS_ATILE / 1024 / 4;   // 1 KiB groups per wave and block: 10
constexpr int WS_MI = WS_BM / 16;             // 4 row blocks of 16
constexpr int WS_NI = 5;                      // 80 columns per wave

__device__ __forceinline__ void ws_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0,
                                           0);
}
template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N, class F, int... Is>
__device__ __forceinline__ void ws_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
  ws_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// EPI: I2V_EPI_NONE / I2V_EPI_GEGLU.  LNF: LayerNorm of A's rows folded in (p.ln_wsum).  ADD: 0 nothing, 1 residual
// [M, N], 2 periodic row-vector table [period, N] (the motion modules' positional table through W).
template <int EPI, bool LNF, int ADD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_ws_kernel(const i2v_gemm_params p, const int tiles_n, const int subs) {
  constexpr bool GEGLU = EPI == I2V_EPI_GEGLU;
  constexpr int OC = GEGLU ? 40 : 80;             // output columns per wave
  constexpr int LD = OC + 4;                      // slab row stride (floats): 16 rows start on 16 distinct bank groups
  constexpr int TPR = OC / 8, NT = 16 * TPR;      // 8-column tasks per row / per 16-row block
  constexpr int QN = (NT + 63) / 64;              // passes over a block's tasks: 3 / 2
  // three A stages, the four waves' transpose slabs, and this column tile's bias / LayerNorm weight-row sums as fp32 (parked
  // once per workgroup: read per block in the epilogue instead of living in 30 registers through the MFMA loop)
  constexpr int SLAB0 = WS_NS * WS_ATILE, VEC0 = SLAB0 + 4 * 16 * 84 * 4;
  // ADD == 2: + this column tile's slice of the positional table, [period <= 16][320] fp16 (10 KB: the epilogue then needs no
  // global loads at all)
  constexpr int PE0 = VEC0 + 2 * WS_BN * 4, WS_MAXP = 16;
  __shared__ __attribute__((aligned(16))) char smem[PE0 + (ADD == 2 ? WS_MAXP * WS_BN * 2 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  // ---- this workgroup's column tile and row-block range (see the header: the 32 workgroups of an XCD share row slices)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  if (local >= subs * tiles_n) return;
  const int tn = local % tiles_n, sub = local / tiles_n;
  const int n_slices = 8 * subs, slice = xcd * subs + sub;
  const int rb_total = p.M / WS_BM;
  const int b0 = (int)(((int64_t)slice * rb_total) / n_slices), b1 = (int)(((int64_t)(slice + 1) * rb_total) / n_slices);
  if (b0 >= b1) return;
  const int n0 = tn * WS_BN;

  const f16* __restrict__ A = reinterpret_cast<const f16*>(p.a);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);

  // ---- the wave's 80 x 320 slice of W as MFMA fragments: lane holds W[n = 16 i + l15][k = 32 ks + 8 g .. + 7]
  f16x8 wf[WS_NI][WS_KSTEPS];
  {
    const f16* wrow = W + (int64_t)(n0 + wn * 80 + l15) * p.ldw + 8 * g;
#pragma unroll
    for (int i = 0; i < WS_NI; ++i)
#pragma unroll
      for (int ks = 0; ks < WS_KSTEPS; ++ks) wf[i][ks] = ld_global_16B(wrow + (int64_t)(16 * i) * p.ldw + 32 * ks);
  }
  // per-column epilogue constants of this tile -> LDS (visible after the first block's barrier)
  float* const lds_bias = reinterpret_cast<float*>(smem + VEC0);
  float* const lds_wsum = lds_bias + WS_BN;
  for (int c = tid; c < WS_BN; c += 256) {
    lds_bias[c] = p.bias ? (float)reinterpret_cast<const f16*>(p.bias)[n0 + c] : 0.f;
    if constexpr (LNF) lds_wsum[c] = reinterpret_cast<const float*>(p.ln_wsum)[n0 + c];
  }
  const int ecol = wn * 80 + 4 * g;   // this lane's columns inside the tile: ecol + 16 i + r
  f16* const lds_pe = reinterpret_cast<f16*>(smem + PE0);
  if constexpr (ADD == 2) {
    const f16* tab = reinterpret_cast<const f16*>(p.rowvec);
    for (int e = tid; e < p.rowvec_period * (WS_BN / 8); e += 256) {
      const int r = e / (WS_BN / 8), c = e - r * (WS_BN / 8);
      *reinterpret_cast<f16x8*>(lds_pe + r * WS_BN + 8 * c) = ld_global_16B(tab + (int64_t)r * p.ld_rowvec + n0 + 8 * c);
    }
  }

  // ---- A by LDS-DMA.  LDS image of a block: 5 sub-tiles (64 k each) of 64 rows x 128 bytes, chunk c of row r at
  //      c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments); the swizzle is applied on the SOURCE side of the DMA.
  //      A wave's 10 instructions per block move the 1 KiB groups (sub-tile kt, rows 8 rg .. 8 rg + 7) with
  //      8 kt + rg = wn + 4 i: rg's parity is the wave's, so the lane's source chunk is one constant.
  const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, (int)((((int64_t)p.M - 1) * p.lda + WS_K) * 2),
                                                      0x00020000);
  const int lr = lane >> 3, lc = lane & 7;
  const int c8 = lc ^ ((4 * (wn & 1) + (lr >> 1)) & 7);
  const unsigned a_lane = (unsigned)((lr * (int)p.lda + 8 * c8) * 2);
  auto issue = [&](int b, int stage) {
    char* sa = smem + stage * WS_ATILE;
    const int m0 = b * WS_BM;
#pragma unroll
    for (int i = 0; i < WS_DMA; ++i) {
      const int gi = wn + 4 * i, kt = gi >> 3, rg = gi & 7;
      int soff = ((m0 + 8 * rg) * (int)p.lda + 64 * kt) * 2;
      asm volatile("" : "+s"(soff));
      ws_dma16(rs_a, sa + kt * 8192 + rg * 1024, a_lane, soff);
    }
  };
  // fragment address of (row block j, k-step ks): stage + (ks >> 1) * 8192 + j * 2048 + (fa ^ ((ks & 1) << 6))
  const int fa = l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4);

  // ---- epilogue operands
  f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
  const int n_out_cols = GEGLU ? p.N / 2 : p.N;
  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((((int64_t)p.M - 1) * p.ldc + n_out_cols) * 2), 0x00020000);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<f16*>(reinterpret_cast<const f16*>(ADD == 1 ? p.residual : p.c)), 0,
      ADD == 1 ? (int)((((int64_t)p.M - 1) * p.ldr + p.N) * 2) : 0, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int out_col0 = GEGLU ? (n0 >> 1) + wn * 40 : n0 + wn * 80;
  float* const stg = reinterpret_cast<float*>(smem + SLAB0) + wn * (16 * 84);
  const float oscale = p.out_scale;
  const float inv_k = 1.0f / (float)WS_K;

  // ---- software pipeline inside the wave.  With one wave per SIMD nothing else runs while a wave is in its epilogue, so the
  //      epilogue of a finished 16-row block ("pending") is cut into slices that sit BETWEEN the MFMAs of the next 16-row
  //      block (the loop order is row block j outer, k-step inner: a row block's 20 accumulators are final after 10 steps).
  //      First form of this kernel, phases in series: 4170 cycles of MFMA loop + 3030 (plain) ... 6040 (GEGLU + LayerNorm)
  //      cycles of epilogue per 64-row block (tools/ws_timeline.py).
  //      Slices of the pending block, by k-step of the running one:  1..5 fold / bias / GEGLU of column block ks - 1 and its
  //      slab write;  7 the row-contiguous re-read;  9 residual / table add, cast, 16-byte stores.
  int t_row[QN], t_c[QN];
  bool t_ok[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int t = lane + 64 * q;
    t_row[q] = (t / TPR) & 15;
    t_c[q] = t - (t / TPR) * TPR;
    t_ok[q] = t < NT;
  }
  f32x4 bb[WS_NI], ww[LNF ? WS_NI : 1];     // this lane's 20 bias values / weight-row sums (LDS copies made above)
  __syncthreads();                         // ... by all 256 threads (also retires every load issued so far)
#pragma unroll
  for (int i = 0; i < WS_NI; ++i) {
    bb[i] = *reinterpret_cast<const f32x4*>(lds_bias + ecol + 16 * i);
    if constexpr (LNF) ww[i] = *reinterpret_cast<const f32x4*>(lds_wsum + ecol + 16 * i);
  }
  issue(b0, 0);
  if (b0 + 1 < b1) issue(b0 + 1, 1);
  f32x4 pend[WS_NI];
#pragma unroll
  for (int i = 0; i < WS_NI; ++i) pend[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float pend_s = 0.f, pend_q = 0.f, pmean = 0.f, prstd = 1.f;
  int pend_m = -1;                         // first row of the pending block (-1: none yet -- its stores are dropped)
  f32x4 lo[QN], hi[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) lo[q] = hi[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 xprev3[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) xprev3[q] = zero8();

  auto epi_slice = [&](auto ksc, const f16x8 (&xs)[QN]) {
    constexpr int ks = decltype(ksc)::value;
    if constexpr (ks == 1 && LNF) {
      // a row's 32 k of one step sit on the 4 lane groups: fold them; every lane then holds its row's moments
      float sv = pend_s, qv = pend_q;
      sv += __shfl_xor(sv, 16, 64);
      qv += __shfl_xor(qv, 16, 64);
      sv += __shfl_xor(sv, 32, 64);
      qv += __shfl_xor(qv, 32, 64);
      pmean = sv * inv_k;
      prstd = rsqrtf(fmaxf(qv * inv_k - pmean * pmean, 0.f) + p.ln_eps);
    }
    if constexpr (ks >= 1 && ks <= WS_NI) {
      constexpr int i = ks - 1;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = pend[i][r];
      if constexpr (LNF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = prstd * (v[r] - pmean * ww[i][r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bb[i][r];
      if (GEGLU)
        *reinterpret_cast<float2*>(stg + l15 * LD + 8 * i + 2 * g) =
            float2{v[0] * gelu_erf(v[1]) * oscale, v[2] * gelu_erf(v[3]) * oscale};
      else
        *reinterpret_cast<f32x4*>(stg + l15 * LD + 16 * i + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
    }
    if constexpr (ks == 7) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        lo[q] = *reinterpret_cast<const f32x4*>(stg + t_row[q] * LD + 8 * t_c[q]);
        hi[q] = *reinterpret_cast<const f32x4*>(stg + t_row[q] * LD + 8 * t_c[q] + 4);
      }
    }
    if constexpr (ks == 9) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        float v[8] = {lo[q][0], lo[q][1], lo[q][2], lo[q][3], hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
        if (!GEGLU) {
          if constexpr (ADD == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)xs[q][e];
          }
          if constexpr (ADD == 2) {   // positional table row (m mod period) of this tile, from LDS
            const f16x8 pe = *reinterpret_cast<const f16x8*>(lds_pe + ((pend_m + t_row[q]) & (p.rowvec_period - 1)) * WS_BN +
                                                             wn * 80 + 8 * t_c[q]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)pe[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= oscale;
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
        const unsigned off = (t_ok[q] && pend_m >= 0) ? (unsigned)(((pend_m + t_row[q]) * (int)p.ldc + out_col0 + 8 * t_c[q]) * 2) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, off, 0, 0);
      }
    }
  };

  // vmcnt at the top of block b: the number of this wave's vector-memory operations YOUNGER than block b's DMA (they may stay
  // in flight; the queue retires in order).  Per block, in program order: [residual loads: 4 QN] [DMA of block b + 2: 10]
  // [stores: 4 QN -- the first block's first QN belong to no row block and are dropped by the range check, but count].
  auto wait_top = [&](int db, bool next_issued) {
    constexpr int X = ADD == 1 ? WS_MI * QN : 0;
#define WS_WAIT_CASE(BASE)                                             \
  if (next_issued) ws_wait_vmcnt<(BASE) + WS_DMA>(); else ws_wait_vmcnt<(BASE)>()
    if (db == 0) { WS_WAIT_CASE(0); }
    else if (db == 1) { WS_WAIT_CASE(X + 4 * QN); }
    else { WS_WAIT_CASE(4 * QN + X + 4 * QN); }
#undef WS_WAIT_CASE
  };

#ifdef I2V_WS_PROBE
  // cycle stamps of workgroup 0 / wave 0 per block: top, after the barrier, after the MFMA loop (tools/ws_timeline.py passes
  // a buffer through p.workspace)
#define WS_STAMP(k)                                                                                             \
  if (p.workspace != nullptr && blockIdx.x == 0 && tid == 0 && b - b0 < 60)                                     \
  reinterpret_cast<long long*>(p.workspace)[(b - b0) * 4 + (k)] = __builtin_amdgcn_s_memtime()
#else
#define WS_STAMP(k)
#endif
  for (int b = b0; b < b1; ++b) {
    const int stage = (b - b0) % WS_NS;
    const int m0 = b * WS_BM;
    WS_STAMP(0);
    wait_top(min(b - b0, 2), b + 1 < b1);
    // every wave's share of block b is in LDS, and every wave has finished reading block b - 1, whose stage is refilled next
    __builtin_amdgcn_s_barrier();
    WS_STAMP(1);

    // residual rows of this block, in flight under the MFMAs (issued BEFORE the DMA: the wait for them must not include
    // the DMA of block b + 2)
    f16x8 xpre[WS_MI][QN];
#pragma unroll
    for (int j = 0; j < WS_MI; ++j)
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        xpre[j][q] = zero8();
        if constexpr (ADD == 1) {
          const unsigned off = t_ok[q] ? (unsigned)(((m0 + 16 * j + t_row[q]) * (int)p.ldr + out_col0 + 8 * t_c[q]) * 2) : OOB;
          xpre[j][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
      }
    if (b + 2 < b1) issue(b + 2, (stage + 2) % WS_NS);

    // ---- 64 x 80 x 320 per wave: 200 MFMAs, A fragments from LDS (requested PF steps ahead: a ring of PF + 1), W from
    //      registers; the pending row block's epilogue slices between the steps
    int so = stage * WS_ATILE;
    asm volatile("" : "+s"(so));
    const char* base = smem + so;
    constexpr int NSTEP = WS_KSTEPS * WS_MI, PF = 3;
    f16x8 ring[PF + 1];
    auto rd = [&](auto sc) {
      constexpr int s = decltype(sc)::value, j = s / WS_KSTEPS, ks = s % WS_KSTEPS;
      ring[s % (PF + 1)] = *reinterpret_cast<const f16x8*>(base + (ks >> 1) * 8192 + (fa ^ ((ks & 1) << 6)) + j * 2048);
    };
    ws_static_for<PF>([&](auto sc) { rd(sc); });
    ws_static_for<WS_MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f32x4 acc[WS_NI];
#pragma unroll
      for (int i = 0; i < WS_NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float ln_s = 0.f, ln_q = 0.f;
      ws_static_for<WS_KSTEPS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value, s = j * WS_KSTEPS + ks;
        if constexpr (s + PF < NSTEP) rd(std::integral_constant<int, s + PF>{});
        const f16x8 af = ring[s % (PF + 1)];
        if constexpr (LNF) {
          const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f16x2 a2 = {af[2 * e], af[2 * e + 1]};
            ln_s = __builtin_amdgcn_fdot2(a2, one2, ln_s, false);
            ln_q = __builtin_amdgcn_fdot2(a2, a2, ln_q, false);
          }
        }
#pragma unroll
        for (int i = 0; i < WS_NI; ++i) acc[i] = mfma16x16x32(wf[i][ks], af, acc[i]);
        if constexpr (j == 0) epi_slice(ksc, xprev3); else epi_slice(ksc, xpre[j > 0 ? j - 1 : 0]);
        // A wave issues in order: a slice placed BEHIND the step's five MFMAs only starts when the last of them has issued,
        // i.e. runs in the shadow of one MFMA (measured: no overlap at all, 7100 cycles per block against 4170 + 3030 in
        // series).  The fillers have to sit IN the gaps: after each MFMA up to WS_FILL non-MFMA instructions (VALU,
        // transcendental, SALU, LDS, vector memory) of this step's slice, in program order.
#ifndef I2V_WS_FILL
#define I2V_WS_FILL 4
#endif
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // the fragment request first
#pragma unroll
        for (int i = 0; i < WS_NI; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x496, I2V_WS_FILL, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep requests PF steps ahead of their use and the slices between the steps
      });
      // this row block becomes the pending one
#pragma unroll
      for (int i = 0; i < WS_NI; ++i) pend[i] = acc[i];
      pend_s = ln_s;
      pend_q = ln_q;
      pend_m = m0 + 16 * j;
    });
#pragma unroll
    for (int q = 0; q < QN; ++q) xprev3[q] = xpre[WS_MI - 1][q];
    WS_STAMP(2);
  }
  // the last row block's epilogue
  ws_static_for<WS_KSTEPS>([&](auto ksc) { epi_slice(ksc, xprev3); });
}

inline bool ws_al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// 1 if this kernel implements the problem (the K = 320 row-major projections of the 64 x 64 level)
int i2v_gemm_ws_ok(const i2v_gemm_params& p) {
  // OFF unless I2V_GEMM_WS=1: measured equal or slower than the 8-wave tile kernel on every flavour (profiles/r4_gemm_ws_ab.txt)
  static const int enabled = getenv("I2V_GEMM_WS") ? atoi(getenv("I2V_GEMM_WS")) : 0;
  if (!enabled || p.residual_lo || p.c_lo) return 0;
  if (p.a_mode != I2V_A_PLAIN || p.a2 != nullptr || p.K != WS_K || p.N % WS_BN != 0 || p.N / WS_BN > 32) return 0;
  if (p.M % WS_BM != 0 || p.M < 16384) return 0;                      // >= 1 row block for each of the 256 workgroups
  if (p.store_mode != I2V_STORE_ROWMAJOR || p.c_is_f32) return 0;
  if (p.rows_per_w > 0 || p.a_perm_frames > 0) return 0;
  if (p.epilogue != I2V_EPI_NONE && p.epilogue != I2V_EPI_GEGLU) return 0;
  if (p.residual && p.rowvec) return 0;
  // (the positional table of a column tile is parked in LDS: 16 rows at most -- 32-frame clips stay on gemm_big.hip)
  if (p.rowvec && !(p.rowvec_period > 0 && p.rowvec_period <= 16 && (p.rowvec_period & (p.rowvec_period - 1)) == 0)) return 0;
  if (p.epilogue == I2V_EPI_GEGLU && (p.residual || p.rowvec)) return 0;
  if (p.ln_wsum && p.residual) return 0;
  // 32-bit byte offsets into A, C and the residual
  if ((int64_t)p.M * p.lda * 2 >= (1ll << 31) || (int64_t)p.M * p.ldc * 2 >= (1ll << 31)) return 0;
  if (p.residual && (int64_t)p.M * p.ldr * 2 >= (1ll << 31)) return 0;
  if (p.lda % 8 != 0 || p.ldw % 8 != 0 || p.ldc % 8 != 0 || !ws_al16(p.a) || !ws_al16(p.w) || !ws_al16(p.c)) return 0;
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 7) != 0) return 0;
  if (p.residual && (p.ldr % 8 != 0 || !ws_al16(p.residual))) return 0;
  if (p.rowvec && (p.ld_rowvec % 8 != 0 || !ws_al16(p.rowvec))) return 0;
  if (p.ln_wsum && !ws_al16(p.ln_wsum)) return 0;
  return 1;
}

// 1 = launched, 0 = not this kernel's problem, < 0 = error
int i2v_gemm_ws_try(const i2v_gemm_params& p, hipStream_t s) {
  if (!i2v_gemm_ws_ok(p)) return 0;
  const int tiles_n = p.N / WS_BN;
  const int subs = 32 / tiles_n;
  const dim3 grid(256), block(256);
  const bool lnf = p.ln_wsum != nullptr;
  const int add = p.residual ? 1 : p.rowvec ? 2 : 0;
#define I2V_WS_LAUNCH(EPI, LNF, ADD) hipLaunchKernelGGL((gemm_ws_kernel<EPI, LNF, ADD>), grid, block, 0, s, p, tiles_n, subs)
  if (p.epilogue == I2V_EPI_GEGLU) {
    if (lnf) I2V_WS_LAUNCH(I2V_EPI_GEGLU, true, 0); else I2V_WS_LAUNCH(I2V_EPI_GEGLU, false, 0);
  } else if (lnf) {
    if (add == 2) I2V_WS_LAUNCH(I2V_EPI_NONE, true, 2); else I2V_WS_LAUNCH(I2V_EPI_NONE, true, 0);
  } else {
    if (add == 1) I2V_WS_LAUNCH(I2V_EPI_NONE, false, 1);
    else if (add == 2) I2V_WS_LAUNCH(I2V_EPI_NONE, false, 2);
    else I2V_WS_LAUNCH(I2V_EPI_NONE, false, 0);
  }
#undef I2V_WS_LAUNCH
  const int rc = i2v_check_launch("i2v_gemm_f16(weight-stationary)");
  return rc < 0 ? rc : 1;
}
