// Short-K GEMM with ALTERNATING wave groups (gfx950): for K <= 640 a 256 x 320 tile of gemm_big.hip spends 45 % of its time
// outside the K loop (tools/tile_timeline.py, 131072 x 2560 x 320 GEGLU: 2.4 us first-DMA wait + 1.5 us bias / GELU + 3.7 us
// staged stores beside 10.5 us of K loop), with one workgroup per CU and nothing to overlap it with.  Here the workgroup's
// eight waves are two groups of four (one wave per SIMD each) that walk the workgroup's tiles in alternation: in slot s
// group (s & 1) runs the K loop of tile s -- a 128 x 320 tile, per wave 128 x 80 = the same 160 accumulators and the same
// per-wave work as the 8-wave kernel -- while the other group runs the epilogue of tile s - 1 out of its registers.  On every
// SIMD one wave issues MFMAs and the other the epilogue's VALU / LDS / store work.  Only ONE group loads at a time, so one
// pipeline's two 64-deep stages (2 x 56 KiB) fit; the epilogue's transpose slabs have their own LDS; the K loop's last
// iteration issues the FIRST K tile of the next slot's tile (the other group's), so no slot starts with a DMA wait.
// s_barrier is workgroup-wide: both roles execute exactly `nkt` barriers per slot (the K loop one per K tile, the epilogue the
// same number spread over its blocks; a group without work in a slot just passes them).
//
// Scope: plain single-source A (M % 128 == 0), N % 320 == 0, K = 320 or 640, row-major store, epilogue none (bias, residual OR
// row vector) or GEGLU, optional folded LayerNorm.  Everything else stays on gemm_big.hip.
//
// MEASURED (round 3, profiles/r3_gemm_alt_ab.txt): bit-identical results, 1.4 - 1.7x SLOWER than the 8-wave kernel on every
// flavour (131072 x 2560 x 320 GEGLU + LayerNorm 365 -> 577 us).  The epilogue does hide -- but four loading waves on a
// 128-row tile finish a K tile's MFMAs in ~1 us while the next 56 KiB stage takes ~2.3 us to land, and LDS has no room for a
// third stage beside the slabs: the K loop runs at the DMA's latency.  OFF unless I2V_GEMM_ALT=1 (N > 320 only) / 2 (all
// eligible problems); kept as a tested A/B switch with that lesson: the bound of these GEMMs is bytes in flight per CU = one
// LDS stage per DMA round trip, for the 256-row kernel too (2.1 us per K tile against 1.3 us of MFMA).
#include <cstdlib>
#include <utility>

#include "../gemm_common.h"

namespace {

constexpr int ABM = 128, ABN = 320, ABK = 64;
constexpr int AWNC = 80, AMI = ABM / 16, ANI = AWNC / 16;
constexpr int ASTAGE = (ABM + ABN) * ABK * 2;        // 57344
constexpr int AAG = ABM * ABK / (512 * 4);           // A groups (1 KiB) per loading wave: 4
constexpr int AWG = ABN * ABK * 2 / 1024 / 4;        // W groups per loading wave: 10

__device__ __forceinline__ void adma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int alt_lds_addr(int row, int kchunk) { return row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ float axor_sum(float x, int partner_addr) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, x)));
}
template <int N, class F, int... Is>
__device__ __forceinline__ void astatic_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void astatic_for(F&& f) {
  astatic_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ int afast_div(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  const int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// NKT: K tiles per output tile (5 or 10: K = 320 / 640) -- a compile-time constant so that the epilogue's barrier stations are
// straight-line code (with a run-time count the stations were loops and the 160 accumulators went to scratch around them)
template <int EPI, bool LNF, int NKT>
__global__ __launch_bounds__(512, 1) void gemm_alt_kernel(const i2v_gemm_params p, const int tiles_n, const int ntiles) {
  constexpr int OC = EPI == I2V_EPI_GEGLU ? AWNC / 2 : AWNC;
  constexpr int LDS_LD = OC + 4;
  constexpr int SLAB = 16 * (AWNC + 4) * 4;                              // sized for the wide form
  // one slab per WAVE (not per group wave): a wave may still be storing its last block when the next slot's epilogue group starts
  __shared__ __attribute__((aligned(16))) char smem[2 * ASTAGE + 1024 + 8 * SLAB + 2 * 1024 + 2 * 1280];
  char* const slabs = smem + 2 * ASTAGE + 1024;
  float2* const lds_st_all = reinterpret_cast<float2*>(slabs + 8 * SLAB);           // [2 groups][128 rows]
  float* const lds_ws_all = reinterpret_cast<float*>(slabs + 8 * SLAB + 2 * 1024);  // [2 groups][320]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wv = wave & 3;           // group, wave inside the group (= N-wave: 80 columns each)
  const int g = lane >> 4, l15 = lane & 15;
  const int M = p.M, N = p.N, K = p.K;
  constexpr int nkt = NKT;
  float2* const lds_st = lds_st_all + grp * ABM;
  float* const lds_ws = lds_ws_all + grp * ABN;

  const f16* __restrict__ A = reinterpret_cast<const f16*>(p.a);
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, (int)(((int64_t)(N - 1) * p.ldw + K) * 2), 0x00020000);
  const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(A), 0, (int)(((int64_t)(M - 1) * p.lda + K) * 2), 0x00020000);
  // DMA lane constants (as gemm_big.hip: swizzle on the SOURCE side; groups of a wave are 32 rows apart, which leaves the
  // swizzle term unchanged)
  const int lr = lane >> 3, lc = lane & 7;
  const int u0 = 8 * wv + lr;
  const int c8 = lc ^ ((u0 >> 1) & 7);
  const unsigned a_lane = (unsigned)((u0 * (int)p.lda + c8 * 8) * 2);
  const unsigned w_lane = (unsigned)((u0 * (int)p.ldw + c8 * 8) * 2);
  auto issue = [&](int kt, int stage, int tm0, int tn0) {
    char* sa = smem + stage * ASTAGE;
    char* sw = sa + ABM * ABK * 2;
    const int kb = kt * ABK;
#pragma unroll
    for (int i = 0; i < AAG; ++i) {
      int soff = ((tm0 + 32 * i) * (int)p.lda + kb) * 2;
      asm volatile("" : "+s"(soff));
      adma16(rs_a, sa + (wv + 4 * i) * 1024, a_lane, soff);
    }
#pragma unroll
    for (int i = 0; i < AWG; ++i) adma16(rs_w, sw + (wv + 4 * i) * 1024, w_lane, ((tn0 + 32 * i) * (int)p.ldw + kb) * 2);
  };
  // tile walk: workgroup b takes tile b of every round of gridDim.x tiles (XCD remap inside a round, column panels of 8)
  const int tiles_m_all = ntiles / tiles_n;
  auto tile_coords = [&](int slot, int& tm0, int& tn0) {
    const int round0 = slot * (int)gridDim.x;
    const int in_round = min((int)gridDim.x, ntiles - round0);
    const int id = round0 + xcd_remap(blockIdx.x, in_round);
    int tm, tn;
    if (tiles_n > 8 && tiles_n % 8 == 0) {
      const int per_panel = tiles_m_all * 8;
      const int panel = id / per_panel, r = id - panel * per_panel;
      tm = r / 8;
      tn = panel * 8 + (r - tm * 8);
    } else {
      tm = id / tiles_n;
      tn = id - tm * tiles_n;
    }
    tm0 = tm * ABM;
    tn0 = tn * ABN;
  };
  // slots of this workgroup: one per tile it owns, + 1 for the last tile's epilogue
  int T = 0;
  for (int r0 = 0; r0 < ntiles; r0 += (int)gridDim.x) T += ((int)blockIdx.x < min((int)gridDim.x, ntiles - r0)) ? 1 : 0;

  f32x4 acc[ANI][AMI];
  int e_m0 = 0, e_n0 = 0;            // tile whose accumulators this group holds
  const int fa_lane = alt_lds_addr(l15, g);
  const int fw_lane = ABM * ABK * 2 + alt_lds_addr(wv * AWNC + l15, g);

  if (grp == 0 && T > 0) {           // the very first K tile: nobody ran a K loop before it
    int tm0, tn0;
    tile_coords(0, tm0, tn0);
    issue(0, 0, tm0, tn0);
  }
  for (int slot = 0; slot <= T; ++slot) {
    const bool k_role = (slot & 1) == grp && slot < T;
    const bool e_role = (slot & 1) != grp && slot >= 1;
    if (k_role) {
      int m0, n0;
      tile_coords(slot, m0, n0);
#pragma unroll
      for (int i = 0; i < ANI; ++i)
#pragma unroll
        for (int j = 0; j < AMI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (LNF) {   // this column tile's 320 weight row sums -> LDS (consumed by this group's epilogue, next slot)
        const float* lnws_g = reinterpret_cast<const float*>(p.ln_wsum);
        const int t4 = tid & 255;
        const float c0 = lnws_g[n0 + t4];
        const float c1 = t4 < ABN - 256 ? lnws_g[n0 + 256 + t4] : 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_ws[t4] = c0;
        if (t4 < ABN - 256) lds_ws[256 + t4] = c1;
      }
      float ln_s[2], ln_q[2];
      ln_s[0] = ln_s[1] = ln_q[0] = ln_q[1] = 0.f;
      for (int kt = 0; kt < nkt; ++kt) {
        // own DMA pieces of K tile kt (K tile 0 of a later slot was issued AND waited for by the other group)
        if (kt > 0 || slot == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int cur = (slot * nkt + kt) & 1;
        if (kt + 1 < nkt) {
          issue(kt + 1, cur ^ 1, m0, n0);
        } else if (slot + 1 < T) {
          int nm0, nn0;
          tile_coords(slot + 1, nm0, nn0);
          issue(0, cur ^ 1, nm0, nn0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 wf[ANI], af[AMI];
          int so = cur * ASTAGE;
          asm volatile("" : "+s"(so));
          const char* pw = smem + ((fw_lane ^ (ks << 6)) + so);
          const char* pa = smem + ((fa_lane ^ (ks << 6)) + so);
#pragma unroll
          for (int i = 0; i < ANI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(pw + i * 2048);
#pragma unroll
          for (int j = 0; j < AMI; ++j) af[j] = *reinterpret_cast<const f16x8*>(pa + j * 2048);
          if (LNF) {   // row sums of x and x^2 from the A fragments: N-wave wv takes row blocks 2 wv, 2 wv + 1
            astatic_for<AMI>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              if (wv == j / 2) {
                const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const f16x2 a2 = {af[j][2 * e], af[j][2 * e + 1]};
                  ln_s[j % 2] = __builtin_amdgcn_fdot2(a2, one2, ln_s[j % 2], false);
                  ln_q[j % 2] = __builtin_amdgcn_fdot2(a2, a2, ln_q[j % 2], false);
                }
              }
            });
          }
#pragma unroll
          for (int i = 0; i < ANI; ++i)
#pragma unroll
            for (int j = 0; j < AMI; ++j) acc[i][j] = mfma16x16x32(wf[i], af[j], acc[i][j]);
        }
      }
      if (LNF) {
        const float inv_k = 1.0f / (float)K;
        int pl = lane;
        asm volatile("" : "+v"(pl));
        const int a16 = (pl ^ 16) << 2, a32 = (pl ^ 32) << 2;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float sv = axor_sum(axor_sum(ln_s[b], a16), a32), qv = axor_sum(axor_sum(ln_q[b], a16), a32);
          const float mean = sv * inv_k;
          const float var = fmaxf(qv * inv_k - mean * mean, 0.f);
          if (g == 0) lds_st[(wv * 2 + b) * 16 + l15] = float2{mean, rsqrtf(var + p.ln_eps)};
        }
      }
      // the next slot's first K tile (issued above) must have landed before the other group passes the next barrier
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      e_m0 = m0;
      e_n0 = n0;
    } else if (e_role) {
      // ------------------------------------------------------------ epilogue of tile (e_m0, e_n0), `nkt` barriers inside
      constexpr int POINTS = AMI + 1;
      // station `point` (0 .. POINTS - 1) passes ceil((point + 1) NKT / POINTS) - ceil(point NKT / POINTS) barriers: NKT in
      // all, spread evenly over the epilogue, at least one at station 0 (this group's LayerNorm statistics / weight row sums
      // of the last slot become visible to all its waves there)
      auto eb = [&](auto pc) {
        constexpr int point = decltype(pc)::value;
        constexpr int cnt = ((point + 1) * NKT + POINTS - 1) / POINTS - (point * NKT + POINTS - 1) / POINTS;
#pragma unroll
        for (int i = 0; i < cnt; ++i) __builtin_amdgcn_s_barrier();
      };
      eb(std::integral_constant<int, 0>{});
      const int m0 = e_m0, n0 = e_n0;
      const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
      const f16* __restrict__ resid = reinterpret_cast<const f16*>(p.residual);
      const f16* __restrict__ rowvec = reinterpret_cast<const f16*>(p.rowvec);
      f16* __restrict__ C = reinterpret_cast<f16*>(p.c);
      const float oscale = p.out_scale;
      const float inv_rpv = 1.0f / (float)(p.rows_per_vec > 0 ? p.rows_per_vec : 1);
      const int ncol0 = n0 + wv * AWNC + g * 4;
      f16x4 b4[ANI];
      astatic_for<ANI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        b4[i] = bias ? *reinterpret_cast<const f16x4*>(bias + ncol0 + i * 16) : f16x4{0, 0, 0, 0};
      });
      constexpr int TPR = OC / 8, NT = 16 * TPR, QN = (NT + 63) / 64;
      const int out_col0 = EPI == I2V_EPI_GEGLU ? (n0 >> 1) + wv * (AWNC / 2) : n0 + wv * AWNC;
      auto task = [&](int j, int q, int& m, int& n) {
        const int t = lane + 64 * q;
        const int row = t / TPR, c = t - row * TPR;
        m = m0 + j * 16 + row;
        n = out_col0 + c * 8;
        return t < NT;
      };
      constexpr unsigned EOOB = 0x80000000u;
      const bool has_res = !LNF && EPI != I2V_EPI_GEGLU && resid != nullptr;
      const bool has_rv = EPI != I2V_EPI_GEGLU && !has_res && rowvec != nullptr;
      const bool has_add = has_res || has_rv;
      const int n_out_cols = EPI == I2V_EPI_GEGLU ? N / 2 : N;
      const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((((int64_t)M - 1) * p.ldc + n_out_cols) * 2), 0x00020000);
      const int rv_rows = p.rowvec_period > 0 ? p.rowvec_period : (M - 1) / (p.rows_per_vec > 0 ? p.rows_per_vec : 1) + 1;
      const int add_ld = has_res ? (int)p.ldr : (int)p.ld_rowvec;
      const int64_t add_rows = has_res ? M : rv_rows;
      const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(has_res ? resid : has_rv ? rowvec : C), 0,
                                                          has_add ? (int)(((add_rows - 1) * add_ld + N) * 2) : 0, 0x00020000);
      constexpr int RES_AHEAD = LNF ? 0 : 3;
      f16x8 xpre[AMI][QN];
      auto fetch_rows = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          int m, n;
          const bool ok = task(j, q, m, n);
          int xrow = m;
          if (!has_res) xrow = p.rowvec_period > 0 ? (m & (p.rowvec_period - 1)) : afast_div(m, p.rows_per_vec > 0 ? p.rows_per_vec : 1, inv_rpv);
          const unsigned off = ok ? (unsigned)((xrow * add_ld + n) * 2) : EOOB;
          xpre[j][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
      };
      if (EPI != I2V_EPI_GEGLU) astatic_for<(RES_AHEAD < AMI ? RES_AHEAD : AMI)>([&](auto jc) { fetch_rows(jc); });
      float* stg = reinterpret_cast<float*>(slabs + wave * SLAB);
      if (LNF) {
        f32x4 ws4[ANI];
        astatic_for<ANI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          ws4[i] = *reinterpret_cast<const f32x4*>(lds_ws + wv * AWNC + g * 4 + i * 16);
        });
        astatic_for<AMI>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const float2 st = lds_st[j * 16 + l15];
          astatic_for<ANI>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = st.y * (acc[i][j][r] - st.x * ws4[i][r]);
          });
        });
      }
      astatic_for<AMI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        astatic_for<ANI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b4[i][r];
          if (EPI == I2V_EPI_GEGLU) {
            acc[i][j][0] = acc[i][j][0] * gelu_erf(acc[i][j][1]) * oscale;
            acc[i][j][1] = acc[i][j][2] * gelu_erf(acc[i][j][3]) * oscale;
          }
        });
      });
      astatic_for<AMI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        eb(std::integral_constant<int, j + 1>{});
        if constexpr (j + RES_AHEAD < AMI && EPI != I2V_EPI_GEGLU) {
          fetch_rows(std::integral_constant<int, j + RES_AHEAD>{});
        }
        astatic_for<ANI>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (EPI == I2V_EPI_GEGLU)
            *reinterpret_cast<float2*>(stg + l15 * LDS_LD + i * 8 + 2 * g) = float2{acc[i][j][0], acc[i][j][1]};
          else
            *reinterpret_cast<f32x4*>(stg + l15 * LDS_LD + i * 16 + 4 * g) = acc[i][j];
        });
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          int m, n;
          const bool ok = task(j, q, m, n);
          const int t = lane + 64 * q;
          const int row = (t / TPR) & 15, c = t - (t / TPR) * TPR;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * LDS_LD + c * 8 + 4);
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (EPI != I2V_EPI_GEGLU) {
            if (has_add) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)xpre[j][q][e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= oscale;
          }
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
          const unsigned off = ok ? (unsigned)((m * (int)p.ldc + n) * 2) : EOOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_c, off, 0, 0);
        }
      });
    } else {
      for (int i = 0; i < nkt; ++i) __builtin_amdgcn_s_barrier();
    }
  }
}

}  // namespace

// 1 = launched, 0 = not for this kernel, < 0 error
int i2v_gemm_alt_try(const i2v_gemm_params& p, hipStream_t s) {
  static const int mode = getenv("I2V_GEMM_ALT") ? atoi(getenv("I2V_GEMM_ALT")) : 0;
  if (mode == 0 || p.residual_lo || p.c_lo) return 0;
  if (p.a_mode != I2V_A_PLAIN || p.a2 != nullptr || p.c_is_f32 || p.store_mode != I2V_STORE_ROWMAJOR) return 0;
  if (p.M % ABM != 0 || p.N % ABN != 0 || (p.K != 320 && p.K != 640)) return 0;   // NKT = 5 / 10 instantiations
  if (p.rows_per_w > 0 || p.a_perm_frames > 0 || p.workspace_bytes < 0) return 0;
  if (p.epilogue != I2V_EPI_NONE && p.epilogue != I2V_EPI_GEGLU) return 0;
  if (p.epilogue == I2V_EPI_GEGLU && (p.residual || p.rowvec)) return 0;
  if (p.residual && p.rowvec) return 0;
  if (p.ln_wsum && p.residual) return 0;
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.ldc % 8 != 0 || !a16(p.c) || (p.residual && (p.ldr % 8 != 0 || !a16(p.residual))) ||
      (p.rowvec && (p.ld_rowvec % 8 != 0 || !a16(p.rowvec))))
    return 0;
  if (p.rowvec && p.rowvec_period > 0 && (p.rowvec_period & (p.rowvec_period - 1)) != 0) return 0;
  if ((int64_t)p.M * p.lda >= (1ll << 30) || (int64_t)p.N * p.ldw >= (1ll << 30) || (int64_t)p.M * p.ldc >= (1ll << 30) ||
      (p.residual && (int64_t)p.M * p.ldr >= (1ll << 30)))
    return 0;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  const int tiles_n = p.N / ABN, ntiles = (p.M / ABM) * tiles_n;
  if (cus <= 0 || ntiles < 4 * cus) return 0;          // the alternation needs several tiles per workgroup
  // HBM-bound N = 320 projections gain nothing from hiding the epilogue (mode 2 takes them too, for the A/B)
  if (mode == 1 && p.N <= ABN) return 0;
  const dim3 grid(cus), block(512);
#define I2V_ALT_LAUNCH(EPI, LNF)                                                                                   \
  do {                                                                                                             \
    if (p.K == 320) hipLaunchKernelGGL((gemm_alt_kernel<EPI, LNF, 5>), grid, block, 0, s, p, tiles_n, ntiles);     \
    else hipLaunchKernelGGL((gemm_alt_kernel<EPI, LNF, 10>), grid, block, 0, s, p, tiles_n, ntiles);               \
  } while (0)
  if (p.ln_wsum != nullptr) {
    if (p.epilogue == I2V_EPI_GEGLU) I2V_ALT_LAUNCH(I2V_EPI_GEGLU, true);
    else I2V_ALT_LAUNCH(I2V_EPI_NONE, true);
  } else {
    if (p.epilogue == I2V_EPI_GEGLU) I2V_ALT_LAUNCH(I2V_EPI_GEGLU, false);
    else I2V_ALT_LAUNCH(I2V_EPI_NONE, false);
  }
#undef I2V_ALT_LAUNCH
  const int rc = i2v_check_launch("i2v_gemm_f16(alternating groups)");
  return rc < 0 ? rc : 1;
}
