// 3x3 / pad 1 / stride 1 convolution with a NARROW output (Cout <= 16): the UNet's conv_out (320 -> 4 channels, unet:879-881, 1443) and
// the VAE decoder's (128 -> 3).  Reached through i2v_gemm_f16's I2V_A_CONV3X3 mode (i2v_conv_thin_try): no entry point of its own.
//
// As an implicit GEMM with N = 4 the tile kernels gather every input pixel nine times through LDS for four useful columns: 755 MB
// through the gather path for a 3-GFLOP problem at the 64^2 level -- 97 us, 0.9 TB/s of the input actually read (r5_step_shapes).
// Here a workgroup owns an 8 x 16 pixel tile of one image.  Per 64-channel block the tile WITH its halo (10 x 18 pixels x 128 B) is
// staged in LDS once -- each input byte leaves HBM 1.4 times instead of passing the gather nine times -- and the nine taps are MFMAs
// against SHIFTED reads of that tile:  D[16 output channels (zero-padded) x 16 pixels] += W_tap[16 x 32] X^T[32 x 16 pixels of a tile
// row, shifted by the tap].  The weights (Cout x 9 Cin, a few tens of KB) sit in LDS for the life of the workgroup.
// Wave w of 4 owns tile rows 2 w, 2 w + 1.  The next channel block's halo tile is fetched into registers before the current block's
// MFMAs and written to the other LDS buffer after them: one barrier per block.
#include <cstdlib>

#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int CT_TH = 8, CT_TW = 16;                 // output tile (rows x columns of pixels)
constexpr int CT_HH = CT_TH + 2, CT_HW = CT_TW + 2;  // with the halo
constexpr int CT_CB = 64;                            // channels per staged block
constexpr int CT_PS = CT_CB * 2 + 16;                // bytes per staged pixel: 128 + 16 (the 16 pixels of a fragment read start on
                                                     // 16 distinct bank groups: 36 l15 mod 64 dwords takes 16 values)
constexpr int CT_PIECES = CT_HH * CT_HW * (CT_CB / 8);   // 16-byte pieces of a halo tile: 1440
constexpr int CT_NLD = (CT_PIECES + 255) / 256;          // per thread: 6
constexpr int CT_MAXCOUT = 16;

__global__ __launch_bounds__(256, 2) void conv_thin_kernel(const i2v_gemm_params p, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][CT_HH * CT_HW * CT_PS] halo tiles, then W [cout][9 cin] fp16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int cin = p.cin, cout = p.N, nblk = cin / CT_CB, K = 9 * cin;
  char* const halo = smem;
  f16* const wl = reinterpret_cast<f16*>(smem + 2 * CT_HH * CT_HW * CT_PS);
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * CT_TH, x0 = tx * CT_TW;
  const f16* __restrict__ X = reinterpret_cast<const f16*>(p.a) + (int64_t)img * p.in_h * p.in_w * p.lda;
  const f16* __restrict__ W = reinterpret_cast<const f16*>(p.w);

  // ---- weights into LDS (16-byte pieces; K % 8 == 0)
  for (int i = tid; i < cout * (K / 8); i += 256) {
    const int r = i / (K / 8), c = i - r * (K / 8);
    *reinterpret_cast<f16x8*>(wl + r * K + c * 8) = ld_global_16B(W + (int64_t)r * p.ldw + c * 8);
  }

  // ---- halo-tile staging: piece i of a thread = 16 bytes (8 channels) of one halo pixel; zero outside the image
  int src_off[CT_NLD], dst_off[CT_NLD];     // element offset into X (< 0: outside), byte offset into the LDS tile (< 0: no piece)
#pragma unroll
  for (int i = 0; i < CT_NLD; ++i) {
    const int piece = tid + 256 * i;
    const int hp = piece >> 3, part = piece & 7;
    const int hy = hp / CT_HW, hx = hp - hy * CT_HW;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    dst_off[i] = piece < CT_PIECES ? hp * CT_PS + part * 16 : -1;
    src_off[i] = (piece < CT_PIECES && y >= 0 && y < p.in_h && x >= 0 && x < p.in_w) ? (int)((y * p.in_w + x) * p.lda + part * 8) : -1;
  }
  f16x8 st[CT_NLD];
  auto fetch = [&](const int blk) {
#pragma unroll
    for (int i = 0; i < CT_NLD; ++i) st[i] = src_off[i] >= 0 ? ld_global_16B(X + src_off[i] + blk * CT_CB) : zero8();
  };
  auto commit = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < CT_NLD; ++i)
      if (dst_off[i] >= 0) *reinterpret_cast<f16x8*>(halo + buf * (CT_HH * CT_HW * CT_PS) + dst_off[i]) = st[i];
  };
  fetch(0);
  commit(0);
  __syncthreads();

  // order of the contraction index (i2v_gemm_params.conv_kblock): 64 = channel-block-major, 0 = tap-major
  const bool kblock = p.conv_kblock == 64;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int blk = 0; blk < nblk; ++blk) {
    if (blk + 1 < nblk) fetch(blk + 1);
    const char* hb = halo + (blk & 1) * (CT_HH * CT_HW * CT_PS);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int kbase = kblock ? (blk * 9 + tap) * CT_CB : tap * cin + blk * CT_CB;
#pragma unroll
      for (int s = 0; s < CT_CB / 32; ++s) {
        // A: W[cout l15][k = kbase + 32 s + 8 g ..]; rows >= cout are zero
        const f16x8 wf = l15 < cout ? *reinterpret_cast<const f16x8*>(wl + l15 * K + kbase + 32 * s + 8 * g) : zero8();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          // B: X[pixel (row 2 wave + r + dy, column l15 + dx) of the halo tile][channels 32 s + 8 g ..]
          const f16x8 xf = *reinterpret_cast<const f16x8*>(hb + ((2 * wave + r + dy) * CT_HW + l15 + dx) * CT_PS + (32 * s + 8 * g) * 2);
          acc[r] = mfma16x16x32(wf, xf, acc[r]);
        }
      }
    }
    if (blk + 1 < nblk) {
      commit((blk + 1) & 1);      // the other buffer was last read in iteration blk - 1, closed by its barrier
      __syncthreads();
    }
  }

  // ---- D[cout 4 g + r][pixel l15] (+ bias) * out_scale -> C[pixel][cout], fp16 or fp32
  const f16* __restrict__ bias = reinterpret_cast<const f16*>(p.bias);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int64_t m = ((int64_t)img * p.in_h + y0 + 2 * wave + r) * p.in_w + x0 + l15;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 4 * g + e;
      v[e] = (acc[r][e] + (bias != nullptr && n < cout ? (float)bias[n] : 0.f)) * p.out_scale;
    }
    if (p.c_is_f32) {
      float* C = reinterpret_cast<float*>(p.c) + m * p.ldc + 4 * g;
      if (4 * g + 4 <= cout && (p.ldc & 3) == 0) {
        *reinterpret_cast<f32x4*>(C) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * g + e < cout) C[e] = v[e];
      }
    } else {
      f16* C = reinterpret_cast<f16*>(p.c) + m * p.ldc + 4 * g;
      if (4 * g + 4 <= cout && (p.ldc & 3) == 0) {
        *reinterpret_cast<f16x4*>(C) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * g + e < cout) C[e] = (f16)v[e];
      }
    }
  }
}

}  // namespace

// 0: not a problem of this kernel (the caller goes on to the tile kernels); 1: launched; < 0: error
int i2v_conv_thin_try(const i2v_gemm_params& p, hipStream_t s) {
  static const int off = getenv("I2V_CONV_THIN") ? (atoi(getenv("I2V_CONV_THIN")) == 0) : 0;
  if (off || p.a_mode != I2V_A_CONV3X3 || p.N > CT_MAXCOUT || p.stride != 1 || p.upsample || p.asym_pad) return 0;
  if (p.cin % CT_CB != 0 || (p.conv_kblock != 0 && p.conv_kblock != 64) || p.in_h % CT_TH != 0 || p.in_w % CT_TW != 0) return 0;
  if (p.epilogue != I2V_EPI_NONE || p.store_mode != I2V_STORE_ROWMAJOR || p.residual || p.rowvec || p.ln_wsum || p.rows_per_w > 0 ||
      p.a_perm_frames > 0 || p.a2 || p.residual_lo || p.c_lo)
    return 0;
  if ((int64_t)p.in_h * p.in_w * p.lda >= (1ll << 31) || (reinterpret_cast<uintptr_t>(p.c) & (p.c_is_f32 ? 15 : 7)) != 0) return 0;
  const size_t lds = 2 * (size_t)CT_HH * CT_HW * CT_PS + (size_t)p.N * 9 * p.cin * sizeof(f16);
  constexpr size_t CT_MAXLDS = 80 * 1024;        // (two workgroups per CU)
  if (lds > CT_MAXLDS) return 0;
  // (the opt-in is cached per kernel and device: asked once, for the most any problem of this kernel may use)
  const int cus = i2v_big_lds_kernel_cus(reinterpret_cast<const void*>(conv_thin_kernel), CT_MAXLDS);
  if (cus <= 0) return 0;
  const int tiles_x = p.in_w / CT_TW, tiles_y = p.in_h / CT_TH;
  const int64_t grid = (int64_t)p.n_img * tiles_x * tiles_y;
  if (grid >= (1ll << 31)) return 0;
  hipLaunchKernelGGL(conv_thin_kernel, dim3((unsigned)grid), dim3(256), lds, s, p, tiles_x, tiles_y);
  const int rc = i2v_check_launch("i2v_gemm_f16(thin conv)");
  return rc < 0 ? rc : 1;
}
