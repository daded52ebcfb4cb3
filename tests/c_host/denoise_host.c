/* The reference's whole denoising loop (pipe:663-700: per step frame-0 overwrite, CFG duplicate, UNet, CFG combine, DDIM update)
 * from a host that is NOT the Python mirror -- libi2v_hip.so's model handle alone (include/i2v_hip.h "Model handle", i2v_unet_run):
 *
 *     denoise_host prepare.plan step.plan weights.bin inputs.bin out.bin n_steps
 *
 * prepare.plan  what the pipeline computes once per sample (context K / V^T of the 16 cross-attention layers, ImageProjection, the
 *               time-embedding table of the schedule)                                   handle.py record_prepare_plan
 * step.plan     ONE iteration of pipe:666-697 as the pipeline's captured step issues it   handle.py record_step_plan
 * weights.bin   every buffer the two plans name: the model's kernel-layout weights AND the per-sample buffers (`sample#...`) the
 *               preparation writes and the step reads                                     handle.py save_weights
 * It runs the preparation once, captures the step as ONE hipGraph, replays it n_steps times between two events, writes the final
 * latents and prints the step time.  TEST INFRASTRUCTURE (tests/test_handle_gpu.py compares out.bin with the Python pipeline's
 * latents bit for bit); built by __graft_entry__.build() with gcc.
 *
 * inputs.bin: "I2VD", int32 x 24: the 14 fields of i2v_unet_config, then batch (CFG copies x samples), frames, height, width,
 * ctx_len, clip_dim (0: no image embeds), table rows T, samples B, 2 reserved; then latents fp32 [B, F, C, H, W], cond fp32
 * [B, C, H, W], context fp16 [batch, ctx_len, D], timesteps fp32 [T], coef fp32 [T, 4], image_embeds fp16 [batch, clip_dim]. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/i2v_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define I2V(x) do { int r_ = (x); if (r_ != I2V_OK) DIE("%s = %d: %s", #x, r_, i2v_last_error()); } while (0)

static unsigned char* read_file(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) DIE("cannot open %s", path);
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* b = (unsigned char*)malloc(*n ? *n : 1);
  if (!b || fread(b, 1, *n, f) != *n) DIE("cannot read %s", path);
  fclose(f);
  return b;
}

static void* to_device(const void* src, size_t bytes) {
  void* d = NULL;
  HIP(hipMalloc(&d, bytes ? bytes : 16));
  if (bytes) HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  if (argc != 7) DIE("usage: %s prepare.plan step.plan weights.bin inputs.bin out.bin n_steps", argv[0]);
  const int n_steps = atoi(argv[6]);
  size_t prep_n, step_n, w_n, in_n;
  unsigned char* prep = read_file(argv[1], &prep_n);
  unsigned char* stepb = read_file(argv[2], &step_n);
  unsigned char* wfile = read_file(argv[3], &w_n);
  unsigned char* in = read_file(argv[4], &in_n);
  if (in_n < 4 + 24 * 4 || memcmp(in, "I2VD", 4) != 0) DIE("%s is not a denoise inputs file", argv[4]);
  int32_t v[24];
  memcpy(v, in + 4, sizeof(v));
  i2v_unet_config cfg = {v[0], v[1], {v[2], v[3], v[4], v[5]}, v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13]};
  const int batch = v[14], frames = v[15], height = v[16], width = v[17], ctx_len = v[18], clip_dim = v[19], T = v[20], B = v[21];
  const size_t lat_bytes = (size_t)B * frames * cfg.in_channels * height * width * 4, cond_bytes = (size_t)B * cfg.in_channels * height * width * 4;
  const size_t ctx_bytes = (size_t)batch * ctx_len * cfg.cross_attention_dim * 2, t_bytes = (size_t)T * 4, coef_bytes = (size_t)T * 16;
  const size_t ie_bytes = (size_t)batch * clip_dim * 2;
  const unsigned char* q = in + 4 + sizeof(v);
  if ((size_t)(q - in) + lat_bytes + cond_bytes + ctx_bytes + t_bytes + coef_bytes + ie_bytes != in_n) DIE("%s: size does not match its header", argv[4]);
  void* d_lat = to_device(q, lat_bytes);                      q += lat_bytes;
  void* d_cond = to_device(q, cond_bytes);                    q += cond_bytes;
  void* d_ctx = to_device(q, ctx_bytes);                      q += ctx_bytes;
  void* d_t = to_device(q, t_bytes);                          q += t_bytes;
  void* d_coef = to_device(q, coef_bytes);                    q += coef_bytes;
  void* d_ie = clip_dim ? to_device(q, ie_bytes) : NULL;
  int32_t zero = 0;
  void* d_step = to_device(&zero, 4);

  i2v_unet *hp = NULL, *hs = NULL;
  I2V(i2v_unet_create(&cfg, &hp));
  I2V(i2v_unet_create(&cfg, &hs));
  i2v_unet_plan_t problem = {batch, frames, height, width, ctx_len, clip_dim ? 1 : 0};
  I2V(i2v_unet_plan(hp, &problem));
  I2V(i2v_unet_plan(hs, &problem));
  I2V(i2v_unet_set_plan(hp, prep, (int64_t)prep_n));
  I2V(i2v_unet_set_plan(hs, stepb, (int64_t)step_n));

  /* weights.bin (handle.py save_weights): every named buffer of both plans, registered with both handles */
  if (w_n < 8 || memcmp(wfile, "I2VW", 4) != 0) DIE("%s is not a weights file", argv[3]);
  uint32_t count;
  memcpy(&count, wfile + 4, 4);
  size_t off = 8;
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t klen, dtype;
    uint64_t bytes;
    char key[4096];
    memcpy(&klen, wfile + off, 4);
    if (klen >= sizeof(key)) DIE("key %u too long", i);
    memcpy(key, wfile + off + 4, klen);
    key[klen] = 0;
    memcpy(&dtype, wfile + off + 4 + klen, 4);
    memcpy(&bytes, wfile + off + 8 + klen, 8);
    off = (off + 16 + klen + 15) & ~(size_t)15;
    if (off + bytes > w_n) DIE("%s: tensor `%s` overruns the file", argv[3], key);
    void* d = to_device(wfile + off, (size_t)bytes);
    const int64_t shape[1] = {(int64_t)(bytes / (dtype == I2V_DTYPE_F16 ? 2 : 4))};
    I2V(i2v_unet_set_weight(hp, key, d, (int32_t)dtype, 1, shape));
    I2V(i2v_unet_set_weight(hs, key, d, (int32_t)dtype, 1, shape));
    off += bytes;
  }
  const int64_t a1 = i2v_unet_activation_bytes(hp), a2 = i2v_unet_activation_bytes(hs), arena_bytes = a1 > a2 ? a1 : a2;
  void* arena = NULL;
  HIP(hipMalloc(&arena, (size_t)arena_bytes));        /* one arena: the preparation has finished before the first step starts */
  I2V(i2v_unet_set_workspace(hp, arena, arena_bytes));
  I2V(i2v_unet_set_workspace(hs, arena, arena_bytes));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  /* once per sample (handle.py PREP_* slots) */
  const void* pio[3] = {d_ctx, d_t, d_ie};
  I2V(i2v_unet_run(hp, pio, 3, stream));
  /* one step, captured; then the loop of pipe:666-697 as graph replays (handle.py STEP_* slots) */
  const void* sio[4] = {d_lat, d_cond, d_step, d_coef};
  I2V(i2v_unet_capture_step(hs, stream));
  const int rc = i2v_unet_run(hs, sio, 4, stream);
  if (rc != I2V_OK) {
    i2v_unet_abort_capture(hs);
    DIE("i2v_unet_run under capture = %d: %s", rc, i2v_last_error());
  }
  I2V(i2v_unet_end_capture(hs));
  hipEvent_t e0, e1;
  HIP(hipEventCreate(&e0));
  HIP(hipEventCreate(&e1));
  HIP(hipEventRecord(e0, stream));
  for (int t = 0; t < n_steps; ++t) I2V(i2v_unet_replay_step(hs, stream));
  HIP(hipEventRecord(e1, stream));
  HIP(hipStreamSynchronize(stream));
  float ms = 0.f;
  HIP(hipEventElapsedTime(&ms, e0, e1));

  unsigned char* out = (unsigned char*)malloc(lat_bytes);
  HIP(hipMemcpy(out, d_lat, lat_bytes, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[5], "wb");
  if (!f || fwrite(out, 1, lat_bytes, f) != lat_bytes) DIE("cannot write %s", argv[5]);
  fclose(f);
  printf("denoise_host: %d + %d launches per (sample, step), arena %.1f MB, %d steps in %.3f ms = %.3f ms per step (%.2f steps/s)\n",
         i2v_unet_plan_launches(hp), i2v_unet_plan_launches(hs), arena_bytes / 1e6, n_steps, ms, ms / (n_steps > 0 ? n_steps : 1),
         n_steps > 0 ? 1e3 * n_steps / ms : 0.0);
  I2V(i2v_unet_destroy(hs));
  I2V(i2v_unet_destroy(hp));
  return 0;
}
