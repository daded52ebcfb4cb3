/* A host that is NOT the Python mirror: runs unet:1289-1451 (UNetMotionCrossFrameAttnModel.forward) through the C ABI of
 * libi2v_hip.so alone -- include/i2v_hip.h "Model handle" -- from three files written offline by handle.py
 * (save_plan / save_weights) and by the test (the inputs).  No Python, no torch, no per-kernel calls on this side:
 *
 *     unet_forward_host plan.bin weights.bin inputs.bin out.bin
 *
 * It registers the weights, installs the launch plan, sizes ONE arena with i2v_unet_activation_bytes, runs i2v_unet_forward once
 * eagerly and once as a captured + replayed hipGraph step (pipe:676-683: one UNet call per DDIM step), checks that both give the
 * same bytes and writes the result.  TEST INFRASTRUCTURE (tests/test_handle_gpu.py compares out.bin with the module API's forward);
 * INTEGRATION.md section 2 walks through it.  Built by __graft_entry__.build() with hipcc (plain C, the HIP runtime for memory).
 *
 * inputs.bin: "I2VI", then int32 x 22: in_channels, out_channels, block_out_channels[4], layers_per_block, num_attention_heads,
 * cross_attention_dim, norm_num_groups, motion_max_seq_length, motion_num_attention_heads, use_motion_mid_block, ip_num_tokens,
 * batch, frames, height, width, ctx_len, clip_dim (0: no image_embeds), sample_is_f32, reserved; then the sample, timesteps (fp32),
 * context (fp16) and image_embeds (fp16) bytes back to back. */
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
  if (argc != 5) DIE("usage: %s plan.bin weights.bin inputs.bin out.bin", argv[0]);
  if (i2v_abi_version() != I2V_ABI_VERSION) DIE("library ABI %d, header ABI %d", i2v_abi_version(), I2V_ABI_VERSION);
  size_t plan_n, w_n, in_n;
  unsigned char* plan = read_file(argv[1], &plan_n);
  unsigned char* wfile = read_file(argv[2], &w_n);
  unsigned char* in = read_file(argv[3], &in_n);

  if (in_n < 4 + 22 * 4 || memcmp(in, "I2VI", 4) != 0) DIE("%s is not an inputs file", argv[3]);
  int32_t v[22];
  memcpy(v, in + 4, sizeof(v));
  i2v_unet_config cfg = {v[0], v[1], {v[2], v[3], v[4], v[5]}, v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13]};
  const int batch = v[14], frames = v[15], height = v[16], width = v[17], ctx_len = v[18], clip_dim = v[19], sample_f32 = v[20];
  const size_t n_lat = (size_t)batch * frames * height * width;
  const size_t sample_bytes = n_lat * cfg.in_channels * (sample_f32 ? 4 : 2), out_bytes = n_lat * cfg.out_channels * (sample_f32 ? 4 : 2);
  const size_t t_bytes = (size_t)batch * 4, ctx_bytes = (size_t)batch * ctx_len * cfg.cross_attention_dim * 2;
  const size_t ie_bytes = (size_t)batch * clip_dim * 2;
  const unsigned char* q = in + 4 + sizeof(v);
  if ((size_t)(q - in) + sample_bytes + t_bytes + ctx_bytes + ie_bytes != in_n) DIE("%s: size does not match its header", argv[3]);
  void* d_sample = to_device(q, sample_bytes);
  void* d_t = to_device(q + sample_bytes, t_bytes);
  void* d_ctx = to_device(q + sample_bytes + t_bytes, ctx_bytes);
  void* d_ie = clip_dim ? to_device(q + sample_bytes + t_bytes + ctx_bytes, ie_bytes) : NULL;
  void *d_out = NULL, *d_out2 = NULL;
  HIP(hipMalloc(&d_out, out_bytes));
  HIP(hipMalloc(&d_out2, out_bytes));
  HIP(hipMemset(d_out, 0xff, out_bytes));
  HIP(hipMemset(d_out2, 0xff, out_bytes));

  i2v_unet* h = NULL;
  I2V(i2v_unet_create(&cfg, &h));
  i2v_unet_plan_t problem = {batch, frames, height, width, ctx_len, clip_dim ? 1 : 0};
  I2V(i2v_unet_plan(h, &problem));
  I2V(i2v_unet_set_plan(h, plan, (int64_t)plan_n));

  /* weights.bin: "I2VW", u32 count, then per tensor: u32 key length, key, u32 dtype, u64 bytes, padding to 16, the bytes */
  if (w_n < 8 || memcmp(wfile, "I2VW", 4) != 0) DIE("%s is not a weights file", argv[2]);
  uint32_t count;
  memcpy(&count, wfile + 4, 4);
  size_t off = 8, total_w = 0;
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
    off += 16 + klen;
    off = (off + 15) & ~(size_t)15;
    if (off + bytes > w_n) DIE("%s: tensor `%s` overruns the file", argv[2], key);
    void* d = to_device(wfile + off, (size_t)bytes);
    const int64_t shape[1] = {(int64_t)(bytes / (dtype == I2V_DTYPE_F16 ? 2 : 4))};
    I2V(i2v_unet_set_weight(h, key, d, (int32_t)dtype, 1, shape));
    off += bytes;
    total_w += bytes;
  }
  for (int32_t i = 0; i < i2v_unet_plan_num_keys(h); ++i) {          /* every key the plan names must have arrived */
    const void* p = NULL;
    I2V(i2v_unet_get_weight(h, i2v_unet_plan_key(h, i), &p, NULL, NULL, NULL));
    if (!p) DIE("the plan names `%s`, which %s does not hold", i2v_unet_plan_key(h, i), argv[2]);
  }

  const int64_t arena_bytes = i2v_unet_activation_bytes(h);
  void* arena = NULL;
  HIP(hipMalloc(&arena, (size_t)arena_bytes));
  I2V(i2v_unet_set_workspace(h, arena, arena_bytes));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  /* eager */
  I2V(i2v_unet_forward(h, d_sample, d_t, d_ctx, d_ie, d_out, stream));
  HIP(hipStreamSynchronize(stream));
  /* one captured step, replayed (a real host also issues i2v_ddim_prep / i2v_ddim_cfg_step inside the capture) */
  I2V(i2v_unet_capture_step(h, stream));
  const int rc = i2v_unet_forward(h, d_sample, d_t, d_ctx, d_ie, d_out2, stream);
  if (rc != I2V_OK) {
    i2v_unet_abort_capture(h);
    DIE("i2v_unet_forward under capture = %d: %s", rc, i2v_last_error());
  }
  I2V(i2v_unet_end_capture(h));
  I2V(i2v_unet_replay_step(h, stream));
  HIP(hipStreamSynchronize(stream));

  unsigned char* o1 = (unsigned char*)malloc(out_bytes);
  unsigned char* o2 = (unsigned char*)malloc(out_bytes);
  HIP(hipMemcpy(o1, d_out, out_bytes, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(o2, d_out2, out_bytes, hipMemcpyDeviceToHost));
  if (memcmp(o1, o2, out_bytes) != 0) DIE("the replayed step differs from the eager forward");
  FILE* f = fopen(argv[4], "wb");
  if (!f || fwrite(o1, 1, out_bytes, f) != out_bytes) DIE("cannot write %s", argv[4]);
  fclose(f);
  printf("unet_forward_host: %d launches, %d weight keys (%.1f MB), arena %.1f MB, output %zu bytes, eager == replay\n",
         i2v_unet_plan_launches(h), i2v_unet_plan_num_keys(h), total_w / 1e6, arena_bytes / 1e6, out_bytes);
  I2V(i2v_unet_destroy(h));
  return 0;
}
