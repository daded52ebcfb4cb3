// The model handle of SURVEY 8(b) (include/i2v_hip.h, "Model handle"): configuration, a registry of the caller's weight buffers by
// state-dict key, the plan of one denoising step and ONE captured hipGraph of the step the host launches through the per-kernel
// entry points (pipe:96, 676-683: `self.unet`, one UNet call per step).  No layer sequencing lives here.
#include <map>
#include <string>

#include "common.h"

struct i2v_unet {
  i2v_unet_config cfg;
  struct weight { const void* ptr; int32_t dtype, ndim; int64_t shape[4]; };
  std::map<std::string, weight> weights;
  i2v_unet_plan_t plan;
  bool planned = false;
  hipStream_t capture_stream = nullptr;
  bool capturing = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

namespace {
void drop_step(i2v_unet* h) {
  if (h->exec) (void)hipGraphExecDestroy(h->exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  h->exec = nullptr;
  h->graph = nullptr;
}
}  // namespace

extern "C" int i2v_unet_create(const i2v_unet_config* cfg, i2v_unet** out) {
  I2V_CHECK_ARG(cfg != nullptr && out != nullptr, "i2v_unet_create: null argument");
  I2V_CHECK_ARG(cfg->in_channels > 0 && cfg->out_channels > 0 && cfg->layers_per_block > 0 && cfg->num_attention_heads > 0 &&
                    cfg->cross_attention_dim > 0 && cfg->norm_num_groups > 0 && cfg->motion_max_seq_length > 0 &&
                    cfg->motion_num_attention_heads > 0 && cfg->ip_num_tokens >= 0,
                "i2v_unet_create: non-positive size in the configuration");
  for (int i = 0; i < 4; ++i) {
    const int c = cfg->block_out_channels[i];
    I2V_CHECK_ARG(c > 0 && c % cfg->norm_num_groups == 0 && c % cfg->num_attention_heads == 0 && c % 8 == 0,
                  "i2v_unet_create: block_out_channels[%d] = %d must be a positive multiple of 8, of norm_num_groups (%d) and of "
                  "num_attention_heads (%d)", i, c, cfg->norm_num_groups, cfg->num_attention_heads);
  }
  i2v_unet* h = new (std::nothrow) i2v_unet();
  if (h == nullptr) I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_create: out of host memory");
  h->cfg = *cfg;
  *out = h;
  return I2V_OK;
}

extern "C" int i2v_unet_destroy(i2v_unet* h) {
  if (h == nullptr) return I2V_OK;
  if (h->capturing) {          // an abandoned capture: end it so that the stream is usable again
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(h->capture_stream, &g);
    if (g) (void)hipGraphDestroy(g);
  }
  drop_step(h);
  delete h;
  return I2V_OK;
}

extern "C" int i2v_unet_set_weight(i2v_unet* h, const char* key, const void* ptr, int32_t dtype, int32_t ndim, const int64_t* shape) {
  I2V_CHECK_ARG(h != nullptr && key != nullptr && key[0] != 0 && ptr != nullptr, "i2v_unet_set_weight: null handle, key or pointer");
  I2V_CHECK_ARG(dtype == I2V_DTYPE_F16 || dtype == I2V_DTYPE_F32, "i2v_unet_set_weight: dtype %d of `%s` (fp16 = 0 or fp32 = 1)", dtype, key);
  I2V_CHECK_ARG(ndim >= 0 && ndim <= 4 && (ndim == 0 || shape != nullptr), "i2v_unet_set_weight: ndim %d of `%s`", ndim, key);
  i2v_unet::weight w = {ptr, dtype, ndim, {1, 1, 1, 1}};
  for (int i = 0; i < ndim; ++i) {
    I2V_CHECK_ARG(shape[i] > 0, "i2v_unet_set_weight: shape[%d] = %lld of `%s`", i, (long long)shape[i], key);
    w.shape[i] = shape[i];
  }
  I2V_CHECK_ARG((reinterpret_cast<uintptr_t>(ptr) & (dtype == I2V_DTYPE_F16 ? 1 : 3)) == 0, "i2v_unet_set_weight: `%s` is misaligned", key);
  try {                          // (no exception crosses the C ABI)
    h->weights[std::string(key)] = w;
  } catch (...) {
    I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_set_weight: out of host memory registering `%s`", key);
  }
  return I2V_OK;
}

extern "C" int i2v_unet_get_weight(const i2v_unet* h, const char* key, const void** ptr, int32_t* dtype, int32_t* ndim, int64_t* shape) {
  I2V_CHECK_ARG(h != nullptr && key != nullptr && ptr != nullptr, "i2v_unet_get_weight: null argument");
  std::map<std::string, i2v_unet::weight>::const_iterator it;
  try {
    it = h->weights.find(std::string(key));
  } catch (...) {
    I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_get_weight: out of host memory looking up `%s`", key);
  }
  if (it == h->weights.end()) {
    *ptr = nullptr;
    return I2V_OK;
  }
  *ptr = it->second.ptr;
  if (dtype) *dtype = it->second.dtype;
  if (ndim) *ndim = it->second.ndim;
  if (shape)
    for (int i = 0; i < it->second.ndim; ++i) shape[i] = it->second.shape[i];
  return I2V_OK;
}

extern "C" int64_t i2v_unet_num_weights(const i2v_unet* h) { return h ? (int64_t)h->weights.size() : 0; }

extern "C" int i2v_unet_plan(i2v_unet* h, const i2v_unet_plan_t* plan) {
  I2V_CHECK_ARG(h != nullptr && plan != nullptr, "i2v_unet_plan: null argument");
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_plan: a capture is in progress");
  I2V_CHECK_ARG(plan->batch > 0 && plan->frames > 0 && plan->height > 0 && plan->width > 0 && plan->ctx_len > 0,
                "i2v_unet_plan: non-positive size");
  I2V_CHECK_ARG(plan->frames <= h->cfg.motion_max_seq_length, "i2v_unet_plan: num_frames %d exceeds the positional table (%d)",
                plan->frames, h->cfg.motion_max_seq_length);
  // three stride-2 down-samplers: the halving must be exact at every level (unet:1304-1311 forward_upsample_size is not taken)
  I2V_CHECK_ARG(plan->height % 8 == 0 && plan->width % 8 == 0, "i2v_unet_plan: latent height %d / width %d must be multiples of 8",
                plan->height, plan->width);
  I2V_CHECK_ARG(!plan->has_ip || h->cfg.ip_num_tokens > 0, "i2v_unet_plan: has_ip without image tokens in the configuration");
  drop_step(h);
  h->plan = *plan;
  h->planned = true;
  return I2V_OK;
}

extern "C" int64_t i2v_unet_activation_bytes(const i2v_unet* h) {
  if (h == nullptr || !h->planned) return 0;
  return (int64_t)h->plan.batch * h->plan.frames * h->plan.height * h->plan.width * h->cfg.block_out_channels[0] * 2;
}

extern "C" int i2v_unet_capture_step(i2v_unet* h, i2v_stream_t stream) {
  I2V_CHECK_ARG(h != nullptr && stream != nullptr, "i2v_unet_capture_step: null handle or the null stream (capture needs a stream of its own)");
  I2V_CHECK_ARG(h->planned, "i2v_unet_capture_step: i2v_unet_plan first");
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_capture_step: a capture is already in progress");
  drop_step(h);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    I2V_FAIL(I2V_ERR_LAUNCH, "i2v_unet_capture_step: hipStreamBeginCapture: %s", hipGetErrorString(e));
  }
  h->capture_stream = s;
  h->capturing = true;
  return I2V_OK;
}

extern "C" int i2v_unet_end_capture(i2v_unet* h) {
  I2V_CHECK_ARG(h != nullptr && h->capturing, "i2v_unet_end_capture: no capture in progress");
  h->capturing = false;
  hipError_t e = hipStreamEndCapture(h->capture_stream, &h->graph);
  if (e != hipSuccess || h->graph == nullptr) {
    (void)hipGetLastError();
    h->graph = nullptr;
    I2V_FAIL(I2V_ERR_LAUNCH, "i2v_unet_end_capture: hipStreamEndCapture: %s", hipGetErrorString(e));
  }
  e = hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    drop_step(h);
    I2V_FAIL(I2V_ERR_LAUNCH, "i2v_unet_end_capture: hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  return I2V_OK;
}

extern "C" int i2v_unet_replay_step(i2v_unet* h, i2v_stream_t stream) {
  I2V_CHECK_ARG(h != nullptr && h->exec != nullptr, "i2v_unet_replay_step: no captured step");
  const hipError_t e = hipGraphLaunch(h->exec, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    I2V_FAIL(I2V_ERR_LAUNCH, "i2v_unet_replay_step: hipGraphLaunch: %s", hipGetErrorString(e));
  }
  return I2V_OK;
}

extern "C" int32_t i2v_unet_has_step(const i2v_unet* h) { return h != nullptr && h->exec != nullptr; }
