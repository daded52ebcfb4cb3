// The model handle of SURVEY 8(b) (include/i2v_hip.h, "Model handle"): configuration, a registry of the caller's weight buffers by
// key, the step's problem, a LAUNCH PLAN of one whole-model forward (unet:1289-1451 as data: entry-point ids + parameter blocks +
// pointer relocations, recorded once from the host mirror by handle.py) that i2v_unet_forward resolves and issues in C, and one
// captured hipGraph of the step (pipe:96, 676-683: `self.unet`, one UNet call per step).  Nothing here allocates device memory,
// synchronises or decides layer sequencing: the plan is the sequencing.
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

// ---- the plan blob (little-endian, written by handle.py `PlanWriter`; every section 8-byte aligned)
constexpr uint32_t PLAN_MAGIC = 0x50563249u;      // "I2VP"
constexpr uint32_t PLAN_VERSION = 1;
struct PlanHeader {
  uint32_t magic, version, abi, n_ops, n_keys, n_relocs;
  int32_t batch, frames, height, width, ctx_len, has_ip;
  uint64_t arena_bytes;
  uint32_t io_read_mask, reserved;
  uint64_t keys_off, ops_off, relocs_off, payload_off, payload_bytes, total_bytes;
};
struct PlanOp {
  uint32_t entry, payload_off, payload_bytes, reloc_begin, reloc_count, struct_bytes;      // payload = [struct | 8-byte slots]
};
struct PlanReloc {
  uint32_t offset, kind, index, pad;      // offset inside the op's payload of an 8-byte pointer; kind: 0 arena, 1 weight, 2 io
  uint64_t addend;
};
enum { RELOC_ARENA = 0, RELOC_WEIGHT = 1, RELOC_IO = 2 };

// entry points a recorded forward may contain; the order is the id (handle.py ENTRY_IDS)
enum Entry {
  E_GEMM = 0, E_ATTN, E_TATTN, E_MOTION_ATTN, E_CROSS_ATTN_FUSED, E_LN_QKV, E_FF_FUSED, E_GROUPNORM, E_LAYERNORM,      // struct + stream
  E_GROUPNORM_FOLD, E_NCHW_TO_TOKENS, E_TOKENS_TO_NCHW, E_TIMESTEP_EMBEDDING, E_SILU, E_REPEAT_ROWS, E_COPY3D, E_SELECT_ROW,
  E_PACK_CTX_FRAGMENTS, E_DDIM_PREP, E_DDIM_CFG_STEP, E_COUNT
};
struct EntryInfo {
  const char* name;
  uint32_t struct_bytes;      // sizeof of the parameter struct the payload starts with (0: none)
  uint32_t slots;             // 8-byte argument slots behind it (integers as int64, a float as a double, pointers)
};
const EntryInfo ENTRIES[E_COUNT] = {
    {"i2v_gemm_f16", sizeof(i2v_gemm_params), 0},
    {"i2v_attention_f16", sizeof(i2v_attn_params), 0},
    {"i2v_temporal_attention_f16", sizeof(i2v_tattn_params), 0},
    {"i2v_motion_attn_f16", sizeof(i2v_motion_attn_params), 0},
    {"i2v_cross_attn_fused_f16", sizeof(i2v_cross_attn_fused_params), 0},
    {"i2v_ln_qkv_f16", sizeof(i2v_ln_qkv_params), 0},
    {"i2v_ff_fused_f16", sizeof(i2v_ff_fused_params), 0},
    {"i2v_groupnorm_f16", sizeof(i2v_gn_params), 0},
    {"i2v_layernorm_f16", sizeof(i2v_ln_params), 0},
    {"i2v_groupnorm_fold_f16", sizeof(i2v_gn_params), 6},
    {"i2v_nchw_to_tokens", 0, 7},
    {"i2v_tokens_to_nchw", 0, 8},
    {"i2v_timestep_embedding", 0, 6},
    {"i2v_silu_f16", 0, 3},
    {"i2v_repeat_rows_f16", 0, 5},
    {"i2v_copy3d_f16", 0, 9},
    {"i2v_select_row_f16", 0, 6},
    {"i2v_pack_ctx_fragments_f16", 0, 10},
    {"i2v_ddim_prep", 0, 9},
    {"i2v_ddim_cfg_step", 0, 13},
};
inline uint32_t pad8(uint32_t n) { return (n + 7u) & ~7u; }

}  // namespace

struct i2v_unet {
  i2v_unet_config cfg;
  struct weight { const void* ptr; int32_t dtype, ndim; int64_t shape[4]; };
  std::map<std::string, weight> weights;
  uint64_t weights_generation = 0;
  i2v_unet_plan_t plan;
  bool planned = false;
  // launch plan
  std::vector<unsigned char> blob;
  std::vector<std::string> keys;
  std::vector<const void*> key_ptr;          // resolved through the registry at `bound_generation`
  uint64_t bound_generation = ~0ull;
  void* arena = nullptr;
  int64_t arena_bytes = 0;
  // captured step
  hipStream_t capture_stream = nullptr;
  bool capturing = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;

  const PlanHeader* header() const { return blob.empty() ? nullptr : reinterpret_cast<const PlanHeader*>(blob.data()); }
};

namespace {
void drop_step(i2v_unet* h) {
  if (h->exec) (void)hipGraphExecDestroy(h->exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  h->exec = nullptr;
  h->graph = nullptr;
}
void drop_plan(i2v_unet* h) {
  h->blob.clear();
  h->keys.clear();
  h->key_ptr.clear();
  h->bound_generation = ~0ull;
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

extern "C" int i2v_unet_abort_capture(i2v_unet* h) {
  I2V_CHECK_ARG(h != nullptr, "i2v_unet_abort_capture: null handle");
  if (!h->capturing) return I2V_OK;
  h->capturing = false;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(h->capture_stream, &g);      // (an invalidated capture returns an error and no graph: both are fine here)
  (void)hipGetLastError();
  if (g) (void)hipGraphDestroy(g);
  return I2V_OK;
}

extern "C" int i2v_unet_destroy(i2v_unet* h) {
  if (h == nullptr) return I2V_OK;
  (void)i2v_unet_abort_capture(h);      // an abandoned capture: end it so that the stream is usable again
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
  ++h->weights_generation;       // the next forward resolves the plan's keys again; a captured step keeps the pointers it was captured with
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
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_plan: a capture is in progress (i2v_unet_end_capture / i2v_unet_abort_capture first)");
  I2V_CHECK_ARG(plan->batch > 0 && plan->frames > 0 && plan->height > 0 && plan->width > 0 && plan->ctx_len > 0,
                "i2v_unet_plan: non-positive size");
  I2V_CHECK_ARG(plan->frames <= h->cfg.motion_max_seq_length, "i2v_unet_plan: num_frames %d exceeds the positional table (%d)",
                plan->frames, h->cfg.motion_max_seq_length);
  // (sizes that are not multiples of 8 take the forward_upsample_size path, unet:1304-1311: whatever the recorded plan does)
  I2V_CHECK_ARG(!plan->has_ip || h->cfg.ip_num_tokens > 0, "i2v_unet_plan: has_ip without image tokens in the configuration");
  drop_step(h);
  if (!h->planned || memcmp(&h->plan, plan, sizeof(*plan)) != 0) drop_plan(h);
  h->plan = *plan;
  h->planned = true;
  return I2V_OK;
}

extern "C" int i2v_unet_set_plan(i2v_unet* h, const void* blob, int64_t bytes) {
  I2V_CHECK_ARG(h != nullptr && blob != nullptr, "i2v_unet_set_plan: null argument");
  I2V_CHECK_ARG(h->planned, "i2v_unet_set_plan: i2v_unet_plan (the problem the plan was recorded for) first");
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_set_plan: a capture is in progress");
  I2V_CHECK_ARG(bytes >= (int64_t)sizeof(PlanHeader), "i2v_unet_set_plan: %lld bytes is not a launch plan", (long long)bytes);
  PlanHeader hd;
  memcpy(&hd, blob, sizeof(hd));
  I2V_CHECK_ARG(hd.magic == PLAN_MAGIC && hd.version == PLAN_VERSION, "i2v_unet_set_plan: bad magic / version (%#x, %u)", hd.magic, hd.version);
  I2V_CHECK_ARG(hd.abi == I2V_ABI_VERSION, "i2v_unet_set_plan: the plan was recorded against ABI %u, this library is ABI %d", hd.abi, I2V_ABI_VERSION);
  I2V_CHECK_ARG(hd.total_bytes == (uint64_t)bytes, "i2v_unet_set_plan: the blob says %llu bytes, %lld were passed", (unsigned long long)hd.total_bytes,
                (long long)bytes);
  I2V_CHECK_ARG(hd.batch == h->plan.batch && hd.frames == h->plan.frames && hd.height == h->plan.height && hd.width == h->plan.width &&
                    hd.ctx_len == h->plan.ctx_len && hd.has_ip == h->plan.has_ip,
                "i2v_unet_set_plan: recorded for (batch %d, frames %d, %d x %d, ctx %d, ip %d), planned (%d, %d, %d x %d, %d, %d)", hd.batch,
                hd.frames, hd.height, hd.width, hd.ctx_len, hd.has_ip, h->plan.batch, h->plan.frames, h->plan.height, h->plan.width,
                h->plan.ctx_len, h->plan.has_ip);
  const uint64_t total = hd.total_bytes;
  auto inside = [&](uint64_t off, uint64_t len) { return off % 8 == 0 && off <= total && len <= total - off; };
  I2V_CHECK_ARG(inside(hd.ops_off, (uint64_t)hd.n_ops * sizeof(PlanOp)) && inside(hd.relocs_off, (uint64_t)hd.n_relocs * sizeof(PlanReloc)) &&
                    inside(hd.payload_off, hd.payload_bytes) && inside(hd.keys_off, 0) && hd.keys_off <= hd.ops_off,
                "i2v_unet_set_plan: a section lies outside the blob");
  I2V_CHECK_ARG(hd.arena_bytes < (1ull << 46), "i2v_unet_set_plan: arena size");
  const unsigned char* base = reinterpret_cast<const unsigned char*>(blob);
  std::vector<std::string> keys;
  try {
    // key table: n_keys x (u32 length, bytes, padded to 4)
    uint64_t off = hd.keys_off;
    for (uint32_t i = 0; i < hd.n_keys; ++i) {
      uint32_t len = 0;
      I2V_CHECK_ARG(off + 4 <= hd.ops_off, "i2v_unet_set_plan: key table overruns");
      memcpy(&len, base + off, 4);
      off += 4;
      I2V_CHECK_ARG(len > 0 && len < 4096 && off + len <= hd.ops_off, "i2v_unet_set_plan: key %u has length %u", i, len);
      keys.emplace_back(reinterpret_cast<const char*>(base + off), len);
      off += (len + 3u) & ~3u;
    }
    // every launch: a known entry point, the struct size this library was compiled with, relocations inside the payload
    const PlanOp* ops = reinterpret_cast<const PlanOp*>(base + hd.ops_off);
    const PlanReloc* rel = reinterpret_cast<const PlanReloc*>(base + hd.relocs_off);
    for (uint32_t i = 0; i < hd.n_ops; ++i) {
      const PlanOp& op = ops[i];
      I2V_CHECK_ARG(op.entry < E_COUNT, "i2v_unet_set_plan: launch %u names entry point %u", i, op.entry);
      const EntryInfo& e = ENTRIES[op.entry];
      I2V_CHECK_ARG(op.struct_bytes == e.struct_bytes && op.payload_bytes == pad8(e.struct_bytes) + 8 * e.slots,
                    "i2v_unet_set_plan: launch %u (%s) carries a %u-byte struct + %u bytes, this library expects %u + %u: recorded against "
                    "another header", i, e.name, op.struct_bytes, op.payload_bytes, e.struct_bytes, pad8(e.struct_bytes) + 8 * e.slots);
      I2V_CHECK_ARG(op.payload_off % 8 == 0 && (uint64_t)op.payload_off + op.payload_bytes <= hd.payload_bytes &&
                        (uint64_t)op.reloc_begin + op.reloc_count <= hd.n_relocs,
                    "i2v_unet_set_plan: launch %u lies outside its section", i);
      for (uint32_t r = op.reloc_begin; r < op.reloc_begin + op.reloc_count; ++r) {
        I2V_CHECK_ARG(rel[r].offset % 8 == 0 && rel[r].offset + 8 <= op.payload_bytes, "i2v_unet_set_plan: relocation %u of launch %u", r, i);
        if (rel[r].kind == RELOC_ARENA)
          I2V_CHECK_ARG(rel[r].addend <= hd.arena_bytes, "i2v_unet_set_plan: relocation %u points outside the arena", r);
        else if (rel[r].kind == RELOC_WEIGHT)
          I2V_CHECK_ARG(rel[r].index < hd.n_keys, "i2v_unet_set_plan: relocation %u names key %u of %u", r, rel[r].index, hd.n_keys);
        else
          I2V_CHECK_ARG(rel[r].kind == RELOC_IO && rel[r].index < I2V_IO_SLOTS, "i2v_unet_set_plan: relocation %u kind %u / slot %u", r,
                        rel[r].kind, rel[r].index);
      }
    }
    drop_step(h);
    drop_plan(h);
    h->blob.assign(base, base + total);
    h->keys.swap(keys);
    h->key_ptr.assign(h->keys.size(), nullptr);
  } catch (...) {
    drop_plan(h);
    I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_set_plan: out of host memory");
  }
  return I2V_OK;
}

extern "C" int64_t i2v_unet_activation_bytes(const i2v_unet* h) {
  const PlanHeader* hd = h ? h->header() : nullptr;
  return hd ? (int64_t)hd->arena_bytes : 0;
}
extern "C" int32_t i2v_unet_plan_launches(const i2v_unet* h) {
  const PlanHeader* hd = h ? h->header() : nullptr;
  return hd ? (int32_t)hd->n_ops : 0;
}
extern "C" int32_t i2v_unet_plan_num_keys(const i2v_unet* h) { return h ? (int32_t)h->keys.size() : 0; }
extern "C" const char* i2v_unet_plan_key(const i2v_unet* h, int32_t i) {
  return (h && i >= 0 && i < (int32_t)h->keys.size()) ? h->keys[(size_t)i].c_str() : nullptr;
}

extern "C" int i2v_unet_set_workspace(i2v_unet* h, void* arena, int64_t bytes) {
  I2V_CHECK_ARG(h != nullptr, "i2v_unet_set_workspace: null handle");
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_set_workspace: a capture is in progress");
  I2V_CHECK_ARG((arena == nullptr) == (bytes == 0) && bytes >= 0 && (reinterpret_cast<uintptr_t>(arena) & 255) == 0,
                "i2v_unet_set_workspace: the arena must be 256-byte aligned with a positive size (or NULL, 0)");
  drop_step(h);          // a captured step reads the old arena
  h->arena = arena;
  h->arena_bytes = bytes;
  return I2V_OK;
}

extern "C" int i2v_unet_run(i2v_unet* h, const void* const* io_args, int32_t n_io, i2v_stream_t stream);

extern "C" int i2v_unet_forward(i2v_unet* h, const void* sample, const void* timesteps, const void* context, const void* image_embeds,
                                void* out, i2v_stream_t stream) {
  const void* io[I2V_IO_SLOTS] = {sample, timesteps, context, image_embeds, out};
  return i2v_unet_run(h, io, I2V_IO_SLOTS, stream);
}

extern "C" int i2v_unet_run(i2v_unet* h, const void* const* io_args, int32_t n_io, i2v_stream_t stream) {
  I2V_CHECK_ARG(h != nullptr && (n_io == 0 || io_args != nullptr) && n_io >= 0 && n_io <= I2V_IO_SLOTS, "i2v_unet_forward: null handle / argument array");
  const PlanHeader* hd = h->header();
  I2V_CHECK_ARG(hd != nullptr, "i2v_unet_forward: no launch plan (i2v_unet_set_plan)");
  I2V_CHECK_ARG(hd->arena_bytes == 0 || (h->arena != nullptr && (uint64_t)h->arena_bytes >= hd->arena_bytes),
                "i2v_unet_forward: the arena holds %lld bytes, the plan needs %llu (i2v_unet_set_workspace)", (long long)h->arena_bytes,
                (unsigned long long)hd->arena_bytes);
  const void* io[I2V_IO_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int s = 0; s < n_io; ++s) io[s] = io_args[s];
  // (the names of the forward's arguments; a plan of another launch sequence -- handle.py record_plan -- gives the slots its own meaning)
  static const char* io_name[I2V_IO_SLOTS] = {"sample", "timesteps", "context", "image_embeds", "out"};
  for (int s = 0; s < I2V_IO_SLOTS; ++s)
    I2V_CHECK_ARG(!((hd->io_read_mask >> s) & 1u) || io[s] != nullptr, "i2v_unet_forward: `%s` (argument %d) is NULL and the plan uses it",
                  io_name[s], s);
  // the plan's weight keys through the registry (again whenever a weight was registered since)
  if (h->bound_generation != h->weights_generation) {
    for (size_t i = 0; i < h->keys.size(); ++i) {
      const auto it = h->weights.find(h->keys[i]);
      if (it == h->weights.end()) {
        h->bound_generation = ~0ull;
        I2V_FAIL(I2V_ERR_INVALID_ARG, "i2v_unet_forward: weight `%s` of the launch plan is not registered (i2v_unet_set_weight)", h->keys[i].c_str());
      }
      h->key_ptr[i] = it->second.ptr;
    }
    h->bound_generation = h->weights_generation;
  }
  const unsigned char* base = h->blob.data();
  const PlanOp* ops = reinterpret_cast<const PlanOp*>(base + hd->ops_off);
  const PlanReloc* rel = reinterpret_cast<const PlanReloc*>(base + hd->relocs_off);
  const unsigned char* payload = base + hd->payload_off;
  // one launch's parameter block, patched on the stack: the largest struct + slots
  alignas(16) unsigned char buf[1024];
  for (uint32_t i = 0; i < hd->n_ops; ++i) {
    const PlanOp& op = ops[i];
    if (op.payload_bytes > sizeof(buf)) I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_forward: launch %u has a %u-byte parameter block", i, op.payload_bytes);
    memcpy(buf, payload + op.payload_off, op.payload_bytes);
    for (uint32_t r = op.reloc_begin; r < op.reloc_begin + op.reloc_count; ++r) {
      const PlanReloc& q = rel[r];
      const unsigned char* target = q.kind == RELOC_ARENA    ? reinterpret_cast<const unsigned char*>(h->arena)
                                    : q.kind == RELOC_WEIGHT ? reinterpret_cast<const unsigned char*>(h->key_ptr[q.index])
                                                             : reinterpret_cast<const unsigned char*>(io[q.index]);
      if (target == nullptr) I2V_FAIL(I2V_ERR_INVALID_ARG, "i2v_unet_forward: launch %u reads `%s`, which is NULL", i, io_name[q.index]);
      const uint64_t v = reinterpret_cast<uint64_t>(target) + q.addend;
      memcpy(buf + q.offset, &v, 8);
    }
    const unsigned char* sl = buf + pad8(op.struct_bytes);
    auto P = [&](int k) { void* v; memcpy(&v, sl + 8 * k, 8); return v; };
    auto I = [&](int k) { int64_t v; memcpy(&v, sl + 8 * k, 8); return v; };
    auto F = [&](int k) { double v; memcpy(&v, sl + 8 * k, 8); return (float)v; };
    int rc = I2V_OK;
    switch (op.entry) {
      case E_GEMM: rc = i2v_gemm_f16(reinterpret_cast<const i2v_gemm_params*>(buf), stream); break;
      case E_ATTN: rc = i2v_attention_f16(reinterpret_cast<const i2v_attn_params*>(buf), stream); break;
      case E_TATTN: rc = i2v_temporal_attention_f16(reinterpret_cast<const i2v_tattn_params*>(buf), stream); break;
      case E_MOTION_ATTN: rc = i2v_motion_attn_f16(reinterpret_cast<const i2v_motion_attn_params*>(buf), stream); break;
      case E_CROSS_ATTN_FUSED: rc = i2v_cross_attn_fused_f16(reinterpret_cast<const i2v_cross_attn_fused_params*>(buf), stream); break;
      case E_LN_QKV: rc = i2v_ln_qkv_f16(reinterpret_cast<const i2v_ln_qkv_params*>(buf), stream); break;
      case E_FF_FUSED: rc = i2v_ff_fused_f16(reinterpret_cast<const i2v_ff_fused_params*>(buf), stream); break;
      case E_GROUPNORM: rc = i2v_groupnorm_f16(reinterpret_cast<const i2v_gn_params*>(buf), stream); break;
      case E_LAYERNORM: rc = i2v_layernorm_f16(reinterpret_cast<const i2v_ln_params*>(buf), stream); break;
      case E_GROUPNORM_FOLD:
        rc = i2v_groupnorm_fold_f16(reinterpret_cast<const i2v_gn_params*>(buf), P(0), I(1), P(2), (int32_t)I(3), P(4), P(5), stream);
        break;
      case E_NCHW_TO_TOKENS:
        rc = i2v_nchw_to_tokens(P(0), (int32_t)I(1), P(2), (int32_t)I(3), (int32_t)I(4), (int32_t)I(5), (int32_t)I(6), stream);
        break;
      case E_TOKENS_TO_NCHW:
        rc = i2v_tokens_to_nchw(P(0), (int32_t)I(1), I(2), P(3), (int32_t)I(4), (int32_t)I(5), (int32_t)I(6), (int32_t)I(7), stream);
        break;
      case E_TIMESTEP_EMBEDDING:
        rc = i2v_timestep_embedding(reinterpret_cast<const float*>(P(0)), reinterpret_cast<const int32_t*>(P(1)), (int32_t)I(2), P(3),
                                    (int32_t)I(4), (int32_t)I(5), stream);
        break;
      case E_SILU: rc = i2v_silu_f16(P(0), P(1), I(2), stream); break;
      case E_REPEAT_ROWS: rc = i2v_repeat_rows_f16(P(0), P(1), I(2), I(3), (int32_t)I(4), stream); break;
      case E_COPY3D: rc = i2v_copy3d_f16(P(0), I(1), I(2), P(3), I(4), I(5), I(6), I(7), I(8), stream); break;
      case E_SELECT_ROW:
        rc = i2v_select_row_f16(P(0), I(1), (int32_t)I(2), reinterpret_cast<const int32_t*>(P(3)), P(4), (int32_t)I(5), stream);
        break;
      case E_PACK_CTX_FRAGMENTS:
        rc = i2v_pack_ctx_fragments_f16(P(0), I(1), P(2), I(3), I(4), P(5), (int32_t)I(6), (int32_t)I(7), (int32_t)I(8), (int32_t)I(9), stream);
        break;
      case E_DDIM_PREP:
        rc = i2v_ddim_prep(reinterpret_cast<float*>(P(0)), reinterpret_cast<const float*>(P(1)), P(2), (int32_t)I(3), (int32_t)I(4), (int32_t)I(5),
                           (int32_t)I(6), (int32_t)I(7), (int32_t)I(8), stream);
        break;
      case E_DDIM_CFG_STEP:
        rc = i2v_ddim_cfg_step(reinterpret_cast<float*>(P(0)), P(1), (int32_t)I(2), I(3), reinterpret_cast<const float*>(P(4)), (int32_t)I(5),
                               reinterpret_cast<int32_t*>(P(6)), F(7), (int32_t)I(8), (int32_t)I(9), (int32_t)I(10), (int32_t)I(11), (int32_t)I(12),
                               stream);
        break;
      default: I2V_FAIL(I2V_ERR_UNSUPPORTED, "i2v_unet_forward: launch %u names entry point %u", i, op.entry);
    }
    if (rc != I2V_OK) return rc;      // (i2v_last_error holds the entry point's own message)
  }
  return I2V_OK;
}

extern "C" int i2v_unet_capture_step(i2v_unet* h, i2v_stream_t stream) {
  I2V_CHECK_ARG(h != nullptr && stream != nullptr, "i2v_unet_capture_step: null handle or the null stream (capture needs a stream of its own)");
  I2V_CHECK_ARG(h->planned, "i2v_unet_capture_step: i2v_unet_plan first");
  I2V_CHECK_ARG(!h->capturing, "i2v_unet_capture_step: a capture is already in progress (i2v_unet_abort_capture discards it)");
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
