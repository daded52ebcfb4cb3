"""The library's model handle (include/i2v_hip.h "Model handle") seen from Python, and the RECORDER that turns one forward
of the host mirror into the launch plan `i2v_unet_forward` executes in C (unet:1289-1451; pipe:96, 676-683).

    blob, weights = record_forward_plan(unet, sample, timesteps, ctx, image_embeds=None)      # once per (problem, switches)
    h = UNetHandle(unet); h.plan(...); h.set_plan(blob); h.set_weights(weights)
    h.set_workspace(torch.empty(h.activation_bytes, dtype=torch.uint8, device=dev))
    h.forward(sample, timesteps, ctx, image_embeds, out, stream)                                # ONE ctypes call: the launches are issued in C

What a host that is NOT this package binds is exactly those C entry points; `save_plan` / `save_weights` write the two files such a
host loads (tests/c_host/unet_forward_host.c is one).  The package's own pipeline keeps torch's graph
capture (its step allocates through torch's allocator, which must know about the capture).

How the recording works: `UNetMotionCrossFrameAttnModel.forward` is run once with `_lib.load()` replaced by a proxy that logs every
launch entry point it calls -- entry id, the bytes of its parameter struct / its flat arguments -- and then calls it.  Afterwards every
device pointer in the log is classified: inside one of the forward's arguments (io slot), inside a persistent tensor of the model
(kernel-layout packs, parameters, buffers: a weight key), or inside the private memory pool the forward allocated from (the arena:
the pool's segments laid end to end, so the plan inherits torch's own buffer reuse).  Anything else is an error, not a guess.
"""
import bisect
import ctypes as C
import struct

import torch

from . import _lib

PLAN_MAGIC, PLAN_VERSION = 0x50563249, 1
# entry points a recorded forward may contain, in the order of csrc/handle.hip `enum Entry`
ENTRY_IDS = {name: i for i, name in enumerate((
    "i2v_gemm_f16", "i2v_attention_f16", "i2v_temporal_attention_f16", "i2v_motion_attn_f16", "i2v_cross_attn_fused_f16", "i2v_ln_qkv_f16",
    "i2v_ff_fused_f16", "i2v_groupnorm_f16", "i2v_layernorm_f16", "i2v_groupnorm_fold_f16", "i2v_nchw_to_tokens", "i2v_tokens_to_nchw",
    "i2v_timestep_embedding", "i2v_silu_f16", "i2v_repeat_rows_f16", "i2v_copy3d_f16", "i2v_select_row_f16",
    "i2v_pack_ctx_fragments_f16", "i2v_ddim_prep", "i2v_ddim_cfg_step"))}
IO_SAMPLE, IO_TIMESTEPS, IO_CONTEXT, IO_IMAGE_EMBEDS, IO_OUT = range(5)
RELOC_ARENA, RELOC_WEIGHT, RELOC_IO = range(3)
_INT_TYPES = (C.c_int32, C.c_int64, C.c_int)


def _pad8(n):
    return (n + 7) & ~7


class _RecordingLib:
    """stands in for the ctypes library while a forward is recorded: launch entry points are logged and then executed, everything
    else (the *_supported / *_bytes / *_rows queries) passes through."""

    def __init__(self, real):
        self._real = real
        self.ops = []            # (entry id, struct bytes, [slot values], [(payload offset, pointer value)])

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("i2v_") or name not in _lib.SIGNATURES:
            return fn
        res, argtypes = _lib.SIGNATURES[name]
        launches = bool(argtypes) and argtypes[-1] is C.c_void_p and res is C.c_int and not name.startswith("i2v_unet_")
        if not launches:
            return fn
        if name not in ENTRY_IDS:
            def refuse(*a, _n=name):
                raise _lib.HipLibraryError(f"{_n} is not an entry point a launch plan can carry (handle.ENTRY_IDS / csrc/handle.hip)")
            return refuse

        def call(*args, _n=name, _fn=fn, _at=argtypes):
            self._log(_n, args, _at)
            return _fn(*args)
        return call

    def _log(self, name, args, argtypes):
        if len(args) != len(argtypes):
            raise TypeError(f"{name}: {len(args)} arguments for {len(argtypes)} parameters")
        sbytes, slots, ptrs = b"", [], []
        for arg, at in zip(args[:-1], argtypes[:-1]):          # the last parameter is the stream
            if isinstance(at, type) and issubclass(at, C._Pointer) and issubclass(at._type_, C.Structure):
                st = arg._obj                                   # C.byref(params)
                if sbytes or slots:
                    raise TypeError(f"{name}: the parameter struct must come first")
                sbytes = bytes(st)
                for fname, ftype in st._fields_:
                    if ftype is C.c_void_p:
                        v = getattr(st, fname)
                        if v:
                            ptrs.append((getattr(type(st), fname).offset, int(v), f"{name}.{fname}"))
            elif at is C.c_void_p:
                v = arg.value if isinstance(arg, C.c_void_p) else arg
                v = int(v) if v else 0
                if v:
                    ptrs.append((_pad8(len(sbytes)) + 8 * len(slots), v, f"{name} argument {len(slots)}"))
                slots.append(struct.pack("<Q", v))
            elif at in _INT_TYPES:
                slots.append(struct.pack("<q", int(arg)))
            elif at is C.c_float:
                slots.append(struct.pack("<d", float(arg)))          # (the C side reads a float slot as a double)
            else:
                raise TypeError(f"{name}: a launch plan carries pointers, integers and floats, not {at}")
        self.ops.append((ENTRY_IDS[name], sbytes, slots, ptrs))


def persistent_tensors(unet):
    """name -> tensor of everything a forward may read that outlives it: the kernel-layout packs of every module
    (`<module path>#<pack name>[#i]`), the fused motion attention's LayerNorm tables, the concatenated time-embedding projection,
    and the parameters / buffers themselves (state-dict keys)."""
    out = {}

    def add(name, v):
        if torch.is_tensor(v):
            if v.is_cuda and v.numel():
                out[name] = v
        elif isinstance(v, (tuple, list)):
            for i, e in enumerate(v):
                add(f"{name}#{i}", e)

    for mname, m in unet.named_modules():
        packed = getattr(m, "_packed", None)
        if isinstance(packed, dict):
            for k, v in dict.items(packed):
                add(f"{mname}#{k}", v)
        for (site, frames), tab in getattr(m, "_ma_tables", {}).items():
            add(f"{mname}#ma_table{site}_{frames}", tab)
    tp = getattr(unet, "_temb_packed", None)
    if tp is not None:
        add("#temb_proj", tp[:2])
    for k, v in unet.state_dict().items():
        add(k, v)
    return out


def _storage_range(t):
    st = t.untyped_storage()
    return st.data_ptr(), st.data_ptr() + st.nbytes()


def pack_plan(keys, ops_bin, relocs, payload, problem, arena_bytes, io_mask):
    """the blob `i2v_unet_set_plan` takes (csrc/handle.hip PlanHeader / PlanOp / PlanReloc; little-endian, sections 8-byte aligned):
    header, key table (u32 length + bytes, padded to 4), ops (6 x u32: entry, payload offset, payload bytes, first relocation,
    relocation count, struct bytes), relocations (offset, kind, index, pad: 4 x u32; addend: u64), payload"""
    key_tab = bytearray()
    for k in keys:
        kb = k.encode() if isinstance(k, str) else bytes(k)
        key_tab += struct.pack("<I", len(kb)) + kb + b"\0" * (-len(kb) % 4)
    key_tab += b"\0" * (-len(key_tab) % 8)
    hdr_size = struct.calcsize("<6I6iQ2I6Q")
    keys_off = hdr_size
    ops_off = keys_off + len(key_tab)
    relocs_off = ops_off + 24 * len(ops_bin)
    payload_off = relocs_off + 24 * len(relocs)
    total = payload_off + len(payload)
    hdr = struct.pack("<6I6iQ2I6Q", PLAN_MAGIC, PLAN_VERSION, _lib.ABI_VERSION, len(ops_bin), len(keys), len(relocs),
                      *[int(v) for v in problem], arena_bytes, io_mask, 0,
                      keys_off, ops_off, relocs_off, payload_off, len(payload), total)
    blob = bytes(hdr + key_tab + b"".join(ops_bin) + b"".join(relocs) + bytes(payload))
    assert len(blob) == total
    return blob


def pack_op(entry, struct_bytes, slots, relocations, payload, relocs):
    """one launch appended to `payload` / `relocs`; returns its PlanOp record.  slots: 8-byte values (bytes); relocations:
    (offset in the launch's block, kind, index, addend)"""
    blockb = bytearray(struct_bytes) + bytearray(_pad8(len(struct_bytes)) - len(struct_bytes))
    for v in slots:
        blockb += v
    begin = len(relocs)
    for off, kind, index, addend in relocations:
        if off + 8 <= len(blockb):                  # (a relocation outside its block is the parser's business to refuse)
            blockb[off: off + 8] = b"\0" * 8
        relocs.append(struct.pack("<IIIIQ", off, kind, index, 0, addend))
    rec = struct.pack("<IIIIII", entry, len(payload), len(blockb), begin, len(relocs) - begin, len(struct_bytes))
    payload += blockb
    return rec


def record_plan(run, *, unet, io, problem, late_io=None, extra_persistent=None, restore=None):
    """The launches `run()` issues, as a launch plan for `i2v_unet_run` (`UNetHandle.run`).
      io                {slot: tensor}: buffers that exist before the call -- the plan's arguments (read and / or written in place)
      late_io           result -> {slot: tensor}: tensors `run()` allocates and returns that the caller wants as arguments (a forward's
                        output): claimed in the LAST launch only (an earlier, freed activation may have had the address)
      extra_persistent  {name: tensor} (or a callable returning it, evaluated after the call): more buffers that outlive the call and
                        are registered by name like weights (per-sample buffers: `sample#...`)
      restore           called before every invocation of `run()` (it is invoked twice: once to build every lazily packed operand,
                        once recorded): puts in / out arguments back to their initial contents
      problem           (batch, frames, height, width, ctx_len, has_ip) for the blob's header (`i2v_unet_plan` must match)
    Returns (blob, weights) as `record_forward_plan`."""
    dev = next(iter(io.values())).device
    for t in io.values():
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("record_plan: the arguments must be contiguous device tensors")
    with torch.no_grad():
        if restore is not None:
            restore()
        run()                                                  # builds every lazily packed operand
        torch.cuda.synchronize(dev)
        if restore is not None:
            restore()
        real = _lib.load()
        rec = _RecordingLib(real)
        pool = torch.cuda.MemPool()
        saved = _lib._lib
        _lib._lib = rec
        try:
            with torch.cuda.use_mem_pool(pool, device=dev):
                result = run()
        finally:
            _lib._lib = saved
        torch.cuda.synchronize(dev)
    late = dict(late_io(result)) if late_io is not None else {}
    for t in late.values():
        if not t.is_contiguous():
            raise RuntimeError("record_plan: unexpected output layout")
    segments = sorted((s["address"], s["total_size"]) for s in pool.snapshot())
    seg_off, arena_bytes = {}, 0
    for addr, size in segments:
        seg_off[addr] = arena_bytes
        arena_bytes += (size + 255) & ~255
    seg_starts = [a for a, _ in segments]
    # io ranges are the tensors themselves (the C caller passes the tensor's first byte)
    io_ranges = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), s, False) for s, t in io.items()]
    io_ranges += [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), s, True) for s, t in late.items()]
    pers = persistent_tensors(unet)
    extra = extra_persistent() if callable(extra_persistent) else (extra_persistent or {})       # (after the call: it may create some)
    pers.update({k: v for k, v in extra.items() if v is not None and v.is_cuda and v.numel()})
    spans = {}
    # one name per storage.  A tensor the kernels read exactly as the checkpoint holds it (fp16 biases, norm affines, Linear / 1x1
    # weights: `w16` of an fp16 parameter is the parameter) goes by its STATE-DICT key -- a host can register the reference's
    # checkpoint tensor for it directly -- and only the re-laid-out operands by `<module path>#<pack>`
    for name in sorted(pers, key=lambda n: ("#" in n, n)):
        lo, hi = _storage_range(pers[name])
        spans.setdefault((lo, hi), name)
    span_list = sorted((lo, hi, name) for (lo, hi), name in spans.items())
    span_starts = [s[0] for s in span_list]

    keys, key_index, relocs, ops_bin, payload = [], {}, [], [], bytearray()
    io_mask = 0
    for op_i, (entry, sbytes, slots, ptrs) in enumerate(rec.ops):
        last = op_i == len(rec.ops) - 1
        hits = []
        for off, v, what in ptrs:
            hit = next(((RELOC_IO, s, v - lo) for lo, hi, s, is_late in io_ranges if lo <= v < hi and (not is_late or last)), None)
            if hit is None:
                i = bisect.bisect_right(span_starts, v) - 1
                if i >= 0 and v < span_list[i][1]:
                    name = span_list[i][2]
                    if name not in key_index:
                        key_index[name] = len(keys)
                        keys.append(name)
                    hit = (RELOC_WEIGHT, key_index[name], v - pers[name].untyped_storage().data_ptr())
            if hit is None:
                i = bisect.bisect_right(seg_starts, v) - 1
                if i >= 0 and v < segments[i][0] + segments[i][1]:
                    hit = (RELOC_ARENA, 0, seg_off[segments[i][0]] + v - segments[i][0])
            if hit is None:
                raise RuntimeError(f"record_plan: {what} = {v:#x} is neither an argument, a persistent tensor of the model nor "
                                   "memory the call allocated -- a buffer the plan cannot name")
            if hit[0] == RELOC_IO:
                io_mask |= 1 << hit[1]
            hits.append((off,) + hit)
        ops_bin.append(pack_op(entry, sbytes, slots, hits, payload, relocs))
    for s_ in late:
        if not (io_mask >> s_) & 1:
            raise RuntimeError("record_plan: the last launch does not write the call's result")
    blob = pack_plan(keys, ops_bin, relocs, payload, problem, arena_bytes, io_mask)
    return blob, {k: pers[k] for k in keys}


def record_forward_plan(unet, sample, timesteps, encoder_hidden_states, image_embeds=None, enable_cross_frame_attn=True,
                        cfg_shared_prefix=False):
    """One `unet.forward` (unet:1289-1451) as a launch plan.  Arguments as `i2v_unet_forward` takes them: sample fp16 / fp32
    [B, F, C, H, W], timesteps fp32 [B], encoder_hidden_states fp16 [B, L, D], image_embeds fp16 [B, clip] or None -- all contiguous
    on the model's device.  Returns (blob, weights): the plan and {key: tensor} of exactly the persistent tensors it names (register
    each with `UNetHandle.set_weights`, or `save_weights` them for another host).  The plan holds for this problem size, these
    switches (I2V_* environment, `blocks.set_precise_stream`) and this library build only."""
    io = {IO_SAMPLE: sample, IO_TIMESTEPS: timesteps, IO_CONTEXT: encoder_hidden_states}
    if image_embeds is not None:
        io[IO_IMAGE_EMBEDS] = image_embeds
    if sample.dim() != 5 or sample.dtype not in (torch.float16, torch.float32) or timesteps.dtype != torch.float32 or \
            tuple(timesteps.shape) != (sample.shape[0],) or encoder_hidden_states.dtype != torch.float16 or \
            (image_embeds is not None and image_embeds.dtype != torch.float16):
        raise ValueError("record_forward_plan: sample [B, F, C, H, W] fp16 / fp32, timesteps fp32 [B], context / image_embeds fp16")
    kw = dict(added_cond_kwargs={"image_embeds": image_embeds} if image_embeds is not None else None,
              cross_attention_kwargs={"cfg_shared_prefix": True} if cfg_shared_prefix else None)
    b, f, _c, hh, ww = sample.shape

    def late(out):
        if out.dtype != sample.dtype:
            raise RuntimeError("record_forward_plan: unexpected output dtype")
        return {IO_OUT: out}
    return record_plan(lambda: unet(sample, timesteps, enable_cross_frame_attn, encoder_hidden_states, **kw).sample, unet=unet, io=io,
                       late_io=late, problem=(b, f, hh, ww, encoder_hidden_states.shape[1], int(image_embeds is not None)))


# ---- the whole denoising loop (pipe:663-700) for a host without Python: a per-sample preparation plan and a per-step plan whose
#      per-sample buffers are named like weights
STEP_LATENTS, STEP_COND, STEP_INDEX, STEP_COEF = range(4)          # io slots of a step plan
PREP_CONTEXT, PREP_TIMESTEPS, PREP_IMAGE_EMBEDS = range(3)         # io slots of a preparation plan


def sample_buffers(unet, st):
    """name -> tensor of the per-sample buffers a prepared pipeline state holds and its captured step reads (`_run_steps`): the
    time-embedding table of the schedule and every cross-attention layer's K / V^T of the context (+ their fragment form)"""
    out = {"sample#temb_table": st["temb_table"]}
    pc = st["ctx_proj"]
    for i, attn in enumerate(unet._cross_attention_layers()):
        for nm, t in zip(("k", "vt", "k_ip", "vt_ip"), pc.kv[attn]):
            if t is not None:
                out[f"sample#ctx.{i}.{nm}"] = t
        for j, t in enumerate(pc.frag.get(attn) or ()):
            if t is not None:
                out[f"sample#ctx.{i}.frag{j}"] = t
    return out


def _step_problem(st):
    b, f, _c, hh, ww = st["latents"].shape
    return (st["copies"] * b, f, hh, ww, st["ctx_text"].shape[1], int(st["ctx_ip"] is not None))


def record_step_plan(pipe, st):
    """ONE iteration of pipe:666-697 -- `I2VAdapterPipeline._step`: frame-0 overwrite + CFG duplicate (i2v_ddim_prep), the UNet as the
    pipeline routes it (projected context, the step's row of the time-embedding table, the CFG prefix computed once), CFG combine +
    DDIM update (i2v_ddim_cfg_step) -- as a launch plan.  `st` is a PREPARED pipeline state (static buffers, `ctx_proj`, `temb_table`).
    io: STEP_LATENTS fp32 [B, F, C, H, W] (in / out), STEP_COND fp32 [B, C, H, W], STEP_INDEX int32 [1] (in / out: advanced by the
    step), STEP_COEF fp32 [T, 4]; the per-sample buffers are weights named `sample#...` (`sample_buffers`)."""
    keep = (st["latents"].clone(), st["step_idx"].clone())

    def restore():
        st["latents"].copy_(keep[0])
        st["step_idx"].copy_(keep[1])
    blob, weights = record_plan(lambda: pipe._step(st), unet=pipe.unet, restore=restore, problem=_step_problem(st),
                                io={STEP_LATENTS: st["latents"], STEP_COND: st["cond"], STEP_INDEX: st["step_idx"], STEP_COEF: st["coef"]},
                                extra_persistent=lambda: sample_buffers(pipe.unet, st))
    restore()
    return blob, weights


def record_prepare_plan(pipe, st, image_embeds=None):
    """What the pipeline computes once per sample before its steps (`_run_steps`; the reference recomputes it inside every UNet call):
    ImageProjection of the image embeds (unet:1284-1287, 1351-1352), to_k / to_v of the context for the 16 cross-attention layers
    (i2v:527-532, unet:1263-1279) written into the state's `ctx_proj` buffers, and time_proj -> time_embedding -> silu -> the 22
    time_emb_proj of every timestep of the schedule (unet:1336-1343) written into `temb_table` -- as a launch plan.
    io: PREP_CONTEXT fp16 [2 B, L, D] (= st["ctx_text"]), PREP_TIMESTEPS fp32 [T] (= st["t_table"]), PREP_IMAGE_EMBEDS fp16 [2 B, clip]."""
    unet = pipe.unet
    io = {PREP_CONTEXT: st["ctx_text"], PREP_TIMESTEPS: st["t_table"]}
    if image_embeds is not None:
        io[PREP_IMAGE_EMBEDS] = image_embeds

    def run():
        ip = unet._project_image_embeds({"image_embeds": image_embeds}) if image_embeds is not None else None
        unet.project_context(st["ctx_text"], ip, out=st["ctx_proj"])
        unet.project_time_table(st["t_table"], out=st["temb_table"])
    try:
        return record_plan(run, unet=unet, io=io, problem=_step_problem(st), extra_persistent=lambda: sample_buffers(unet, st))
    finally:
        st["ctx_proj"].ip = st["ctx_ip"]          # (not the projection the recorded call allocated from its private pool)


def base_tensor(t):
    """the whole storage of a persistent tensor as a flat tensor of its dtype (what a weight key's relocation offsets are relative to)"""
    st = t.untyped_storage()
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(st, 0, (st.nbytes() // t.element_size(),))


def save_plan(blob, path):
    with open(path, "wb") as f:
        f.write(blob)


def save_weights(weights, path):
    """the plan's persistent tensors for a host without torch: "I2VW", u32 count, then per tensor u32 key length, key, u32 dtype
    (0 fp16 / 1 fp32), u64 bytes, padding to 16, the bytes (whole storage, as `base_tensor`)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<4sI", b"I2VW", len(weights)))
        for k, t in weights.items():
            bt = base_tensor(t).cpu().contiguous()
            kb = k.encode()
            raw = bt.numpy().tobytes()
            f.write(struct.pack("<I", len(kb)) + kb + struct.pack("<IQ", 0 if bt.dtype == torch.float16 else 1, len(raw)))
            f.write(b"\0" * (-f.tell() % 16))
            f.write(raw)


class UNetHandle:
    def __init__(self, unet_or_config, ip_num_tokens: int = 0):
        self._h = C.c_void_p()
        cfg = getattr(unet_or_config, "config", unet_or_config)
        get = (lambda k, d=None: cfg.get(k, d) if isinstance(cfg, dict) else getattr(cfg, k, d))
        c = _lib.UnetConfig()
        c.in_channels, c.out_channels = get("in_channels", 4), get("out_channels", 4)
        for i, v in enumerate(get("block_out_channels", (320, 640, 1280, 1280))):
            c.block_out_channels[i] = v
        c.layers_per_block, c.num_attention_heads = get("layers_per_block", 2), get("num_attention_heads", 8)
        c.cross_attention_dim, c.norm_num_groups = get("cross_attention_dim", 768), get("norm_num_groups", 32)
        c.motion_max_seq_length, c.motion_num_attention_heads = get("motion_max_seq_length", 32), get("motion_num_attention_heads", 8)
        c.use_motion_mid_block, c.ip_num_tokens = int(get("use_motion_mid_block", True)), ip_num_tokens
        self._lib = _lib.load()
        _lib.check(self._lib.i2v_unet_create(C.byref(c), C.byref(self._h)), "i2v_unet_create")
        self._keep = {}          # the registry holds raw pointers: keep the tensors alive on this side

    def close(self):
        if getattr(self, "_h", None):
            self._lib.i2v_unet_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_weights(self, tensors):
        """register every fp16 / fp32 tensor of {key: device tensor} -- a state dict, or the `weights` of `record_forward_plan`
        (registered as the WHOLE storage behind each tensor: a plan's offsets are relative to it)"""
        for k, t in tensors.items():
            if t.dtype not in (torch.float16, torch.float32):
                continue
            t = base_tensor(t.detach())
            shape = (C.c_int64 * 1)(t.numel())
            _lib.check(self._lib.i2v_unet_set_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()),
                                                     0 if t.dtype == torch.float16 else 1, 1, shape), "i2v_unet_set_weight")
            self._keep[k] = t
        return int(self._lib.i2v_unet_num_weights(self._h))

    def weight_ptr(self, key):
        p, dt, nd = C.c_void_p(), C.c_int32(), C.c_int32()
        shape = (C.c_int64 * 4)()
        _lib.check(self._lib.i2v_unet_get_weight(self._h, key.encode(), C.byref(p), C.byref(dt), C.byref(nd), shape), "i2v_unet_get_weight")
        return (p.value, dt.value, tuple(shape[i] for i in range(nd.value))) if p.value else None

    def plan(self, batch, frames, height, width, ctx_len=77, has_ip=False):
        pl = _lib.UnetPlan(batch, frames, height, width, ctx_len, int(has_ip))
        _lib.check(self._lib.i2v_unet_plan(self._h, C.byref(pl)), "i2v_unet_plan")

    def set_plan(self, blob: bytes):
        """install a launch plan (`record_forward_plan`) for the planned problem; returns (launches, weight keys it names)"""
        self._blob = bytes(blob)
        _lib.check(self._lib.i2v_unet_set_plan(self._h, self._blob, len(self._blob)), "i2v_unet_set_plan")
        n = int(self._lib.i2v_unet_plan_num_keys(self._h))
        return int(self._lib.i2v_unet_plan_launches(self._h)), [self._lib.i2v_unet_plan_key(self._h, i).decode() for i in range(n)]

    @property
    def activation_bytes(self):
        return int(self._lib.i2v_unet_activation_bytes(self._h))

    def set_workspace(self, arena):
        _lib.check(self._lib.i2v_unet_set_workspace(self._h, C.c_void_p(arena.data_ptr()) if arena is not None else None,
                                                    arena.numel() * arena.element_size() if arena is not None else 0), "i2v_unet_set_workspace")
        self._arena = arena

    def forward(self, sample, timesteps, context, image_embeds, out, stream=None):
        """unet:1289-1451 through the C ABI alone: one call, the launches are issued by the library from the installed plan"""
        s = stream if stream is not None else torch.cuda.current_stream(sample.device)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(self._lib.i2v_unet_forward(self._h, p(sample), p(timesteps), p(context), p(image_embeds), p(out),
                                              C.c_void_p(s.cuda_stream)), "i2v_unet_forward")
        return out

    def run(self, io, stream=None):
        """the installed plan with {slot: tensor} as its arguments (`i2v_unet_run`): plans of `record_step_plan` / `record_prepare_plan`"""
        dev_t = next(iter(io.values()))
        s = stream if stream is not None else torch.cuda.current_stream(dev_t.device)
        arr = (C.c_void_p * 5)(*[(io[i].data_ptr() if io.get(i) is not None else None) for i in range(5)])
        _lib.check(self._lib.i2v_unet_run(self._h, arr, 5, C.c_void_p(s.cuda_stream)), "i2v_unet_run")

    def capture(self, stream, launch):
        """capture what `launch()` issues on `stream` (a torch.cuda.Stream made current for the call) as the handle's step.
        `launch` must not allocate through torch: `self.forward` and kernel wrappers given their `out=`."""
        s = C.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            _lib.check(self._lib.i2v_unet_capture_step(self._h, s), "i2v_unet_capture_step")
            try:
                launch()
            except BaseException:
                self._lib.i2v_unet_abort_capture(self._h)
                raise
            _lib.check(self._lib.i2v_unet_end_capture(self._h), "i2v_unet_end_capture")

    def replay(self, stream):
        _lib.check(self._lib.i2v_unet_replay_step(self._h, C.c_void_p(stream.cuda_stream)), "i2v_unet_replay_step")

    @property
    def has_step(self):
        return bool(self._lib.i2v_unet_has_step(self._h))


# ---- export for a host without Python: everything tests/c_host/denoise_host.c (or a service of the reader's own) loads
def export_denoiser(pipe, out_dir, *, num_frames, latent_height, latent_width, batch=1, ctx_len=77, clip_dim=None,
                    num_inference_steps=25, guidance_scale=7.5):
    """Writes the reference's denoising loop (pipe:663-700) for ONE problem size as files a C host runs through `i2v_unet_run`:
        prepare.plan   once per sample (`record_prepare_plan`)       step.plan   once per DDIM step (`record_step_plan`)
        weights.bin    every buffer the two plans name (`save_weights`: model weights by state-dict key / `<module>#<pack>`, per-sample
                       buffers `sample#...` -- the preparation overwrites those)
        manifest.json  the problem, the io slots of both plans with their dtypes and shapes, the schedule's tables (timesteps,
                       DDIM coefficients) and the values baked into the step (guidance scale, CFG copies, table length)
    `pipe`: an `I2VAdapterPipeline` on the GPU (its UNet with or without the IP-Adapter: clip_dim = the image embeds' width, None
    without).  Returns the manifest.  A plan holds for this size, these switches and this library build."""
    import json
    import os
    unet = pipe.unet
    dev = unet.device
    sch = pipe.scheduler
    sch.set_timesteps(num_inference_steps)
    ts = sch.timesteps
    B, F, hh, ww = batch, num_frames, latent_height, latent_width
    c_in, d_ctx = unet.config.in_channels, unet.config.cross_attention_dim
    with torch.no_grad():
        ie = torch.zeros(2 * B, clip_dim, dtype=torch.float16, device=dev) if clip_dim else None
        st = dict(latents=torch.zeros(B, F, c_in, hh, ww, device=dev), cond=torch.zeros(B, c_in, hh, ww, device=dev), copies=2,
                  num_frames=F, guidance=float(guidance_scale), t_table=ts.float().to(dev), coef=sch.step_coefficients(ts).to(dev),
                  step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
                  ctx_text=torch.zeros(2 * B, ctx_len, d_ctx, dtype=torch.float16, device=dev),
                  ctx_ip=unet._project_image_embeds({"image_embeds": ie}) if ie is not None else None)
        st["ctx_proj"] = unet.project_context(st["ctx_text"], st["ctx_ip"])
        st["temb_table"] = unet.project_time_table(st["t_table"])
        step_blob, w_step = record_step_plan(pipe, st)
        prep_blob, w_prep = record_prepare_plan(pipe, st, image_embeds=ie)
    os.makedirs(out_dir, exist_ok=True)
    save_plan(prep_blob, os.path.join(out_dir, "prepare.plan"))
    save_plan(step_blob, os.path.join(out_dir, "step.plan"))
    weights = {**w_step, **w_prep}
    save_weights(weights, os.path.join(out_dir, "weights.bin"))
    desc = lambda t: None if t is None else {"dtype": str(t.dtype).replace("torch.", ""), "shape": list(t.shape)}
    cfg = unet.config
    manifest = {
        "abi_version": _lib.ABI_VERSION,
        "unet_config": {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in dict(cfg).items()
                        if isinstance(v, (int, float, str, bool, tuple, list)) or v is None},
        "ip_num_tokens": 4 if ie is not None else 0,
        "problem": dict(zip(("batch", "frames", "height", "width", "ctx_len", "has_ip"), _step_problem(st))),
        "samples": B, "cfg_copies": 2, "guidance_scale": float(guidance_scale), "num_inference_steps": int(num_inference_steps),
        "prepare": {"plan": "prepare.plan", "launches": int.from_bytes(prep_blob[12:16], "little"),
                    "io": {str(PREP_CONTEXT): {"name": "context [negative ; positive] prompt embeds", **desc(st["ctx_text"])},
                           str(PREP_TIMESTEPS): {"name": "timesteps of the schedule", **desc(st["t_table"])},
                           **({str(PREP_IMAGE_EMBEDS): {"name": "image embeds [zeros ; image]", **desc(ie)}} if ie is not None else {})}},
        "step": {"plan": "step.plan", "launches": int.from_bytes(step_blob[12:16], "little"),
                 "io": {str(STEP_LATENTS): {"name": "latents (in / out)", **desc(st["latents"])},
                        str(STEP_COND): {"name": "condition image latents", **desc(st["cond"])},
                        str(STEP_INDEX): {"name": "device-side step counter (in / out; start at 0)", **desc(st["step_idx"])},
                        str(STEP_COEF): {"name": "DDIM coefficients per step", **desc(st["coef"])}}},
        "timesteps": [float(v) for v in st["t_table"].cpu()], "ddim_coefficients": st["coef"].cpu().tolist(),
        "arena_bytes": max(int.from_bytes(prep_blob[48:56], "little"), int.from_bytes(step_blob[48:56], "little")),
        "weights": {"file": "weights.bin", "tensors": len(weights), "state_dict_keys": sum("#" not in k for k in weights),
                    "packs": sum("#" in k and not k.startswith("sample#") for k in weights),
                    "per_sample_buffers": sum(k.startswith("sample#") for k in weights)},
    }
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return manifest


def write_denoise_inputs(path, unet_config, manifest, latents, cond, context, image_embeds=None):
    """the inputs file of tests/c_host/denoise_host.c for an `export_denoiser` manifest: initial latents fp32 [B, F, C, H, W] (after
    the first-frame prior and add_noise, pipe:647-656), condition latents fp32 [B, C, H, W], context fp16 [2 B, L, D] (negative rows
    first), image embeds fp16 [2 B, clip] (zeros first) -- host or device tensors."""
    cfg, pr = unet_config, manifest["problem"]
    get = (lambda k: cfg.get(k) if isinstance(cfg, dict) else getattr(cfg, k))
    clip = 0 if image_embeds is None else image_embeds.shape[1]
    ints = [get("in_channels"), get("out_channels"), *get("block_out_channels"), get("layers_per_block"), get("num_attention_heads"),
            get("cross_attention_dim"), get("norm_num_groups"), get("motion_max_seq_length"), get("motion_num_attention_heads"), 1,
            manifest["ip_num_tokens"], pr["batch"], pr["frames"], pr["height"], pr["width"], pr["ctx_len"], clip,
            len(manifest["timesteps"]), manifest["samples"], 0, 0]
    tt = torch.tensor(manifest["timesteps"], dtype=torch.float32)
    coef = torch.tensor(manifest["ddim_coefficients"], dtype=torch.float32)
    with open(path, "wb") as f:
        f.write(b"I2VD" + struct.pack("<24i", *[int(v) for v in ints]))
        for t, dt in ((latents, torch.float32), (cond, torch.float32), (context, torch.float16), (tt, torch.float32), (coef, torch.float32)) + \
                (((image_embeds, torch.float16),) if image_embeds is not None else ()):
            f.write(t.detach().to("cpu", dt).contiguous().numpy().tobytes())
