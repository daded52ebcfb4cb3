"""The library's model handle (include/i2v_hip.h "Model handle") seen from Python: what a host that is NOT this package's
module mirror binds -- configuration, the weight registry under the reference's state-dict keys, the plan of one denoising step
and one captured step (pipe:96, 676-683).  The package's own pipeline keeps torch's graph capture (its step allocates through
torch's allocator, which must know about the capture); this wrapper exists for hosts with their own buffers, and for the tests."""
import ctypes as C

import torch

from . import _lib


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

    def set_weights(self, state_dict):
        """register every fp16 / fp32 tensor of a (device) state dict under its key (unet.state_dict() names)"""
        for k, t in state_dict.items():
            if t.dtype not in (torch.float16, torch.float32) or t.dim() > 4:
                continue
            t = t.detach().contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check(self._lib.i2v_unet_set_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()),
                                                     0 if t.dtype == torch.float16 else 1, t.dim(), shape), "i2v_unet_set_weight")
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
        return int(self._lib.i2v_unet_activation_bytes(self._h))

    def capture(self, stream, launch):
        """capture what `launch()` issues on `stream` (a torch.cuda.Stream made current for the call) as the handle's step.
        `launch` must not allocate through torch (pass every kernel wrapper its `out=`)."""
        s = C.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            _lib.check(self._lib.i2v_unet_capture_step(self._h, s), "i2v_unet_capture_step")
            try:
                launch()
            finally:
                rc = self._lib.i2v_unet_end_capture(self._h)
            _lib.check(rc, "i2v_unet_end_capture")

    def replay(self, stream):
        _lib.check(self._lib.i2v_unet_replay_step(self._h, C.c_void_p(stream.cuda_stream)), "i2v_unet_replay_step")

    @property
    def has_step(self):
        return bool(self._lib.i2v_unet_has_step(self._h))
