"""fp16-emulating mode of the CPU oracle  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's own inference driver (pipe:733-757) calls `from_pretrained` WITHOUT `torch_dtype` and `.to(device)`,
i.e. it runs fp32; its fp16 mode is the trainer's mixed precision (`--mixed_precision fp16`,
src/train_image_to_video.py:306-308, weight_dtype :658-665) and what BASELINE config 2 ("fp16") would be on the
reference's op graph: each torch op takes fp16 tensors, accumulates in fp32 inside the kernel (addmm / convolution /
native_group_norm / native_layer_norm / softmax / flash SDPA) and ROUNDS ITS RESULT TO fp16.  `emulate_reference_fp16()` reproduces that
rounding pattern on the CPU: the oracle's op graph is unchanged and runs in fp32, and a TorchDispatchMode rounds the
floating-point result of every aten op that produces new values to the nearest fp16 (views and copies of already
rounded data are left alone).  Two exceptions, both as in the reference:

  * `Timesteps` (sinusoid of t) is computed in fp32 and cast once (unet:1336-1341: `t_emb.to(dtype=self.dtype)`);
  * attention uses `F.scaled_dot_product_attention` (AttnProcessor2_0): scores and probabilities stay in fp32
    inside the fused kernel and only the result is rounded.  The oracle's explicit softmax form is swapped for
    the fused op while the mode is active (tests/test_oracle.py checks that the two agree to 1e-6 in fp32).

What it is for ("two tolerances"): the distance  |fp16-emulated oracle - fp32 oracle|  is the
error an fp16 run of the REFERENCE's op graph would have against exact arithmetic on the same weights.  It is reported
next to the HIP error as INFORMATION (a yardstick for what fp16 storage costs on this graph); the binding gate of every
parity test is the fixed tolerance against the fp32 oracle.
"""
import contextlib

import torch
import torch.nn.functional as F
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_map

from . import blocks as _blocks

_F16_MAX = 65504.0


def round_fp16(t: torch.Tensor) -> torch.Tensor:
    """nearest-even fp16 rounding of an fp32 tensor, kept in fp32 (overflow saturates to inf like the cast does)."""
    return t.half().float()


def _storages(args):
    out = set()

    def visit(a):
        if isinstance(a, torch.Tensor) and a.numel() > 0:
            out.add(a.untyped_storage().data_ptr())
        elif isinstance(a, (list, tuple)):
            for b in a:
                visit(b)
        elif isinstance(a, dict):
            for b in a.values():
                visit(b)
    visit(args)
    return out


class _RoundEveryOp(TorchDispatchMode):
    """Round the fp32 result of every value-producing aten op to fp16 (result aliasing an input = view: untouched)."""

    def __init__(self):
        super().__init__()
        self.rounded_ops = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        ins = _storages((args, kwargs))

        def fix(t):
            # 0-dim results are scheduler scalars (alphas_cumprod[t] ** 0.5, ...): fp32 in the reference too
            if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() == 0 or t.numel() == 0:
                return t
            if t.untyped_storage().data_ptr() in ins:
                # a view of an input, or an in-place op: round in place only for the latter
                name = func.__name__ if hasattr(func, "__name__") else str(func)
                if name.split(".")[0].endswith("_"):
                    t.copy_(round_fp16(t))
                    self.rounded_ops += 1
                return t
            self.rounded_ops += 1
            return round_fp16(t)
        return tree_map(fix, out)


def _sdpa_fused(self, q, k, v):
    return F.scaled_dot_product_attention(q, k, v, scale=self.scale)


@contextlib.contextmanager
def emulate_reference_fp16():
    """Context manager: every oracle forward inside it rounds like the reference's fp16 GPU path."""
    mode = _RoundEveryOp()
    orig_ts_forward = _blocks.Timesteps.forward
    orig_sdpa = _blocks.Attention._sdpa

    def ts_forward(self, timesteps):
        with torch.utils._python_dispatch._disable_current_modes():
            e = orig_ts_forward(self, timesteps)
        return round_fp16(e)

    _blocks.Timesteps.forward = ts_forward
    _blocks.Attention._sdpa = _sdpa_fused
    try:
        with mode:
            yield mode
    finally:
        _blocks.Timesteps.forward = orig_ts_forward
        _blocks.Attention._sdpa = orig_sdpa
