"""Host-side building blocks of the I2V-Adapter path on the HIP kernels.

These classes mirror the `diffusers` modules the reference constructs: same names, ctor
kwargs, attribute names and state-dict keys, so reference checkpoints load by key.  torch.nn modules are used
ONLY as parameter containers; every `forward` runs the hand-written HIP kernels through
`i2v_adapter_unofficial_amd.kernels` (C ABI).  Internally activations are token-major fp16:
images are [N, H, W, C] tensors, token matrices are [rows, C]; the public `forward`s accept / return the
reference's NCHW tensors and convert at the edge.
"""
import math
import os
from typing import Optional

import torch
from torch import nn

from . import kernels as K
from . import streams
from ._lib import (I2V_EPI_GEGLU, I2V_EPI_GELU, I2V_EPI_NONE, I2V_STORE_ROWPERM, HipLibraryError)
from .checkpoint import PretrainedMixin

f16 = torch.float16


# ----------------------------------------------------------------------------------------------------------
class HipModule(nn.Module):
    """Base class: lazily (re)builds kernel-layout fp16 copies of the parameters (`_pack`) whenever a
    parameter was moved, cast or overwritten (data_ptr / version check)."""

    def __init__(self):
        super().__init__()
        self._packed = None
        self._packed_key = None

    def _pack(self):
        return {}

    def __getstate__(self):
        # the kernel-layout packs are a cache (rebuilt on first use) and hold thunks: neither is pickled with the module
        state = self.__dict__.copy()
        state["_packed"], state["_packed_key"] = None, None
        return state

    def packed(self):
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in self.parameters())
        key += tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._packed is None or key != self._packed_key:
            for p in self.parameters():
                if not p.is_cuda:
                    raise HipLibraryError(
                        f"{type(self).__name__} has parameters on {p.device}: the HIP path has no CPU fallback")
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed


def w16(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(f16).contiguous()


class LazyPack(dict):
    """the dict a `_pack` returns, with entries that are built on FIRST USE: `p.lazy(name, fn)` / `p.lazy_group(names, fn)`.
    The operands of the one-launch fused kernels (fragment-ordered weights, fp32 LayerNorm tables) are such entries: an
    inference forward builds them once, while a training step -- whose optimiser writes the adapter's parameters in place, so
    that the block's pack is rebuilt every step, and whose trainers never call a fused kernel -- does not pay their index
    gathers and allocations."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def lazy(self, name, fn):
        self._lazy[name] = (fn, None)

    def lazy_group(self, names, fn):
        """fn() returns one value per name (operands that are made together)"""
        for i, n in enumerate(names):
            self._lazy[n] = (fn, (tuple(names), i))

    def __missing__(self, key):
        if key not in self._lazy:
            raise KeyError(key)
        fn, grp = self._lazy[key]
        with torch.no_grad():
            val = fn()
        if grp is None:
            del self._lazy[key]
            self[key] = val
        else:
            for n, v in zip(grp[0], val):
                self._lazy.pop(n, None)
                dict.__setitem__(self, n, v)
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default


def pack_conv3x3(weight: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout, 9 * Cin'] (Cin zero-padded to Cin') in the contraction order the conv kernel walks:
    tap-major k = (ky * 3 + kx) * Cin' + ci, or, when Cin' % 64 == 0 (`K.conv_k_block`), channel-block-major
    k = ((ci // 64) * 9 + ky * 3 + kx) * 64 + ci % 64."""
    co, ci = weight.shape[:2]
    w = weight.detach().permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad != ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
        ci = cin_pad
    if K.conv_k_block(ci):
        w = w.reshape(co, 9, ci // 64, 64).permute(0, 2, 1, 3)          # [co, channel block, tap, 64]
    return w.reshape(co, 9 * ci).to(f16).contiguous()


def pack_geglu(weight, bias):
    """rows (value_i, gate_i) interleaved so that the gate pair lands in one lane of the GEMM epilogue."""
    inner = weight.shape[0] // 2
    w = torch.stack([weight[:inner], weight[inner:]], dim=1).reshape(2 * inner, weight.shape[1])
    b = torch.stack([bias[:inner], bias[inner:]], dim=1).reshape(2 * inner)
    return w16(w), w16(b)


# LayerNorm folded into the consuming GEMMs (i2v_gemm_params.ln_wsum); I2V_LN_FOLD=0 keeps the materialised LayerNorm
# everywhere (same-box A/B)
LN_FOLD = os.environ.get("I2V_LN_FOLD", "1") != "0"


def fold_layernorm(weight, bias, gamma, beta):
    """LayerNorm(gamma, beta) followed by Linear(weight, bias), as the operands of a LayerNorm-folded GEMM:
        (LN(x) W^T + b)[m][n] = rstd_m (x W'^T - mean_m wsum)[m][n] + b'[n]
    with W' = W o gamma (fp16), wsum = row sums of the fp16-ROUNDED W' (fp32: the mean term must cancel against what
    the MFMA actually multiplies), b' = W beta + b (fp16)."""
    wf = (weight.detach().float() * gamma.detach().float()[None, :]).to(f16).contiguous()
    wsum = wf.float().sum(dim=1).contiguous()
    b = weight.detach().float() @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    return wf, wsum, b.to(f16).contiguous()


def fold_layernorm_geglu(proj_weight, proj_bias, gamma, beta):
    """the same for a GEGLU projection, rows interleaved (value_i, gate_i) like pack_geglu."""
    wf, wsum, b = fold_layernorm(proj_weight, proj_bias, gamma, beta)
    inner = wf.shape[0] // 2
    il = lambda t: torch.stack([t[:inner], t[inner:]], dim=1).reshape((2 * inner,) + tuple(t.shape[1:])).contiguous()
    return il(wf), il(wsum), il(b)


class LnFoldPlan:
    """per-module cache of `does the library fold the LayerNorm for this problem?` answers (i2v_gemm_ln_supported),
    keyed by the problem shape: the fold is used at a LayerNorm site only when EVERY consumer of that site supports it."""

    def __init__(self):
        self._cache = {}

    def get(self, key, probe):
        if not LN_FOLD:
            return False
        if key not in self._cache:
            self._cache[key] = bool(probe())
        return self._cache[key]


class ProjectedContext:
    """Text (+ IP-Adapter image) context with its per-layer K / V^T projections already computed.

    `to_k(ctx)` / `to_v(ctx)` of the 16 cross-attention layers (i2v:527-532, unet:1263-1279) depend only on the prompt
    (and image) embeddings and the weights, not on the latents or the timestep: the reference recomputes them in every
    UNet call of every denoising step; here they are computed once per sample (`UNetMotionCrossFrameAttnModel.
    project_context`) and the captured step only reads them (-32 ... -64 small GEMM launches per step).  Passed down the
    blocks in place of the raw context tensor."""

    def __init__(self, text, ip):
        self.text, self.ip = text, ip
        self.kv = {}                       # Attention module -> (k, vt, k_ip, vt_ip)
        self.frag = {}                     # Attention module -> K / V as MFMA fragments (K.pack_ctx_fragments), made on first use

    @property
    def shape(self):
        return self.text.shape


class ProjectedTemb:
    """`time_emb_proj(silu(temb))` of every ResnetBlock2D as column slices of ONE GEMM over the concatenated
    weights: 22 launches of an M = 2 GEMM per step become one."""

    def __init__(self, all_proj, slices):
        self.all_proj, self.slices = all_proj, slices      # [B, sum Cout] fp16; {resnet: (offset, Cout)}

    def view_for(self, resnet):
        off, n = self.slices[resnet]
        return self.all_proj[:, off: off + n]

    def first_half(self):
        """the rows of the first CFG half of the batch (unet._fwd_tokens, cfg_shared): one row per batch entry -> the first
        half of them; a single row (the pipeline's step: one timestep for the whole batch) serves either half"""
        r = self.all_proj.shape[0]
        return self if r == 1 else ProjectedTemb(self.all_proj[: r // 2], self.slices)


GN_FOLD = os.environ.get("I2V_GN_FOLD", "1") != "0"
GN_FOLD_RATIO = float(os.environ.get("I2V_GN_FOLD_RATIO", "4"))   # fold when activations >= ratio x the weight stack


def gn_proj_in(x, gamma, beta, groups, eps, w, bias, frames=0):
    """proj_in(GroupNorm(x)) at the entry of a spatial transformer (i2v:218-226) or, with frames > 0, of a motion module
    (statistics over the clip's frames, rows re-ordered to (b, pixel, frame)).  x: [N, H, W, C] tokens.

    Where it pays -- the per-group weight stack is small beside the activations -- the norm is FOLDED into the GEMM: its
    per-image (per-clip) scale multiplies the weights, its shift becomes a per-image bias (`K.groupnorm_fold`), and the
    GEMM reads the un-normalised x directly (for the motion module through the permuted row gather).  The normalised
    tensor is neither written nor read back: one pass over x for the statistics instead of two passes + one write."""
    n, hh, ww, c = x.shape
    hw = hh * ww
    fps = frames if frames > 0 else 1
    s_groups, n_out = n // fps, w.shape[0]
    if GN_FOLD and GN_FOLD_RATIO * s_groups * n_out <= n * hw:   # weight stack small beside the activation bytes
        kw = dict(rows_per_vec=fps * hw, w_rows=fps * hw, a_perm=(frames, hw) if frames > 0 else None)
        if _fold_supported(x, s_groups, n_out, kw):
            w_s, b_s = K.groupnorm_fold(x, gamma, beta, groups, eps, w, bias, frames_per_stat=fps)
            return K.gemm(x.view(-1, c), w_s, None, rowvec=b_s, **kw)
    if frames > 0:
        h = K.groupnorm(x, gamma, beta, groups, eps, frames_per_stat=frames, out_perm=True, frames=frames)
    else:
        h = K.groupnorm(x, gamma, beta, groups, eps).view(-1, c)
    return K.gemm(h, w, bias)


_FOLD_OK = {}


def _fold_supported(x, s_groups, n_out, kw):
    """does the library implement the batched-weight (+ permuted gather) GEMM for this shape?  (asked once per shape)"""
    key = (tuple(x.shape), s_groups, n_out, kw["w_rows"], kw["a_perm"])
    ok = _FOLD_OK.get(key)
    if ok is None:
        c = x.shape[3]
        w_s = torch.empty((s_groups, n_out, c), dtype=f16, device=x.device)
        b_s = torch.empty((s_groups, n_out), dtype=f16, device=x.device)
        ok = _FOLD_OK[key] = K.gemm(x.view(-1, c), w_s, None, rowvec=b_s, query_batch_support=True, **kw)
    return ok


def to_tokens(x: torch.Tensor, c_pad: Optional[int] = None) -> torch.Tensor:
    """reference NCHW tensor -> token-major fp16 [N, H, W, C]."""
    return K.nchw_to_tokens(x, c_pad)


def from_tokens(x: torch.Tensor, dtype) -> torch.Tensor:
    return K.tokens_to_nchw(x, dtype=dtype if dtype in (torch.float32, f16) else f16)


def _as_f16_matrix(t: torch.Tensor) -> torch.Tensor:
    """user-facing 2-D / 3-D tensor (fp16 / fp32) -> contiguous fp16 (dtype cast is plumbing at the API edge)."""
    if not t.is_cuda:
        raise HipLibraryError(f"tensor on {t.device}: the HIP path has no CPU fallback")
    return t.to(f16).contiguous()


# ---------------------------------------------------------------------------------------------------------- A1
class Timesteps(nn.Module):
    """A1 (unet:763).  No parameters."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0):
        super().__init__()
        if not flip_sin_to_cos or downscale_freq_shift != 0:
            raise NotImplementedError("hot path uses Timesteps(C, flip_sin_to_cos=True, freq_shift=0)")
        self.num_channels = num_channels

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        return K.timestep_embedding(timesteps.to(torch.float32).contiguous(), self.num_channels)


class TimestepEmbedding(HipModule):
    """A1 (unet:766-770)."""

    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def _pack(self):
        return dict(w1=w16(self.linear_1.weight), b1=w16(self.linear_1.bias), w2=w16(self.linear_2.weight),
                    b2=w16(self.linear_2.bias))

    def forward(self, sample, condition=None):
        p = self.packed()
        h = K.silu(K.gemm(_as_f16_matrix(sample), p["w1"], p["b1"]))
        return K.gemm(h, p["w2"], p["b2"])


# ---------------------------------------------------------------------------------------------------------- precise stream
# The residual stream BETWEEN modules (outputs of conv_in, every ResnetBlock2D, spatial transformer, motion module and sampler) as
# an fp16 pair hi + lo (kernels.lo_of; i2v_gemm_params.residual_lo / c_lo): the identity path of every module's residual add is
# then exact to 2^-22 instead of re-rounding the stream to fp16 at each of the 66 modules, which alone is 1.36e-3 max-abs /
# 0.78 of the 0.89e-3 relative rms of the config-2 forward (pipe:666-697 runs the reference's CPU path in fp32).
# What a module's branch reads as an MFMA / norm operand stays the fp16 high half.  OFF by default: +4 bytes of traffic per stream
# element and module; `set_precise_stream(True)` / I2V_STREAM_PRECISE=1 (a captured hipGraph keeps the mode it was captured in).
_PRECISE_STREAM = os.environ.get("I2V_STREAM_PRECISE", "0") != "0"


def precise_stream() -> bool:
    return _PRECISE_STREAM


def set_precise_stream(on: bool) -> bool:
    """switch the precise residual stream on / off for forwards issued from now on; returns the previous setting"""
    global _PRECISE_STREAM
    prev, _PRECISE_STREAM = _PRECISE_STREAM, bool(on)
    return prev


# ---------------------------------------------------------------------------------------------------------- A2/A3
# norm2's GroupNorm statistics written by conv1's epilogue (i2v_gemm_params.gn_partial) instead of a pass over conv1's output
GN_FROM_CONV = os.environ.get("I2V_GN_FROM_CONV", "1") != "0"


class ResnetBlock2D(HipModule):
    """A2.  GN+SiLU -> conv3x3 (+bias +time-embedding row add fused) -> GN+SiLU -> conv3x3 (+bias + shortcut /
    residual fused).  The skip concat of the up blocks (unet:478) is never materialised: GroupNorm reads both
    sources and the 1x1 shortcut GEMM takes two K ranges."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6, groups=32,
                 output_scale_factor=1.0, **_unused):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.groups, self.eps = groups, eps
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = (nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0)
                              if in_channels != out_channels else None)

    def _pack(self):
        p = dict(g1=w16(self.norm1.weight), b1=w16(self.norm1.bias), w1=pack_conv3x3(self.conv1.weight),
                 cb1=w16(self.conv1.bias), g2=w16(self.norm2.weight), b2=w16(self.norm2.bias),
                 w2=pack_conv3x3(self.conv2.weight), cb2=w16(self.conv2.bias))
        if self.time_emb_proj is not None:
            p["wt"], p["bt"] = w16(self.time_emb_proj.weight), w16(self.time_emb_proj.bias)
        if self.conv_shortcut is not None:
            p["ws"] = w16(self.conv_shortcut.weight.reshape(self.out_channels, self.in_channels))
            p["bs"] = w16(self.conv_shortcut.bias)
        return p

    def _fwd(self, x, temb_act, x2=None):
        """x [N, H, W, C1] (+ x2 [N, H, W, C2] = channel concat); temb_act = silu(temb) [R, Tc] with N % R == 0."""
        p = self.packed()
        n, hh, ww, c1 = x.shape
        cin = c1 + (x2.shape[3] if x2 is not None else 0)
        if cin != self.in_channels:
            raise ValueError(f"ResnetBlock2D expects {self.in_channels} input channels, got {cin}")
        if self.conv_shortcut is None and x2 is not None:
            raise ValueError("concat input needs a conv_shortcut")
        s = x
        # the 1x1 shortcut depends only on the block's input: at the levels where one launch cannot fill the chip it runs
        # on a side stream beside GroupNorm -> conv1 -> GroupNorm (streams.fork), joined before conv2 adds it
        with streams.fork(self.conv_shortcut is not None and n * hh * ww <= streams.MAX_ROWS, x.device) as fk:
            if self.conv_shortcut is not None:
                with fk.side():
                    a2 = None if x2 is None else x2.view(-1, x2.shape[3])
                    s = K.sview(K.gemm(x.view(-1, c1), p["ws"], p["bs"], a2=a2, precise=precise_stream()), n, hh, ww, self.out_channels)
            h = K.groupnorm(x, p["g1"], p["b1"], self.groups, self.eps, x2=x2, silu=True)
            rowvec, rpv = None, 0
            if self.time_emb_proj is not None and temb_act is not None:
                rowvec = (temb_act.view_for(self) if isinstance(temb_act, ProjectedTemb)
                          else K.gemm(temb_act, p["wt"], p["bt"]))
                rpv = (n // rowvec.shape[0]) * hh * ww
            # norm2's statistics come out of conv1's epilogue where that form exists (the 64^2 / 32^2 levels): no statistics pass
            if GN_FROM_CONV:
                h, st = K.conv3x3(h, p["w1"], p["cb1"], rowvec=rowvec, rows_per_vec=rpv, gn_stats_groups=self.groups)
            else:
                h, st = K.conv3x3(h, p["w1"], p["cb1"], rowvec=rowvec, rows_per_vec=rpv), None
            h = K.groupnorm(h, p["g2"], p["b2"], self.groups, self.eps, silu=True, stats=st)
        return K.conv3x3(h, p["w2"], p["cb2"], residual=s, out_scale=1.0 / self.output_scale_factor, precise=precise_stream())

    def forward(self, input_tensor, temb, scale: float = 1.0):
        x = to_tokens(input_tensor)
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        return from_tokens(self._fwd(x, ta), input_tensor.dtype)


class Downsample2D(HipModule):
    """A3 (unet:250-259): conv 3x3, stride 2, padding 1.  padding = 0 is the VAE encoder's form: one zero row / column
    appended at the bottom / right, then the stride-2 conv without padding (the `asym_pad` gather of the conv kernel)."""

    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        if not use_conv or padding not in (0, 1):
            raise NotImplementedError("Downsample2D(use_conv=True, padding in {0, 1}) only")
        out_channels = out_channels or channels
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels, 3, stride=2, padding=padding)

    def _pack(self):
        return dict(w=pack_conv3x3(self.conv.weight), b=w16(self.conv.bias))

    def _fwd(self, x):
        p = self.packed()
        return K.conv3x3(x, p["w"], p["b"], stride=2, asym_pad=self.padding == 0, precise=precise_stream())

    def forward(self, hidden_states, scale: float = 1.0):
        return from_tokens(self._fwd(to_tokens(hidden_states)), hidden_states.dtype)


class Upsample2D(HipModule):
    """A3 (unet:431-432): nearest x2 folded into the conv's input gather (no upsampled tensor is written)."""

    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        if not use_conv:
            raise NotImplementedError("hot path uses Upsample2D(use_conv=True)")
        out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, out_channels, 3, padding=1)

    def _pack(self):
        return dict(w=pack_conv3x3(self.conv.weight), b=w16(self.conv.bias))

    def _fwd(self, x, output_size=None):
        # output_size: the skip tensor's size when the latent size is not a multiple of 8 (unet:1304-1311, 1414-1415): 2x or 2x - 1
        p = self.packed()
        return K.conv3x3(x, p["w"], p["b"], upsample=True,          # (read by conv_shortcut resnets only: no low half needed)
                         output_size=None if output_size is None else tuple(int(v) for v in output_size))

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        return from_tokens(self._fwd(to_tokens(hidden_states), output_size), hidden_states.dtype)


# ---------------------------------------------------------------------------------------------------------- A4/A5
class Attention(HipModule):
    """A4 parameter container (+A5 IP-Adapter branch) with a stand-alone forward.

    The transformer blocks do not call this forward: they fuse the projections of several Attention modules
    (e.g. attn1.to_q | attn1.to_k | i2v_adapter.to_q in one GEMM).  The stand-alone forward is the generic
    form: q / k GEMMs, V^T emitted by the GEMM epilogue, flash attention, out-projection."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True,
                 dropout=0.0, upcast_attention=False):
        super().__init__()
        if bias:
            raise NotImplementedError("attention_bias=True is not on the hot path")
        self.inner_dim = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.query_dim = query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=False)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.ip_num_tokens, self.ip_scale = 0, 1.0
        self.to_k_ip = None
        self.to_v_ip = None

    def install_ip_adapter(self, to_k_ip_weight, to_v_ip_weight, num_tokens=4, scale=1.0):
        """A5 (unet:1264-1279)."""
        dev, dt = self.to_q.weight.device, self.to_q.weight.dtype
        self.to_k_ip = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=False).to(device=dev, dtype=dt)
        self.to_v_ip = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=False).to(device=dev, dtype=dt)
        with torch.no_grad():
            self.to_k_ip.weight.copy_(to_k_ip_weight)
            self.to_v_ip.weight.copy_(to_v_ip_weight)
        self.ip_num_tokens, self.ip_scale = num_tokens, scale

    def _pack(self):
        ob = self.to_out[0].bias
        p = dict(wq=w16(self.to_q.weight), wk=w16(self.to_k.weight), wv=w16(self.to_v.weight),
                 wo=w16(self.to_out[0].weight), bo=None if ob is None else w16(ob))
        if self.to_k_ip is not None:
            p["wk_ip"], p["wv_ip"] = w16(self.to_k_ip.weight), w16(self.to_v_ip.weight)
        return p

    def project_kv(self, ctx_text, ctx_ip, out=None):
        """(k, vt, k_ip, vt_ip) of context tensors [Bc, L, D]; `out` = a previous result to overwrite in place."""
        p = self.packed()
        bc, lt, dc = ctx_text.shape
        o = out if out is not None else (None, None, None, None)
        k = K.gemm(ctx_text.view(-1, dc), p["wk"], out=o[0])
        vt = K.project_vt(ctx_text.view(-1, dc), p["wv"], lt, out=o[1])
        kip = vtip = None
        if ctx_ip is not None and self.ip_num_tokens:
            kip = K.gemm(ctx_ip.view(-1, dc), p["wk_ip"], out=o[2])
            vtip = K.project_vt(ctx_ip.view(-1, dc), p["wv_ip"], ctx_ip.shape[1], out=o[3])
        return k, vt, kip, vtip

    def context_kv(self, ctx_text, ctx_ip):
        """(k, vt, k_ip, vt_ip, text length, image-token length) of a context [Bc, L, D] or a ProjectedContext."""
        if isinstance(ctx_text, ProjectedContext):
            k, vt, kip, vtip = ctx_text.kv[self]
            lt = ctx_text.text.shape[1]
            li = ctx_text.ip.shape[1] if ctx_text.ip is not None else 0
        else:
            k, vt, kip, vtip = self.project_kv(ctx_text, ctx_ip)
            lt = ctx_text.shape[1]
            li = ctx_ip.shape[1] if ctx_ip is not None else 0
        return k, vt, kip, vtip, lt, li

    def context_fragments(self, ctx_text, kv):
        """(text fragments, image-token fragments or None): this layer's K / V of the context as the fragments
        i2v_cross_attn_fused_f16 keeps in registers -- cached on a ProjectedContext (once per prompt; `project_context(..., out=)`
        rewrites them in place), packed per call otherwise."""
        k, vt, kip, vtip, lt, li = kv
        use_ip = kip is not None and bool(self.ip_num_tokens)
        if not isinstance(ctx_text, ProjectedContext):
            return K.pack_ctx_fragments(k, vt, self.heads, lt), (K.pack_ctx_fragments(kip, vtip, self.heads, li) if use_ip else None)
        hit = ctx_text.frag.get(self)
        if hit is None:
            hit = ctx_text.frag[self] = (K.pack_ctx_fragments(k, vt, self.heads, lt),
                                         K.pack_ctx_fragments(kip, vtip, self.heads, li) if use_ip else None)
        return hit

    def refresh_context_fragments(self, pc):
        """after `project_kv(..., out=...)` rewrote this layer's K / V^T of a ProjectedContext in place (the next prompt of a
        captured hipGraph): bring the fragments made from them up to date in THEIR buffers -- the graph reads that memory and no
        Python runs between its replays.  Unconditionally: the library's kernels write through raw pointers, no tensor version
        counter moves."""
        hit = pc.frag.get(self)
        if hit is not None:
            k, vt, kip, vtip, lt, li = self.context_kv(pc, None)
            K.pack_ctx_fragments(k, vt, self.heads, lt, out=hit[0])
            if hit[1] is not None:
                K.pack_ctx_fragments(kip, vtip, self.heads, li, out=hit[1])

    def _cross(self, q, ctx_text, ctx_ip, batch_q, lq, kv_group, kv=None):
        """softmax(q Kt^T) Vt (+ ip_scale * softmax(q Kip^T) Vip) for context tensors [Bc, L, D] (or a
        ProjectedContext holding this layer's K / V^T); kv = a `context_kv` result already at hand."""
        k, vt, kip, vtip, lt, li = kv if kv is not None else self.context_kv(ctx_text, ctx_ip)
        o = K.attention(q, k, vt, batch_q=batch_q, lq=lq, lk=lt, heads=self.heads, head_dim=self.dim_head,
                        kv_group=kv_group, scale=self.scale)
        if kip is not None and self.ip_num_tokens:
            K.attention(q, kip, vtip, batch_q=batch_q, lq=lq, lk=li, heads=self.heads, head_dim=self.dim_head,
                        kv_group=kv_group, scale=self.scale, out=o, accumulate=True, acc_scale=self.ip_scale)
        return o

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        p = self.packed()
        x = _as_f16_matrix(hidden_states)
        b, l, c = x.shape
        q = K.gemm(x.view(-1, c), p["wq"])
        ctx = x if encoder_hidden_states is None else _as_f16_matrix(encoder_hidden_states)
        ctx_ip = None
        if self.ip_num_tokens and encoder_hidden_states is not None:
            end = ctx.shape[1] - self.ip_num_tokens
            ctx_text = torch.empty((ctx.shape[0], end, ctx.shape[2]), dtype=f16, device=x.device)
            ctx_ip = torch.empty((ctx.shape[0], self.ip_num_tokens, ctx.shape[2]), dtype=f16, device=x.device)
            K.copy3d(ctx[:, :end], ctx_text)
            K.copy3d(ctx[:, end:], ctx_ip)
            ctx = ctx_text
        if b % ctx.shape[0] != 0:
            raise ValueError(f"context batch {ctx.shape[0]} does not divide query batch {b}")
        o = self._cross(q, ctx, ctx_ip, b, l, b // ctx.shape[0])
        out = K.gemm(o, p["wo"], p["bo"])
        return out.view(b, l, -1).to(hidden_states.dtype)


# ---------------------------------------------------------------------------------------------------------- A7
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        if approximate != "none":
            raise NotImplementedError("only erf GELU is implemented")
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)


def ff_tail_operands(weight, bias, c_in, c_out):
    """`pack_ff_tail` of a block's closing Linear where the fused feed-forward can take it as its tail (a square Linear of the
    fused width), else None."""
    if c_in != c_out or not K.ff_fused_tail_supported(128, c_in, 4 * c_in):
        return None
    return K.pack_ff_tail(weight.detach().reshape(c_out, c_in), bias)


class FeedForward(HipModule):
    """A7: Linear(C -> 8C) with the GEGLU gate fused into the GEMM epilogue, then Linear(4C -> C) with the
    residual add fused."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", inner_dim=None, bias=True):
        super().__init__()
        if not bias:
            raise NotImplementedError("ff_bias=False is not on the hot path")
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        self.activation_fn = activation_fn
        if activation_fn == "geglu":
            act = GEGLU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        else:
            raise ValueError(f"unsupported activation_fn {activation_fn}")
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)])

    def _pack(self):
        proj = self.net[0].proj
        if self.activation_fn == "geglu":
            w1, b1 = pack_geglu(proj.weight.detach(), proj.bias.detach())
        else:
            w1, b1 = w16(proj.weight), w16(proj.bias)
        out = LazyPack(w1=w1, b1=b1, w2=w16(self.net[2].weight), b2=w16(self.net[2].bias))
        if self.activation_fn == "geglu" and K.ff_fused_supported(128, self.net[2].weight.shape[0], self.net[2].weight.shape[1]):
            # operands of the one-launch form (i2v_ff_fused_f16: the SD-1.5 64^2 width), built on first use
            # (thunks capture the child modules, never `self`: module -> _packed -> thunk -> module would be a reference cycle that
            #  keeps the fp16 packs alive until a cyclic gc pass after `del unet`)
            out.lazy("fused", lambda proj=proj, lin=self.net[2]: K.pack_ff_fused(proj.weight, proj.bias, lin.weight, lin.bias))
        return out

    def _fwd(self, n2d, residual2d, **store):
        p = self.packed()
        epi = I2V_EPI_GEGLU if self.activation_fn == "geglu" else I2V_EPI_GELU
        h = K.gemm(n2d, p["w1"], p["b1"], epilogue=epi)
        return K.gemm(h, p["w2"], p["b2"], residual=residual2d, **store)

    def fused_supported(self, x2d):
        """does the one-launch LayerNorm + GEGLU feed-forward + residual kernel take this problem?"""
        p = self.packed()
        return FUSED_FF and "fused" in p and K.ff_fused_supported(x2d.shape[0], x2d.shape[1], p["b1"].numel() // 2)

    def _fwd_fused(self, x2d, gamma32, beta32, eps, tail=None):
        """x + FF(LayerNorm(x)) in one launch: the inner activation never leaves the CU.  `tail` (see `tail_supported`): the
        Linear that follows the block -- the spatial transformer's / the motion module's proj_out with its residual -- in the
        same launch."""
        return K.ff_fused(x2d, gamma32, beta32, self.packed()["fused"], eps=eps, tail=tail, precise=precise_stream())

    def tail_supported(self, x2d, tail):
        """tail = (pack_ff_tail(w, b), residual rows in output order, perm_frames, perm_hw): can the fused launch take it?"""
        p = self.packed()
        return (FUSED_FF and FUSED_FF_TAIL and tail is not None and tail[0] is not None and "fused" in p and
                K.ff_fused_tail_supported(x2d.shape[0], x2d.shape[1], p["b1"].numel() // 2, tail[2], tail[3]))

    def fold_norm(self, norm):
        """(W', wsum, b') of `norm` (LayerNorm) followed by the first projection, or None when the activation has no
        LayerNorm-folded epilogue."""
        if self.activation_fn != "geglu":
            return None
        proj = self.net[0].proj
        return fold_layernorm_geglu(proj.weight, proj.bias, norm.weight, norm.bias)

    def _fwd_folded(self, x2d, eps, folded, **store):
        """x + FF(LayerNorm(x)) with the LayerNorm folded into the GEGLU projection (x2d is the UN-normalised input)."""
        p = self.packed()
        wf, wsum, bf = folded
        h = K.gemm(x2d, wf, bf, epilogue=I2V_EPI_GEGLU, ln=(wsum, eps))
        return K.gemm(h, p["w2"], p["b2"], residual=x2d, **store)

    def folded_supported(self, x2d, eps, folded):
        wf, wsum, bf = folded
        return K.gemm(x2d, wf, bf, epilogue=I2V_EPI_GEGLU, ln=(wsum, eps), query_ln_support=True)

    def forward(self, hidden_states, scale: float = 1.0):
        x = _as_f16_matrix(hidden_states)
        shp = x.shape
        return self._fwd(x.view(-1, shp[-1]), None).view(*shp[:-1], -1).to(hidden_states.dtype)


# ---------------------------------------------------------------------------------------------------------- A10
class SinusoidalPositionalEmbedding(nn.Module):
    """A10: buffer `pe` [1, max_len, C]; the add is fused into the LayerNorm kernel."""

    def __init__(self, embed_dim: int, max_seq_length: int = 32):
        super().__init__()
        self.embed_dim, self.max_seq_length = embed_dim, max_seq_length
        self.register_buffer("pe", self._table(embed_dim, max_seq_length))

    @staticmethod
    def _table(embed_dim, max_seq_length):
        position = torch.arange(max_seq_length, device="cpu").unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embed_dim, 2, device="cpu") * (-math.log(10000.0) / embed_dim))
        pe = torch.zeros(1, max_seq_length, embed_dim, device="cpu")
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        return pe

    def reset_table_(self):
        """recompute the table in place (modules materialised from the meta device hold garbage here)."""
        with torch.no_grad():
            self.pe.copy_(self._table(self.embed_dim, self.max_seq_length).to(self.pe.dtype))


FUSED_MOTION_ATTN = os.environ.get("I2V_MOTION_FUSED", "1") != "0"
FUSED_FF = os.environ.get("I2V_FF_FUSED", "1") != "0"
FUSED_FF_TAIL = os.environ.get("I2V_FF_TAIL", "1") != "0"       # proj_out (+ residual, row order) inside the fused feed-forward
# to_out (+ residual) inside the fused attention launches (r5: the fourth pass costs nearly what the HBM-bound GEMM it replaces
# does -- same-box -0.1 .. -0.3 ms per step on every configuration, profiles/r5_attn_outproj.txt)
FUSED_ATTN_OUT = os.environ.get("I2V_ATTN_OUTP", "1") != "0"


class TemporalTransformerBlock(HipModule):
    """BasicTransformerBlock(double_self_attention=True, positional_embeddings="sinusoidal") of A9, on tokens in
    (b, pixel, frame) order: LN(+pe) -> fused q|k GEMM + V^T GEMM -> temporal attention -> out-proj (+residual),
    twice; then LN -> GEGLU FF (+residual)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", attention_bias=False, double_self_attention=True,
                 norm_elementwise_affine=True, norm_eps=1e-5, positional_embeddings="sinusoidal",
                 num_positional_embeddings=32, **_unused):
        super().__init__()
        if cross_attention_dim is not None or not double_self_attention or positional_embeddings != "sinusoidal":
            raise NotImplementedError("motion modules use double self-attention with sinusoidal positions (A9)")
        self.dim, self.heads, self.dim_head = dim, num_attention_heads, attention_head_dim
        self.eps = norm_eps
        self.max_len = num_positional_embeddings
        self.pos_embed = SinusoidalPositionalEmbedding(dim, max_seq_length=num_positional_embeddings)
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn2 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias)
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self._plan = LnFoldPlan()

    def _pack(self):
        p = LazyPack(pe=w16(self.pos_embed.pe[0]))
        for i, (norm, attn) in enumerate(((self.norm1, self.attn1), (self.norm2, self.attn2)), 1):
            p[f"g{i}"], p[f"b{i}"] = w16(norm.weight), w16(norm.bias)
            p[f"wqk{i}"] = w16(torch.cat([attn.to_q.weight, attn.to_k.weight], dim=0))
            p[f"wv{i}"] = w16(attn.to_v.weight)
            p[f"wo{i}"], p[f"bo{i}"] = w16(attn.to_out[0].weight), w16(attn.to_out[0].bias)
        p["g3"], p["b3"] = w16(self.norm3.weight), w16(self.norm3.bias)
        # LayerNorm(+ positional table) folded into the q|k and V^T projections, LayerNorm 3 into the GEGLU projection:
        # (LN(t) + pe[f]) W^T = rstd (t W'^T - mean wsum) + W beta + pe[f] W^T
        pe = self.pos_embed.pe[0].detach().float()
        for i, (norm, attn) in enumerate(((self.norm1, self.attn1), (self.norm2, self.attn2)), 1):
            wqk = torch.cat([attn.to_q.weight, attn.to_k.weight], dim=0)
            p[f"f_wqk{i}"], p[f"f_sqk{i}"], p[f"f_cqk{i}"] = fold_layernorm(wqk, None, norm.weight, norm.bias)
            p[f"f_peqk{i}"] = (pe @ wqk.detach().float().T).to(f16).contiguous()                      # [max_len, 2C]
            p[f"f_wv{i}"], p[f"f_sv{i}"], p[f"f_cv{i}"] = fold_layernorm(attn.to_v.weight, None, norm.weight, norm.bias)
            p[f"f_pev{i}"] = (pe @ attn.to_v.weight.detach().float().T).T.to(f16).contiguous()       # [C, max_len]
        p["f_ff"] = self.ff.fold_norm(self.norm3)
        # the fused LayerNorm + q / k / v + attention kernel's weights (per head, rows padded to 16) and the fused feed-forward's
        # fp32 LayerNorm constants: built on first use (LazyPack)
        for i, attn in enumerate((self.attn1, self.attn2), 1):
            p.lazy(f"wqkv{i}", lambda attn=attn, heads=self.heads: K.pack_motion_qkv(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, heads))
            p.lazy(f"wo_frag{i}", lambda attn=attn, heads=self.heads: K.pack_attn_out(attn.to_out[0].weight, attn.to_out[0].bias, heads))
        self._ma_tables = {}          # (site, frames) -> (gamma fp32, beta + pe[frame] fp32), made on first use
        p.lazy("g3_f32", lambda n3=self.norm3: n3.weight.detach().float().contiguous())
        p.lazy("b3_f32", lambda n3=self.norm3: n3.bias.detach().float().contiguous())
        return p

    def _fold_ok(self, t, frames):
        """(attention sites, feed-forward site): is the LayerNorm fold implemented for this problem's GEMMs?"""
        def probe_attn():
            p = self.packed()
            if frames & (frames - 1):
                return False                                     # the positional table is indexed by a mask
            return (K.gemm(t, p["f_wqk1"], p["f_cqk1"], ln=(p["f_sqk1"], self.eps), rowvec=p["f_peqk1"],
                           rowvec_period=frames, query_ln_support=True) and
                    K.project_vt(t, p["f_wv1"], frames, bias=p["f_cv1"], ln=(p["f_sv1"], self.eps), pe_t=p["f_pev1"],
                                 pe_period=frames, query_ln_support=True))

        def probe_ff():
            p = self.packed()
            return p["f_ff"] is not None and self.ff.folded_supported(t, self.eps, p["f_ff"])

        key = (t.shape[0], frames)
        return self._plan.get(("attn",) + key, probe_attn), self._plan.get(("ff",) + key, probe_ff)

    def _fwd(self, t, n_pixels, frames, tail=None):
        """t [n_pixels * frames, C] in (b, pixel, frame) order.  With `tail` (FeedForward.tail_supported) returns
        (result, applied): applied = the module's proj_out (+ residual, rows back in (b, frame, pixel) order) ran inside the
        fused feed-forward launch and `result` is the module's output; otherwise the caller's proj_out GEMM does it."""
        if frames > self.max_len:
            raise ValueError(f"num_frames {frames} exceeds the positional table ({self.max_len})")
        p = self.packed()
        c = self.dim
        fold_attn, fold_ff = self._fold_ok(t, frames)
        overlap = t.shape[0] <= streams.MAX_ROWS       # q|k beside V^T on two streams where neither fills the chip
        fused = FUSED_MOTION_ATTN and K.motion_attn_supported(t.shape[0], c, self.heads, self.dim_head, frames)
        for i in (1, 2):
            if fused:      # LayerNorm + pe, q / k / v and the attention over the frames in one launch (64^2 level of SD-1.5)
                tab = self._ma_tables.get((i, frames))
                if tab is None:
                    tab = self._ma_tables[(i, frames)] = K.motion_attn_tables(p[f"g{i}"], p[f"b{i}"], p["pe"], frames)
                if FUSED_ATTN_OUT:
                    t = K.motion_attn(t, tab[0], tab[1], p[f"wqkv{i}"], heads=self.heads, head_dim=self.dim_head, frames=frames,
                                      eps=self.eps, out_proj=p[f"wo_frag{i}"])
                    continue
                o = K.motion_attn(t, tab[0], tab[1], p[f"wqkv{i}"], heads=self.heads, head_dim=self.dim_head, frames=frames,
                                  eps=self.eps)
                t = K.gemm(o, p[f"wo{i}"], p[f"bo{i}"], residual=t)
                continue
            n = None if fold_attn else K.layernorm(t, p[f"g{i}"], p[f"b{i}"], self.eps, pe=p["pe"], pe_period=frames)
            with streams.fork(overlap, t.device) as fk:
                with fk.side():
                    if fold_attn:
                        vt = K.project_vt(t, p[f"f_wv{i}"], frames, bias=p[f"f_cv{i}"], ln=(p[f"f_sv{i}"], self.eps),
                                          pe_t=p[f"f_pev{i}"], pe_period=frames)
                    else:
                        vt = K.project_vt(n, p[f"wv{i}"], frames)
                if fold_attn:
                    qk = K.gemm(t, p[f"f_wqk{i}"], p[f"f_cqk{i}"], ln=(p[f"f_sqk{i}"], self.eps),
                                rowvec=p[f"f_peqk{i}"], rowvec_period=frames)
                else:
                    qk = K.gemm(n, p[f"wqk{i}"])
            o = K.temporal_attention(qk[:, :c], qk[:, c:], vt, n_pixels=n_pixels, frames=frames, heads=self.heads,
                                     head_dim=self.dim_head, scale=self.dim_head ** -0.5)
            t = K.gemm(o, p[f"wo{i}"], p[f"bo{i}"], residual=t)
        def ret(v, applied=False):
            return v if tail is None else (v, applied)
        if self.ff.fused_supported(t):
            if self.ff.tail_supported(t, tail):
                return ret(self.ff._fwd_fused(t, p["g3_f32"], p["b3_f32"], self.eps, tail=tail), True)
            return ret(self.ff._fwd_fused(t, p["g3_f32"], p["b3_f32"], self.eps))
        if fold_ff:
            return ret(self.ff._fwd_folded(t, self.eps, p["f_ff"]))
        n = K.layernorm(t, p["g3"], p["b3"], self.eps)
        return ret(self.ff._fwd(n, t))


class TransformerTemporalModel(HipModule):
    """A9 motion module (unet:232-244, 413-425, 607-619).

    MI355X layout: the GroupNorm entry kernel (statistics over (C/G, F, H, W) per clip) writes its output in
    (b, pixel, frame) row order, so the sequence of one pixel is `frames` consecutive rows for every GEMM /
    LayerNorm / attention inside the module, and the proj_out GEMM epilogue stores rows back in (b, frame, pixel)
    order while adding the residual.  The reference's four permute+contiguous copies per module disappear."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None,
                 num_layers=1, dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False,
                 activation_fn="geglu", norm_elementwise_affine=True, double_self_attention=True,
                 positional_embeddings=None, num_positional_embeddings=None):
        super().__init__()
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels, self.inner_dim = in_channels, inner_dim
        self.groups = norm_num_groups
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, dropout=dropout,
                                     cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                     attention_bias=attention_bias, double_self_attention=double_self_attention,
                                     norm_elementwise_affine=norm_elementwise_affine,
                                     positional_embeddings=positional_embeddings,
                                     num_positional_embeddings=num_positional_embeddings)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def _pack(self):
        p = LazyPack(g=w16(self.norm.weight), b=w16(self.norm.bias), wi=w16(self.proj_in.weight),
                     bi=w16(self.proj_in.bias), wo=w16(self.proj_out.weight), bo=w16(self.proj_out.bias))
        # proj_out as the tail of the last block's fused feed-forward (the SD-1.5 64^2 width: a square 320 x 320 Linear)
        p.lazy("tail", lambda po=self.proj_out, ci=self.inner_dim, co=self.in_channels: ff_tail_operands(po.weight, po.bias, ci, co))
        return p

    def _fwd(self, x, num_frames):
        p = self.packed()
        n, hh, ww, c = x.shape
        if n % num_frames != 0:
            raise ValueError(f"batch {n} is not a multiple of num_frames {num_frames}")
        n_pixels = (n // num_frames) * hh * ww
        t = gn_proj_in(x, p["g"], p["b"], self.groups, 1e-6, p["wi"], p["bi"], frames=num_frames)   # rows now (b, pixel, frame)
        applied = False
        for j, blk in enumerate(self.transformer_blocks):
            if j + 1 == len(self.transformer_blocks):
                t, applied = blk._fwd(t, n_pixels, num_frames, tail=(p["tail"], K.sview(x, -1, c), num_frames, hh * ww))
            else:
                t = blk._fwd(t, n_pixels, num_frames)
        if applied:                                                         # proj_out + residual ran inside the feed-forward
            return K.sview(t, n, hh, ww, c)
        out = K.gemm(t, p["wo"], p["bo"], residual=K.sview(x, -1, c), store=I2V_STORE_ROWPERM, frames=num_frames,
                     hw=hh * ww, precise=precise_stream())                  # rows back in (b, frame, pixel)
        return K.sview(out, n, hh, ww, c)

    def forward(self, hidden_states, encoder_hidden_states=None, num_frames: int = 1, return_dict=False, **_unused):
        return (from_tokens(self._fwd(to_tokens(hidden_states), num_frames), hidden_states.dtype),)


# ---------------------------------------------------------------------------------------------------------- A6
class ImageProjection(HipModule):
    """A6 (unet:1284-1287)."""

    def __init__(self, image_embed_dim=768, cross_attention_dim=768, num_image_text_embeds=4):
        super().__init__()
        self.num_image_text_embeds = num_image_text_embeds
        self.cross_attention_dim = cross_attention_dim
        self.image_embeds = nn.Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def _pack(self):
        return dict(w=w16(self.image_embeds.weight), b=w16(self.image_embeds.bias), g=w16(self.norm.weight),
                    be=w16(self.norm.bias))

    def forward(self, image_embeds):
        p = self.packed()
        x = _as_f16_matrix(image_embeds)
        y = K.gemm(x, p["w"], p["b"]).view(-1, self.cross_attention_dim)
        y = K.layernorm(y, p["g"], p["be"], self.norm.eps)
        return y.view(x.shape[0], self.num_image_text_embeds, self.cross_attention_dim)


# ---------------------------------------------------------------------------------------------------------- A11
def _motion(out_channels, heads, groups, cross_dim, max_seq):
    return TransformerTemporalModel(num_attention_heads=heads, in_channels=out_channels, norm_num_groups=groups,
                                    cross_attention_dim=cross_dim, attention_bias=False, activation_fn="geglu",
                                    positional_embeddings="sinusoidal", num_positional_embeddings=max_seq,
                                    attention_head_dim=out_channels // heads)


class DownBlockMotion(nn.Module):
    """A11 (unet:54-68)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1,
                 temporal_num_attention_heads=1, temporal_cross_attention_dim=None, temporal_max_seq_length=32,
                 **_unused):
        super().__init__()
        resnets, motion_modules = [], []
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock2D(cin, out_channels, temb_channels=temb_channels, eps=resnet_eps,
                                         groups=resnet_groups, output_scale_factor=output_scale_factor))
            motion_modules.append(_motion(out_channels, temporal_num_attention_heads, resnet_groups,
                                          temporal_cross_attention_dim, temporal_max_seq_length))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def _fwd(self, x, temb_act, num_frames):
        states = ()
        for resnet, motion in zip(self.resnets, self.motion_modules):
            x = resnet._fwd(x, temb_act)
            x = motion._fwd(x, num_frames)
            states += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d._fwd(x)
            states += (x,)
        return x, states

    def forward(self, hidden_states, temb=None, scale: float = 1.0, num_frames: int = 1):
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        x, states = self._fwd(to_tokens(hidden_states), ta, num_frames)
        dt = hidden_states.dtype
        return from_tokens(x, dt), tuple(from_tokens(s, dt) for s in states)


class UpBlockMotion(nn.Module):
    """A11 (unet:122-137)."""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, resolution_idx=None,
                 num_layers=1, resnet_eps=1e-6, resnet_groups=32, output_scale_factor=1.0, add_upsample=True,
                 temporal_num_attention_heads=1, temporal_cross_attention_dim=None, temporal_max_seq_length=32,
                 **_unused):
        super().__init__()
        resnets, motion_modules = [], []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(resnet_in_channels + res_skip_channels, out_channels,
                                         temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         output_scale_factor=output_scale_factor))
            motion_modules.append(_motion(out_channels, temporal_num_attention_heads, resnet_groups,
                                          temporal_cross_attention_dim, temporal_max_seq_length))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)
        self.resolution_idx = resolution_idx

    def _fwd(self, x, res_tuple, temb_act, num_frames, upsample_size=None):
        for resnet, motion in zip(self.resnets, self.motion_modules):
            skip = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            x = resnet._fwd(x, temb_act, x2=skip)        # torch.cat([x, skip], 1) never materialised (unet:478)
            x = motion._fwd(x, num_frames)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u._fwd(x, upsample_size)
        return x

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, scale: float = 1.0,
                num_frames: int = 1):
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        res = tuple(to_tokens(r) for r in res_hidden_states_tuple)
        return from_tokens(self._fwd(to_tokens(hidden_states), res, ta, num_frames, upsample_size),
                           hidden_states.dtype)


class MotionAdapter(PretrainedMixin, nn.Module):
    """Weight container with diffusers `MotionAdapter`'s state-dict layout (unet:1028-1036) and its
    `save_pretrained` / `from_pretrained` files (pipe:734,745)."""

    def __init__(self, block_out_channels=(320, 640, 1280, 1280), motion_layers_per_block=2,
                 motion_mid_block_layers_per_block=1, motion_num_attention_heads=8, motion_norm_num_groups=32,
                 motion_max_seq_length=32, use_motion_mid_block=True):
        super().__init__()
        self.config = dict(block_out_channels=tuple(block_out_channels),
                           motion_layers_per_block=motion_layers_per_block,
                           motion_mid_block_layers_per_block=motion_mid_block_layers_per_block,
                           motion_num_attention_heads=motion_num_attention_heads,
                           motion_norm_num_groups=motion_norm_num_groups,
                           motion_max_seq_length=motion_max_seq_length,
                           use_motion_mid_block=use_motion_mid_block)

        def mm(ch, n):
            m = nn.Module()
            m.motion_modules = nn.ModuleList([
                _motion(ch, motion_num_attention_heads, motion_norm_num_groups, None, motion_max_seq_length)
                for _ in range(n)])
            return m

        self.down_blocks = nn.ModuleList([mm(c, motion_layers_per_block) for c in block_out_channels])
        self.mid_block = (mm(block_out_channels[-1], motion_mid_block_layers_per_block)
                          if use_motion_mid_block else None)
        self.up_blocks = nn.ModuleList([mm(c, motion_layers_per_block + 1) for c in reversed(block_out_channels)])


# ---------------------------------------------------------------------------------------------------------- A12
class DDIMScheduler:
    """A12 host side: the beta / alpha tables, the timestep list and the per-step coefficient table that the
    device-side `i2v_ddim_cfg_step` kernel reads (pipe:755-757: SD-1.5 scheduler config, clip_sample=False,
    timestep_spacing="linspace", steps_offset=1, epsilon prediction, set_alpha_to_one=False)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                 timestep_spacing="linspace", prediction_type="epsilon", **_unused):
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, timestep_spacing=timestep_spacing,
                           prediction_type=prediction_type)
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(beta_schedule)
        if clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("hot path uses clip_sample=False, epsilon prediction")
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset
        self.timestep_spacing = timestep_spacing
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None, **overrides):
        """pipe:755-757: `DDIMScheduler.from_pretrained(model_path, subfolder="scheduler", clip_sample=False,
        timestep_spacing="linspace", steps_offset=1)`: scheduler_config.json + keyword overrides."""
        import json
        import os
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(path, "scheduler_config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(overrides)
        return cls(**cfg)

    def save_pretrained(self, save_directory: str):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "scheduler_config.json"), "w") as f:
            json.dump({"_class_name": "DDIMScheduler", **self.config}, f, indent=2, sort_keys=True)

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np
        self.num_inference_steps = num_inference_steps
        if self.timestep_spacing == "linspace":
            ts = (np.linspace(0, self.num_train_timesteps - 1, num_inference_steps).round()[::-1]
                  .copy().astype(np.int64))
        elif self.timestep_spacing == "leading":
            ratio = self.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timesteps, eta: float = 0.0) -> torch.Tensor:
        """[len(timesteps), 4] fp32: sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), sqrt(1 - a_prev - sigma_t^2) with
        t_prev = t - num_train_timesteps // num_inference_steps (the FULL step count, A12) and sigma_t of `step_sigmas`
        (0 for eta = 0: the deterministic DDIM update of the hot path)."""
        rows = []
        sig = self.step_sigmas(timesteps, eta)
        for t, s in zip([int(v) for v in timesteps], sig):
            prev_t = t - self.num_train_timesteps // self.num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            rows.append(torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p - s ** 2) ** 0.5]))
        return torch.stack(rows).to(torch.float32)

    def step_sigmas(self, timesteps, eta: float = 0.0):
        """sigma_t = eta sqrt((1 - a_prev) / (1 - a_t)) sqrt(1 - a_t / a_prev) per step (diffusers DDIMScheduler.step's
        `std_dev_t`, pipe:550, 659-660): the weight of the fresh noise a stochastic step adds."""
        out = []
        for t in [int(v) for v in timesteps]:
            prev_t = t - self.num_train_timesteps // self.num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            out.append(float(eta) * float((((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)) ** 0.5))
        return out

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise
