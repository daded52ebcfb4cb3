"""UNetMotionCrossFrameAttnModel on the HIP kernels: drop-in for
/root/reference/src/models/unet_motion_cross_frame_attn.py (block factories unet:29-162, block classes
unet:164-694, model unet:696-1451).  Same class names, ctor kwargs, attribute names, state-dict keys,
tensor layouts ((B, F, C, H, W) sample) and error behaviour; the arithmetic runs in libi2v_hip.so.

Inside the model activations stay token-major fp16 ([N, H, W, C]) from conv_in to conv_out: NCHW exists only at
the API edge.  Text / image context K,V are projected once per clip (not once per frame): the reference's
`repeat_interleave(num_frames)` of the context (unet:1355) and of the time embedding (unet:1344) become
index arithmetic (kv_group / rows_per_vec) inside the kernels.
"""
import inspect
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn

from . import kernels as K
from ._lib import HipLibraryError
from .blocks import (Attention, DownBlockMotion, Downsample2D, HipModule, ImageProjection, MotionAdapter,
                     ProjectedContext, ProjectedTemb, ResnetBlock2D, TimestepEmbedding, Timesteps, UpBlockMotion,
                     Upsample2D, _as_f16_matrix, _motion, from_tokens, pack_conv3x3, precise_stream, to_tokens, w16)
from .checkpoint import PretrainedMixin
from .i2v_adapter import I2VAdapterModule, I2VAdapterTransformer2DModel

f16 = torch.float16


def _split_ctx(block, encoder_hidden_states):
    """(text tokens, IP tokens) of a reference-style concatenated context (unet:1353)."""
    t2d = block.attentions[0]
    return t2d.transformer_blocks[0]._split_ctx(encoder_hidden_states)


class CrossFrameAttnDownBlockMotion(nn.Module):
    """unet:164-340."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, transformer_layers_per_block: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, num_attention_heads: int = 1, cross_attention_dim: int = 1280,
                 output_scale_factor: float = 1.0, downsample_padding: int = 1, add_downsample: bool = True,
                 dual_cross_attention: bool = False, use_linear_projection: bool = False,
                 only_cross_attention: bool = False, upcast_attention: bool = False,
                 attention_type: str = "default", temporal_cross_attention_dim: Optional[int] = None,
                 temporal_num_attention_heads: int = 8, temporal_max_seq_length: int = 32):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        resnets, attentions, motion_modules = [], [], []
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock2D(cin, out_channels, temb_channels=temb_channels, eps=resnet_eps,
                                         groups=resnet_groups, output_scale_factor=output_scale_factor))
            attentions.append(I2VAdapterTransformer2DModel(
                num_attention_heads, out_channels // num_attention_heads, in_channels=out_channels,
                num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim,
                norm_num_groups=resnet_groups, use_linear_projection=use_linear_projection,
                only_cross_attention=only_cross_attention))
            motion_modules.append(_motion(out_channels, temporal_num_attention_heads, resnet_groups,
                                          temporal_cross_attention_dim, temporal_max_seq_length))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def _fwd(self, x, temb_act, enable, ctx_text, ctx_ip, num_frames, additional_residuals=None, cfg_shared=False):
        """cfg_shared (first down block only, unet._fwd_tokens): x is ONE of the two identical CFG halves; the first resnet
        and the prompt-independent stage of the first transformer run on it, the outputs carry both halves."""
        states = ()
        n_layers = len(self.resnets)
        for i, (resnet, attn, motion) in enumerate(zip(self.resnets, self.attentions, self.motion_modules)):
            shared = cfg_shared and i == 0
            x = resnet._fwd(x, temb_act.first_half() if shared else temb_act)         # unet:312
            x = attn._fwd(x, enable, num_frames, ctx_text, ctx_ip, cfg_expand=shared)   # unet:313-322
            x = motion._fwd(x, num_frames)                                            # unet:323-326
            if i == n_layers - 1 and additional_residuals is not None:
                raise NotImplementedError("additional_residuals (ControlNet) are not on the hot path")
            states += (x,)
        if self.downsamplers is not None:                                             # unet:334-338
            for d in self.downsamplers:
                x = d._fwd(x)
            states += (x,)
        return x, states

    def forward(self, hidden_states, temb=None, enable_cross_frame_attn: bool = False,
                encoder_hidden_states=None, attention_mask=None, num_frames: int = 1,
                encoder_attention_mask=None, cross_attention_kwargs=None, additional_residuals=None):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        ct, ci = _split_ctx(self, encoder_hidden_states)
        x, states = self._fwd(to_tokens(hidden_states), ta, enable_cross_frame_attn, ct, ci, num_frames,
                              additional_residuals)
        dt = hidden_states.dtype
        return from_tokens(x, dt), tuple(from_tokens(s, dt) for s in states)


class CrossFrameAttnUpBlockMotion(nn.Module):
    """unet:342-529."""

    def __init__(self, in_channels: int, out_channels: int, prev_output_channel: int, temb_channels: int,
                 resolution_idx: Optional[int] = None, dropout: float = 0.0, num_layers: int = 1,
                 transformer_layers_per_block: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, num_attention_heads: int = 1, cross_attention_dim: int = 1280,
                 output_scale_factor: float = 1.0, add_upsample: bool = True, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, only_cross_attention: bool = False,
                 upcast_attention: bool = False, attention_type: str = "default",
                 temporal_cross_attention_dim: Optional[int] = None, temporal_num_attention_heads: int = 8,
                 temporal_max_seq_length: int = 32):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        resnets, attentions, motion_modules = [], [], []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(resnet_in_channels + res_skip_channels, out_channels,
                                         temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                                         output_scale_factor=output_scale_factor))
            attentions.append(I2VAdapterTransformer2DModel(
                num_attention_heads, out_channels // num_attention_heads, in_channels=out_channels,
                num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim,
                norm_num_groups=resnet_groups, use_linear_projection=use_linear_projection,
                only_cross_attention=only_cross_attention))
            motion_modules.append(_motion(out_channels, temporal_num_attention_heads, resnet_groups,
                                          temporal_cross_attention_dim, temporal_max_seq_length))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)
        self.resolution_idx = resolution_idx

    def _fwd(self, x, res_tuple, temb_act, enable, ctx_text, ctx_ip, num_frames, upsample_size=None):
        for resnet, attn, motion in zip(self.resnets, self.attentions, self.motion_modules):
            skip = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            x = resnet._fwd(x, temb_act, x2=skip)                 # cat([x, skip], 1) (unet:478) never materialised
            x = attn._fwd(x, enable, num_frames, ctx_text, ctx_ip)
            x = motion._fwd(x, num_frames)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u._fwd(x, upsample_size)
        return x

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, enable_cross_frame_attn: bool = False,
                encoder_hidden_states=None, cross_attention_kwargs=None, upsample_size=None,
                attention_mask=None, encoder_attention_mask=None, num_frames: int = 1):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        ct, ci = _split_ctx(self, encoder_hidden_states)
        res = tuple(to_tokens(r) for r in res_hidden_states_tuple)
        x = self._fwd(to_tokens(hidden_states), res, ta, enable_cross_frame_attn, ct, ci, num_frames, upsample_size)
        return from_tokens(x, hidden_states.dtype)


class UNetMidBlockCrossFrameAttnMotion(nn.Module):
    """unet:531-694."""

    def __init__(self, in_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 transformer_layers_per_block: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, num_attention_heads: int = 1, output_scale_factor: float = 1.0,
                 cross_attention_dim: int = 1280, dual_cross_attention=False, use_linear_projection=False,
                 upcast_attention=False, attention_type: str = "default", temporal_num_attention_heads: int = 1,
                 temporal_cross_attention_dim: Optional[int] = None, temporal_max_seq_length: int = 32):
        super().__init__()
        self.has_cross_attention = True
        self.num_attention_heads = num_attention_heads
        resnet_groups = resnet_groups if resnet_groups is not None else min(in_channels // 4, 32)

        def res():
            return ResnetBlock2D(in_channels, in_channels, temb_channels=temb_channels, eps=resnet_eps,
                                 groups=resnet_groups, output_scale_factor=output_scale_factor)

        resnets, attentions, motion_modules = [res()], [], []
        for _ in range(num_layers):
            attentions.append(I2VAdapterTransformer2DModel(
                num_attention_heads, in_channels // num_attention_heads, in_channels=in_channels,
                num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim,
                norm_num_groups=resnet_groups, use_linear_projection=use_linear_projection))
            resnets.append(res())
            motion_modules.append(_motion(in_channels, temporal_num_attention_heads, resnet_groups,
                                          temporal_cross_attention_dim, temporal_max_seq_length))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)

    def _fwd(self, x, temb_act, enable, ctx_text, ctx_ip, num_frames):
        x = self.resnets[0]._fwd(x, temb_act)                                         # unet:639
        for attn, resnet, motion in zip(self.attentions, self.resnets[1:], self.motion_modules):
            x = attn._fwd(x, enable, num_frames, ctx_text, ctx_ip)                    # unet:678-692
            x = motion._fwd(x, num_frames)
            x = resnet._fwd(x, temb_act)
        return x

    def forward(self, hidden_states, temb=None, enable_cross_frame_attn: bool = False,
                encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None,
                encoder_attention_mask=None, num_frames: int = 1):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        ta = K.silu(_as_f16_matrix(temb)) if temb is not None else None
        ct, ci = _split_ctx(self, encoder_hidden_states)
        return from_tokens(self._fwd(to_tokens(hidden_states), ta, enable_cross_frame_attn, ct, ci, num_frames),
                           hidden_states.dtype)


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample,
                   resnet_eps, resnet_act_fn, num_attention_heads, resnet_groups=None, cross_attention_dim=None,
                   downsample_padding=None, dual_cross_attention=False, use_linear_projection=True,
                   only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                   temporal_num_attention_heads=8, temporal_max_seq_length=32, transformer_layers_per_block=1):
    """unet:29-92."""
    if down_block_type == "DownBlockMotion":
        return DownBlockMotion(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                               temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                               resnet_groups=resnet_groups, downsample_padding=downsample_padding,
                               temporal_num_attention_heads=temporal_num_attention_heads,
                               temporal_max_seq_length=temporal_max_seq_length)
    if down_block_type == "CrossFrameAttnDownBlockMotion":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossFrameAttnDownBlockMotion")
        return CrossFrameAttnDownBlockMotion(
            num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
            temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
            resnet_groups=resnet_groups, downsample_padding=downsample_padding,
            cross_attention_dim=cross_attention_dim, num_attention_heads=num_attention_heads,
            use_linear_projection=use_linear_projection, only_cross_attention=only_cross_attention,
            temporal_num_attention_heads=temporal_num_attention_heads,
            temporal_max_seq_length=temporal_max_seq_length)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, resnet_eps, resnet_act_fn, num_attention_heads, resolution_idx=None,
                 resnet_groups=None, cross_attention_dim=None, dual_cross_attention=False,
                 use_linear_projection=True, only_cross_attention=False, upcast_attention=False,
                 resnet_time_scale_shift="default", temporal_num_attention_heads=8,
                 temporal_cross_attention_dim=None, temporal_max_seq_length=32, transformer_layers_per_block=1,
                 dropout=0.0):
    """unet:94-162."""
    if up_block_type == "UpBlockMotion":
        return UpBlockMotion(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                             prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                             add_upsample=add_upsample, resnet_eps=resnet_eps, resnet_groups=resnet_groups,
                             resolution_idx=resolution_idx,
                             temporal_num_attention_heads=temporal_num_attention_heads,
                             temporal_max_seq_length=temporal_max_seq_length)
    if up_block_type == "CrossFrameAttnUpBlockMotion":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossFrameAttnUpBlockMotion")
        return CrossFrameAttnUpBlockMotion(
            num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
            prev_output_channel=prev_output_channel, temb_channels=temb_channels, add_upsample=add_upsample,
            resnet_eps=resnet_eps, resnet_groups=resnet_groups, cross_attention_dim=cross_attention_dim,
            num_attention_heads=num_attention_heads, use_linear_projection=use_linear_projection,
            only_cross_attention=only_cross_attention, resolution_idx=resolution_idx,
            temporal_num_attention_heads=temporal_num_attention_heads,
            temporal_max_seq_length=temporal_max_seq_length)
    raise ValueError(f"{up_block_type} does not exist.")


class _Config(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


class UNet3DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


class AttnProcessorHIP:
    """What `attn_processors` reports for an Attention module: the reference holds diffusers processor objects
    (`AttnProcessor2_0`, `IPAdapterAttnProcessor2_0(hidden_size, cross_attention_dim, num_tokens, scale)`,
    unet:1258-1279); here the attention arithmetic is the HIP kernel and the processor is a descriptor of the
    branch configuration (`num_tokens` = 0: plain attention; > 0: + decoupled image cross-attention)."""

    def __init__(self, num_tokens: int = 0, scale: float = 1.0):
        self.num_tokens, self.scale = num_tokens, scale

    def __repr__(self):
        if self.num_tokens:
            return f"IPAdapterAttnProcessorHIP(num_tokens={self.num_tokens}, scale={self.scale})"
        return "AttnProcessorHIP()"

    def __eq__(self, other):
        return isinstance(other, AttnProcessorHIP) and (self.num_tokens, self.scale) == (other.num_tokens, other.scale)


class UNetMotionCrossFrameAttnModel(PretrainedMixin, HipModule):
    """unet:696-1451."""

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 down_block_types: Tuple[str, ...] = ("CrossFrameAttnDownBlockMotion",
                                                      "CrossFrameAttnDownBlockMotion",
                                                      "CrossFrameAttnDownBlockMotion", "DownBlockMotion"),
                 up_block_types: Tuple[str, ...] = ("UpBlockMotion", "CrossFrameAttnUpBlockMotion",
                                                    "CrossFrameAttnUpBlockMotion", "CrossFrameAttnUpBlockMotion"),
                 block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 use_linear_projection: bool = False, num_attention_heads: Union[int, Tuple[int, ...]] = 8,
                 motion_max_seq_length: int = 32, motion_num_attention_heads: int = 8,
                 use_motion_mid_block: int = True, encoder_hid_dim: Optional[int] = None,
                 encoder_hid_dim_type: Optional[str] = None):
        super().__init__()
        self.config = _Config(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            downsample_padding=downsample_padding, mid_block_scale_factor=mid_block_scale_factor, act_fn=act_fn,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
            use_linear_projection=use_linear_projection, num_attention_heads=num_attention_heads,
            motion_max_seq_length=motion_max_seq_length, motion_num_attention_heads=motion_num_attention_heads,
            use_motion_mid_block=use_motion_mid_block, encoder_hid_dim=encoder_hid_dim,
            encoder_hid_dim_type=encoder_hid_dim_type)
        self.sample_size = sample_size
        self.layers_per_block = layers_per_block
        self.num_attention_heads = num_attention_heads
        if act_fn not in ("silu", "swish"):
            raise NotImplementedError("hot path uses act_fn='silu'")
        if len(down_block_types) != len(up_block_types):
            raise ValueError(
                f"Must provide the same number of `down_block_types` as `up_block_types`. `down_block_types`: "
                f"{down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `block_out_channels` as `down_block_types`. "
                f"`block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(num_attention_heads, int) and len(num_attention_heads) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `num_attention_heads` as `down_block_types`. "
                f"`num_attention_heads`: {num_attention_heads}. `down_block_types`: {down_block_types}.")

        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], kernel_size=3, padding=1)   # unet:757
        time_embed_dim = block_out_channels[0] * 4
        self.time_proj = Timesteps(block_out_channels[0], True, 0)                               # unet:763
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim, act_fn=act_fn)
        self.encoder_hid_proj = None
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        if isinstance(num_attention_heads, int):
            num_attention_heads = (num_attention_heads,) * len(down_block_types)

        output_channel = block_out_channels[0]
        for i, t in enumerate(down_block_types):                                                 # unet:783-807
            input_channel = output_channel
            output_channel = block_out_channels[i]
            is_final = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(
                t, num_layers=layers_per_block, in_channels=input_channel, out_channels=output_channel,
                temb_channels=time_embed_dim, add_downsample=not is_final, resnet_eps=norm_eps,
                resnet_act_fn=act_fn, resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                num_attention_heads=num_attention_heads[i], downsample_padding=downsample_padding,
                use_linear_projection=use_linear_projection,
                temporal_num_attention_heads=motion_num_attention_heads,
                temporal_max_seq_length=motion_max_seq_length))

        self.mid_block = UNetMidBlockCrossFrameAttnMotion(                                       # unet:810-822
            in_channels=block_out_channels[-1], temb_channels=time_embed_dim, resnet_eps=norm_eps,
            output_scale_factor=mid_block_scale_factor, cross_attention_dim=cross_attention_dim,
            num_attention_heads=num_attention_heads[-1], resnet_groups=norm_num_groups,
            temporal_num_attention_heads=motion_num_attention_heads,
            temporal_max_seq_length=motion_max_seq_length)

        self.num_upsamplers = 0
        rev_ch = list(reversed(block_out_channels))
        rev_heads = list(reversed(num_attention_heads))
        output_channel = rev_ch[0]
        for i, t in enumerate(up_block_types):                                                   # unet:831-866
            is_final = i == len(block_out_channels) - 1
            prev_output_channel = output_channel
            output_channel = rev_ch[i]
            input_channel = rev_ch[min(i + 1, len(block_out_channels) - 1)]
            add_upsample = not is_final
            if add_upsample:
                self.num_upsamplers += 1
            self.up_blocks.append(get_up_block(
                t, num_layers=layers_per_block + 1, in_channels=input_channel, out_channels=output_channel,
                prev_output_channel=prev_output_channel, temb_channels=time_embed_dim,
                add_upsample=add_upsample, resnet_eps=norm_eps, resnet_act_fn=act_fn,
                resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                num_attention_heads=rev_heads[i], resolution_idx=i, use_linear_projection=use_linear_projection,
                temporal_num_attention_heads=motion_num_attention_heads,
                temporal_max_seq_length=motion_max_seq_length))

        if norm_num_groups is not None:                                                          # unet:869-876
            self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups,
                                              eps=norm_eps)
            self.conv_act = nn.SiLU()
        else:
            raise NotImplementedError("norm_num_groups=None is not on the hot path")
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, kernel_size=3, padding=1)

    # ------------------------------------------------------------------ weight assembly
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_unet2d(cls, unet, motion_adapter, i2v_adapter: Optional[I2VAdapterModule] = None,
                    load_weights: bool = True):
        """unet:883-977."""
        config = dict(unet.config)
        config["_class_name"] = cls.__name__
        config["down_block_types"] = ["CrossFrameAttnDownBlockMotion" if "CrossAttn" in t else "DownBlockMotion"
                                      for t in config["down_block_types"]]
        config["up_block_types"] = ["CrossFrameAttnUpBlockMotion" if "CrossAttn" in t else "UpBlockMotion"
                                    for t in config["up_block_types"]]
        config["motion_num_attention_heads"] = motion_adapter.config["motion_num_attention_heads"]
        config["motion_max_seq_length"] = motion_adapter.config["motion_max_seq_length"]
        config["use_motion_mid_block"] = motion_adapter.config["use_motion_mid_block"]
        if not config.get("num_attention_heads"):
            config["num_attention_heads"] = config["attention_head_dim"]                 # unet:918-919
        model = cls.from_config(config)
        if not load_weights:
            return model
        model.conv_in.load_state_dict(unet.conv_in.state_dict())
        model.time_embedding.load_state_dict(unet.time_embedding.state_dict())
        for i, down_block in enumerate(unet.down_blocks):
            model.down_blocks[i].resnets.load_state_dict(down_block.resnets.state_dict())
            if hasattr(model.down_blocks[i], "attentions"):
                for mine, theirs in zip(model.down_blocks[i].attentions, down_block.attentions):
                    mine.from_transformer2d_model(theirs)
            if model.down_blocks[i].downsamplers:
                model.down_blocks[i].downsamplers.load_state_dict(down_block.downsamplers.state_dict())
        for i, up_block in enumerate(unet.up_blocks):
            model.up_blocks[i].resnets.load_state_dict(up_block.resnets.state_dict())
            if hasattr(model.up_blocks[i], "attentions"):
                for mine, theirs in zip(model.up_blocks[i].attentions, up_block.attentions):
                    mine.from_transformer2d_model(theirs)
            if model.up_blocks[i].upsamplers:
                model.up_blocks[i].upsamplers.load_state_dict(up_block.upsamplers.state_dict())
        model.mid_block.resnets.load_state_dict(unet.mid_block.resnets.state_dict())
        if hasattr(model.mid_block, "attentions"):
            for mine, theirs in zip(model.mid_block.attentions, unet.mid_block.attentions):
                mine.from_transformer2d_model(theirs)
        if unet.conv_norm_out is not None:
            model.conv_norm_out.load_state_dict(unet.conv_norm_out.state_dict())
        model.conv_out.load_state_dict(unet.conv_out.state_dict())
        model.load_motion_modules(motion_adapter)
        if i2v_adapter is not None:
            model.load_i2v_adapter(i2v_adapter)
        p = next(unet.parameters())
        model.to(device=p.device, dtype=p.dtype)                                         # unet:975
        return model

    def freeze_unet_params(self, freeze_animatediff=True) -> None:
        """unet:979-1026: the adapter's to_q / to_out train; with freeze_animatediff=False (`--update_motion_modules`) every
        motion-module parameter too.  (training.UNetAdapterTrainer / AdapterOptimizer take the same switch.)"""
        for p in self.parameters():
            p.requires_grad = False
        for name, p in self.named_parameters():
            if ".i2v_adapter.to_q." in name or ".i2v_adapter.to_out." in name:
                p.requires_grad = True
            if not freeze_animatediff and ".motion_modules." in name:
                p.requires_grad = True

    def load_motion_modules(self, motion_adapter) -> None:
        """unet:1028-1036."""
        for i, down_block in enumerate(motion_adapter.down_blocks):
            self.down_blocks[i].motion_modules.load_state_dict(down_block.motion_modules.state_dict())
        for i, up_block in enumerate(motion_adapter.up_blocks):
            self.up_blocks[i].motion_modules.load_state_dict(up_block.motion_modules.state_dict())
        if hasattr(self.mid_block, "motion_modules"):
            self.mid_block.motion_modules.load_state_dict(motion_adapter.mid_block.motion_modules.state_dict())

    def load_i2v_adapter(self, i2v_adapter: I2VAdapterModule):
        """unet:1038-1041."""
        self.down_blocks.load_state_dict(i2v_adapter.down_blocks.state_dict(), strict=False)
        self.up_blocks.load_state_dict(i2v_adapter.up_blocks.state_dict(), strict=False)
        self.mid_block.load_state_dict(i2v_adapter.mid_block.state_dict(), strict=False)

    def obtain_i2v_adapter_modules(self):
        """unet:1043-1058."""
        sd = {k: v for k, v in self.state_dict().items() if "i2v_adapter" in k}
        m = I2VAdapterModule(self.layers_per_block, self.config.block_out_channels, self.num_attention_heads)
        m.load_state_dict(sd)
        return m

    def obtain_motion_modules(self):
        """unet:1060-1078."""
        sd = {k: v for k, v in self.state_dict().items() if "motion_modules" in k}
        ma = MotionAdapter(block_out_channels=self.config["block_out_channels"],
                           motion_layers_per_block=self.config["layers_per_block"],
                           motion_norm_num_groups=self.config["norm_num_groups"],
                           motion_num_attention_heads=self.config["motion_num_attention_heads"],
                           motion_max_seq_length=self.config["motion_max_seq_length"],
                           use_motion_mid_block=self.config["use_motion_mid_block"])
        ma.load_state_dict(sd)
        return ma

    def save_i2v_adapter_modules(self, save_directory: str, is_main_process: bool = True,
                                 safe_serialization: bool = True, variant: Optional[str] = None,
                                 push_to_hub: bool = False, **kwargs):
        """unet:1080-1097."""
        self.obtain_i2v_adapter_modules().save_pretrained(
            save_directory=save_directory, is_main_process=is_main_process, safe_serialization=safe_serialization,
            variant=variant, push_to_hub=push_to_hub, **kwargs)

    def save_motion_modules(self, save_directory: str, is_main_process: bool = True,
                            safe_serialization: bool = True, variant: Optional[str] = None,
                            push_to_hub: bool = False, **kwargs) -> None:
        """unet:1099-1116."""
        self.obtain_motion_modules().save_pretrained(
            save_directory=save_directory, is_main_process=is_main_process, safe_serialization=safe_serialization,
            variant=variant, push_to_hub=push_to_hub, **kwargs)

    @property
    def attn_processors(self) -> Dict[str, AttnProcessorHIP]:
        """unet:1118-1136: {"<module path>.processor": processor} for every Attention, in registration order."""
        modules = dict(self.named_modules())
        out = {}
        for name in self.attn_processor_names():
            a = modules[name[: -len(".processor")]]
            out[name] = AttnProcessorHIP(a.ip_num_tokens, a.ip_scale)
        return out

    def set_attn_processor(self, processor, _remove_lora=False):
        """unet:1138-1161.  Accepts one descriptor for all layers or a dict keyed like `attn_processors`.  A
        descriptor can change the IP branch's scale or switch the branch off (num_tokens = 0); switching it ON
        needs weights and goes through `_load_ip_adapter_weights`."""
        names = self.attn_processor_names()
        if isinstance(processor, dict) and len(processor) != len(names):
            raise ValueError(
                f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                f" number of attention layers: {len(names)}. Please make sure to pass {len(names)} processor classes.")
        modules = dict(self.named_modules())
        for name in names:
            proc = processor[name] if isinstance(processor, dict) else processor
            if not isinstance(proc, AttnProcessorHIP):
                raise TypeError(f"{name}: expected an AttnProcessorHIP descriptor, got {type(proc).__name__}")
            a = modules[name[: -len(".processor")]]
            if proc.num_tokens and a.to_k_ip is None:
                raise ValueError(f"{name}: the IP-Adapter branch has no weights; load them with "
                                 "`_load_ip_adapter_weights` first")
            a.ip_num_tokens, a.ip_scale = proc.num_tokens, proc.scale

    def attn_processor_names(self):
        """Enumeration order of the reference's `attn_processors` (unet:1118-1136): one entry per Attention,
        walking named_children() => down_blocks, up_blocks, mid_block (registration order unet:776-777,810)."""
        names = []

        def rec(name, module):
            if isinstance(module, Attention):
                names.append(f"{name}.processor")
            for sub, child in module.named_children():
                rec(f"{name}.{sub}", child)

        for name, module in self.named_children():
            rec(name, module)
        return names

    def _load_ip_adapter_weights(self, state_dict):
        """unet:1230-1287 (plain IP-Adapter: 4 image tokens through ImageProjection)."""
        if "proj.weight" not in state_dict["image_proj"]:
            raise NotImplementedError("only the plain IP-Adapter (ip-adapter_sd15.bin layout) is on the hot path")
        num_tokens = 4
        self.encoder_hid_proj = None
        modules = dict(self.named_modules())
        key_id = 1
        for name in self.attn_processor_names():
            if not name.endswith("attn2.processor") or "motion_modules" in name:
                continue                                                                   # unet:1258-1262
            attn = modules[name[: -len(".processor")]]
            attn.install_ip_adapter(state_dict["ip_adapter"][f"{key_id}.to_k_ip.weight"],
                                    state_dict["ip_adapter"][f"{key_id}.to_v_ip.weight"],
                                    num_tokens=num_tokens, scale=1.0)
            key_id += 2                                                                    # unet:1279
        ip = state_dict["image_proj"]
        clip_dim = ip["proj.weight"].shape[-1]
        cross_dim = ip["proj.weight"].shape[0] // 4
        proj = ImageProjection(image_embed_dim=clip_dim, cross_attention_dim=cross_dim,
                               num_image_text_embeds=num_tokens)
        proj.load_state_dict({"image_embeds.weight": ip["proj.weight"], "image_embeds.bias": ip["proj.bias"],
                              "norm.weight": ip["norm.weight"], "norm.bias": ip["norm.bias"]})
        self.encoder_hid_proj = proj.to(device=self.device, dtype=self.dtype)
        self.config.encoder_hid_dim_type = "ip_image_proj"

    # ------------------------------------------------------------------ forward
    def _pack(self):
        cin_pad = K.pad8(self.config.in_channels)
        return dict(w_in=pack_conv3x3(self.conv_in.weight, cin_pad=cin_pad), b_in=w16(self.conv_in.bias),
                    g_out=w16(self.conv_norm_out.weight), be_out=w16(self.conv_norm_out.bias),
                    w_out=pack_conv3x3(self.conv_out.weight), b_out=w16(self.conv_out.bias), cin_pad=cin_pad)

    def packed(self):
        # only the UNet's own leaf parameters feed this pack (children pack themselves)
        leaves = [self.conv_in.weight, self.conv_in.bias, self.conv_norm_out.weight, self.conv_norm_out.bias,
                  self.conv_out.weight, self.conv_out.bias]
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in leaves)
        if self._packed is None or key != self._packed_key:
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed

    def _cross_attention_layers(self):
        """the spatial blocks' attn2 modules, in attn_processors order."""
        modules = dict(self.named_modules())
        return [modules[n[: -len(".processor")]] for n in self.attn_processor_names()
                if n.endswith("attn2.processor") and "motion_modules" not in n]

    def project_context(self, ctx_text, ctx_ip=None, out: Optional[ProjectedContext] = None) -> ProjectedContext:
        """K / V^T of the text (+ image) context for every cross-attention layer, computed ONCE per sample instead of in
        every UNet call (they do not depend on the latents or the timestep).  `out`: a previous result whose buffers
        are overwritten in place (a captured hipGraph keeps reading the same memory for the next sample)."""
        pc = out if out is not None else ProjectedContext(ctx_text, ctx_ip)
        if out is not None:
            pc.text, pc.ip = ctx_text, ctx_ip
        for attn in self._cross_attention_layers():
            pc.kv[attn] = attn.project_kv(ctx_text, ctx_ip, out=pc.kv.get(attn))
            attn.refresh_context_fragments(pc)       # (the fused text cross-attention's packed K / V, where a forward made them)
        return pc

    def _temb_pack(self):
        """time_emb_proj weights of all resnets concatenated ([sum Cout, 4 C0]) + the column slice of each resnet."""
        resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
        key = tuple((r.time_emb_proj.weight.data_ptr(), r.time_emb_proj.weight._version, r.time_emb_proj.bias._version)
                    for r in resnets)
        if getattr(self, "_temb_packed_key", None) != key:
            with torch.no_grad():
                w = w16(torch.cat([r.time_emb_proj.weight for r in resnets], dim=0))
                b = w16(torch.cat([r.time_emb_proj.bias for r in resnets], dim=0))
            slices, off = {}, 0
            for r in resnets:
                slices[r] = (off, r.out_channels)
                off += r.out_channels
            self._temb_packed, self._temb_packed_key = (w, b, slices), key
        return self._temb_packed

    def project_time_table(self, timesteps_f32, out=None):
        """[T, sum Cout] fp16: `time_emb_proj(silu(time_embedding(time_proj(t))))` of every ResnetBlock2D (unet:1336-1343,
        SURVEY A2) for ALL T timesteps of a schedule at once.  The chain depends on t only -- not on the latents -- so the
        pipeline computes it once per sample (like `project_context`) and a replayed step selects its row by the device-side
        step counter (`K.select_row`): 6 launches of M = 1 products per step become one 40 KB copy."""
        wt, bt, _ = self._temb_pack()
        temb = self._embed_time(timesteps_f32.to(torch.float32).contiguous())
        return K.gemm(K.silu(temb), wt, bt, out=out)

    def _fwd_tokens(self, x, temb, enable_cross_frame_attn, ctx_text, ctx_ip, num_frames, cfg_shared=False, temb_proj=None,
                    forward_upsample_size=False):
        """x: model-input tokens [B*F, H, W, cin_pad] fp16; temb [B, 4*C0] fp16 (pre-SiLU) -- or temb_proj [B or 1, sum Cout]:
        the resnets' time-embedding projections already computed (project_time_table); ctx_text [B, Lt, D]
        (+ ctx_ip [B, 4, D]) or a ProjectedContext; returns noise-prediction tokens [B*F, H, W, out_channels]."""
        p = self.packed()
        wt, bt, slices = self._temb_pack()
        # ResnetBlock2D: time_emb_proj(nonlinearity(temb)) of all 22 resnets in one GEMM
        temb_act = ProjectedTemb(temb_proj if temb_proj is not None else K.gemm(K.silu(temb), wt, bt), slices)
        temb_rows = temb_proj.shape[0] if temb_proj is not None else temb.shape[0]
        # cfg_shared: the caller's batch is [unconditional half ; conditional half] of the SAME latents at the SAME timestep
        # (pipe:672-673 `torch.cat([latents] * 2)`).  Until the first text cross-attention the two halves are the same
        # numbers -- conv_in, the first resnet, GroupNorm / proj_in and the whole self- + cross-frame attention stage of the
        # first transformer (two 4096 x 4096 attentions per frame at 512 x 512) -- so they are computed once and duplicated
        # where the prompt enters.  Same arithmetic per element; the bits are identical whenever the dispatcher picks the same
        # kernel forms for the half and the full batch (it does at 16 f x 512 x 512), else equal to reduction-order rounding
        # (tests/test_full_width_gpu.py).
        cfg_shared = (cfg_shared and x.shape[0] % (2 * num_frames) == 0 and (temb_rows == 1 or temb_rows % 2 == 0) and
                      isinstance(self.down_blocks[0], CrossFrameAttnDownBlockMotion))
        if cfg_shared:
            x = x[: x.shape[0] // 2]
        x = K.conv3x3(x, p["w_in"], p["b_in"], precise=precise_stream())                # unet:1359
        res = (K.duplicate_batch(x) if cfg_shared else x,)
        for bi, blk in enumerate(self.down_blocks):                                     # unet:1362-1377
            if getattr(blk, "has_cross_attention", False):
                x, r = blk._fwd(x, temb_act, enable_cross_frame_attn, ctx_text, ctx_ip, num_frames,
                                cfg_shared=cfg_shared and bi == 0)
            else:
                x, r = blk._fwd(x, temb_act, num_frames)
            res += r
        x = self.mid_block._fwd(x, temb_act, enable_cross_frame_attn, ctx_text, ctx_ip, num_frames)
        upsample_size = None
        for bi, blk in enumerate(self.up_blocks):                                       # unet:1406-1436
            r = res[-len(blk.resnets):]
            res = res[: -len(blk.resnets)]
            if bi + 1 < len(self.up_blocks) and forward_upsample_size:                  # unet:1414-1415
                upsample_size = tuple(res[-1].shape[1:3])
            if getattr(blk, "has_cross_attention", False):
                x = blk._fwd(x, r, temb_act, enable_cross_frame_attn, ctx_text, ctx_ip, num_frames, upsample_size)
            else:
                x = blk._fwd(x, r, temb_act, num_frames, upsample_size)
        x = K.groupnorm(x, p["g_out"], p["be_out"], self.config.norm_num_groups, self.config.norm_eps, silu=True)
        # the 4-channel noise prediction stays fp32: the CFG combine (pipe:686-688) would amplify an fp16 rounding here
        # by up to 2 * guidance - 1, and the reference returns `sample`'s dtype (fp32 latents in its driver)
        return K.conv3x3(x, p["w_out"], p["b_out"], out_f32=True)                       # unet:1439-1443

    def _embed_time(self, timesteps_f32, t_index=None):
        """Timesteps -> TimestepEmbedding (unet:1336-1343); one row per sample (the per-frame repeat of unet:1344
        is index arithmetic in the conv epilogue)."""
        t_emb = K.timestep_embedding(timesteps_f32, self.config.block_out_channels[0], t_index=t_index)
        return self.time_embedding(t_emb)

    def _project_image_embeds(self, added_cond_kwargs):
        if self.encoder_hid_proj is not None and self.config.encoder_hid_dim_type == "ip_image_proj":
            if added_cond_kwargs is None or "image_embeds" not in added_cond_kwargs:
                raise ValueError(
                    f"{self.__class__} has the config param `encoder_hid_dim_type` set to 'ip_image_proj' which "
                    "requires the keyword argument `image_embeds` to be passed in  `added_conditions`")
            return self.encoder_hid_proj(added_cond_kwargs.get("image_embeds"))          # unet:1351-1352
        return None

    def forward(self, sample, timestep, enable_cross_frame_attn: bool, encoder_hidden_states,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict: bool = True):
        """unet:1289-1451.  sample (B, F, C, H, W) -> noise prediction (B, F, C, H, W) in sample's dtype.
        `cross_attention_kwargs={"cfg_shared_prefix": True}` (an addition; the reference ignores the dict on this path): the
        caller asserts that sample[B/2:] == sample[:B/2] and that all timesteps are equal -- the classifier-free-guidance
        batch of pipe:672-673 -- and the prompt-independent prefix of the network is computed once (see _fwd_tokens)."""
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        if down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            raise NotImplementedError("ControlNet residuals are not on the hot path")
        if not sample.is_cuda:
            raise HipLibraryError(f"sample is on {sample.device}: the HIP path has no CPU fallback")
        if sample.dim() != 5:
            raise ValueError(f"sample must be (batch, frames, channels, height, width), got {tuple(sample.shape)}")
        b, num_frames, c, hh, ww = sample.shape
        # unet:1304-1311: sizes that do not halve exactly at every level forward the skip tensors' sizes to the up-samplers
        forward_upsample_size = any(s % (2 ** self.num_upsamplers) != 0 for s in (hh, ww))
        timesteps = timestep                                                              # unet:1319-1334
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float32, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None]
        timesteps = timesteps.to(device=sample.device, dtype=torch.float32).expand(b).contiguous()
        temb = self._embed_time(timesteps)
        if isinstance(encoder_hidden_states, ProjectedContext):     # K / V^T already projected (project_context)
            ctx_text, ctx_ip = encoder_hidden_states, encoder_hidden_states.ip
            if self.encoder_hid_proj is not None and ctx_ip is None:
                self._project_image_embeds(None)                                          # raises like unet:1347-1350
        else:
            ctx_text = _as_f16_matrix(encoder_hidden_states)
            ctx_ip = self._project_image_embeds(added_cond_kwargs)
        p = self.packed()
        x = K.nchw_to_tokens(sample.reshape(b * num_frames, c, hh, ww), p["cin_pad"])     # unet:1358
        cfg_shared = bool(cross_attention_kwargs and cross_attention_kwargs.get("cfg_shared_prefix", False))
        y = self._fwd_tokens(x, temb, enable_cross_frame_attn, ctx_text, ctx_ip, num_frames, cfg_shared=cfg_shared,
                             forward_upsample_size=forward_upsample_size)
        out_dt = sample.dtype if sample.dtype in (torch.float32, f16) else f16
        out = K.tokens_to_nchw(y, dtype=out_dt).reshape(b, num_frames, -1, hh, ww)        # unet:1446
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)



# ---------------------------------------------------------------------------------------------------------
# Weight container standing in for diffusers `UNet2DConditionModel` (SD-1.5 layout), the *source* argument of
# `from_unet2d` (unet:883): same module tree / state-dict keys, no forward.
class _Transformer2DSource(I2VAdapterTransformer2DModel):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for blk in self.transformer_blocks:
            del blk.i2v_adapter


class UNet2DConditionModel(PretrainedMixin, nn.Module):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                   "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=768, attention_head_dim=8, use_linear_projection=False):
        super().__init__()
        self.config = _Config(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                              down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              downsample_padding=downsample_padding,
                              mid_block_scale_factor=mid_block_scale_factor, act_fn=act_fn,
                              norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                              cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                              use_linear_projection=use_linear_projection)
        heads = attention_head_dim
        temb = block_out_channels[0] * 4
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.time_proj = Timesteps(block_out_channels[0], True, 0)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], temb)

        def t2d(c):
            return _Transformer2DSource(heads, c // heads, in_channels=c, cross_attention_dim=cross_attention_dim,
                                        norm_num_groups=norm_num_groups,
                                        use_linear_projection=use_linear_projection)

        def res(cin, cout):
            return ResnetBlock2D(cin, cout, temb_channels=temb, eps=norm_eps, groups=norm_num_groups)

        self.down_blocks = nn.ModuleList()
        oc = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            ic, oc = oc, block_out_channels[i]
            blk = nn.Module()
            blk.resnets = nn.ModuleList([res(ic if j == 0 else oc, oc) for j in range(layers_per_block)])
            if "CrossAttn" in t:
                blk.attentions = nn.ModuleList([t2d(oc) for _ in range(layers_per_block)])
            blk.downsamplers = (nn.ModuleList([Downsample2D(oc, padding=downsample_padding)])
                                if i != len(block_out_channels) - 1 else None)
            self.down_blocks.append(blk)
        c = block_out_channels[-1]
        self.mid_block = nn.Module()
        self.mid_block.resnets = nn.ModuleList([res(c, c), res(c, c)])
        self.mid_block.attentions = nn.ModuleList([t2d(c)])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        oc = rev[0]
        for i, t in enumerate(up_block_types):
            prev, oc = oc, rev[i]
            ic = rev[min(i + 1, len(rev) - 1)]
            blk = nn.Module()
            n = layers_per_block + 1
            blk.resnets = nn.ModuleList([
                res((prev if j == 0 else oc) + (ic if j == n - 1 else oc), oc) for j in range(n)])
            if "CrossAttn" in t:
                blk.attentions = nn.ModuleList([t2d(oc) for _ in range(n)])
            blk.upsamplers = nn.ModuleList([Upsample2D(oc)]) if i != len(rev) - 1 else None
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)
