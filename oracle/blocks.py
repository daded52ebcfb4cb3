"""Oracle restatement of the `diffusers` ops the reference's hot path imports.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain fp32 torch, unfused, one
torch op per reference op, so that the op graph is the one the reference runs
(reference imports: src/models/unet_motion_cross_frame_attn.py:5-25,
src/modules/i2v_adapter.py:6-11, src/pipelines/pipeline_i2v_adapter.py:18-42).
Each class cites the SURVEY.md Appendix-A item it follows and the reference call
site that constructs it.  Module / parameter names equal the diffusers ones so
that state-dict keys match SURVEY.md Appendix C.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- A1
class Timesteps(nn.Module):
    """A1.  Constructed at unet:763 as Timesteps(C0, flip_sin_to_cos=True, freq_shift=0)."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    """A1.  unet:766-770: Linear(C0,4C0) -> SiLU -> Linear(4C0,4C0)."""

    def __init__(self, in_channels: int, time_embed_dim: int, act_fn: str = "silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# --------------------------------------------------------------------------- A2/A3
class ResnetBlock2D(nn.Module):
    """A2.  Constructed at unet:203-214, 384-395, 563-574, 594-605."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6, groups=32,
                 output_scale_factor=1.0, **_unused):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = (nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0)
                              if in_channels != out_channels else None)

    def forward(self, input_tensor, temb, scale: float = 1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.nonlinearity(self.norm2(h)))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    """A3.  unet:250-259: Conv2d(C, C, 3, stride 2, padding=downsample_padding).  padding = 0 (the VAE encoder's
    DownEncoderBlock2D): the input is first padded by one zero row / column at the bottom / right."""

    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        out_channels = out_channels or channels
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states, scale: float = 1.0):
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    """A3.  unet:431-432: nearest x2 (or `output_size`) then Conv2d(C, C, 3, padding 1)."""

    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


# --------------------------------------------------------------------------- A4/A5
class Attention(nn.Module):
    """A4 (+A5 when `ip_num_tokens` is set by `install_ip_adapter`).

    diffusers `Attention(query_dim, heads, dim_head, cross_attention_dim, bias=False,
    out_bias=True)` with `AttnProcessor2_0`; constructed at i2v:32-37, i2v:409-418 and
    inside BasicTransformerBlock (attn1/attn2).
    """

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True,
                 dropout=0.0, upcast_attention=False):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        # IP-Adapter decoupled branch (A5), installed by `install_ip_adapter`
        self.ip_num_tokens = 0
        self.ip_scale = 1.0
        self.to_k_ip = None
        self.to_v_ip = None

    def install_ip_adapter(self, to_k_ip_weight, to_v_ip_weight, num_tokens=4, scale=1.0):
        """A5: IPAdapterAttnProcessor2_0(hidden_size, cross_attention_dim, num_tokens, scale) (unet:1264-1279)."""
        hidden = self.inner_dim
        self.to_k_ip = nn.Linear(self.cross_attention_dim, hidden, bias=False)
        self.to_v_ip = nn.Linear(self.cross_attention_dim, hidden, bias=False)
        with torch.no_grad():
            self.to_k_ip.weight.copy_(to_k_ip_weight)
            self.to_v_ip.weight.copy_(to_v_ip_weight)
        self.ip_num_tokens = num_tokens
        self.ip_scale = scale

    def _split(self, t):
        b, l, _ = t.shape
        return t.view(b, l, self.heads, self.dim_head).transpose(1, 2)

    def _sdpa(self, q, k, v):
        # explicit softmax(q k^T * scale) v in fp32 == F.scaled_dot_product_attention without mask
        w = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        w = w.softmax(dim=-1)
        return torch.matmul(w, v)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        ip_ctx = None
        if self.ip_num_tokens and encoder_hidden_states is not None:
            end = ctx.shape[1] - self.ip_num_tokens
            ctx, ip_ctx = ctx[:, :end, :], ctx[:, end:, :]
        q = self._split(self.to_q(hidden_states))
        k = self._split(self.to_k(ctx))
        v = self._split(self.to_v(ctx))
        o = self._sdpa(q, k, v)
        if ip_ctx is not None:
            ipk = self._split(self.to_k_ip(ip_ctx))
            ipv = self._split(self.to_v_ip(ip_ctx))
            o = o + self.ip_scale * self._sdpa(q, ipk, ipv)
        b = hidden_states.shape[0]
        o = o.transpose(1, 2).reshape(b, -1, self.inner_dim)
        o = self.to_out[0](o)
        o = self.to_out[1](o)
        return o


# --------------------------------------------------------------------------- A7
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, scale: float = 1.0):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states, scale: float = 1.0):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class FeedForward(nn.Module):
    """A7: net = [GEGLU(dim, 4dim) | GELU, Dropout, Linear(4dim, dim)]."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", inner_dim=None, bias=True):
        super().__init__()
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        if activation_fn == "geglu":
            act = GEGLU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        else:
            raise ValueError(f"unsupported activation_fn {activation_fn}")
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)])

    def forward(self, hidden_states, scale: float = 1.0):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


# --------------------------------------------------------------------------- A10
class SinusoidalPositionalEmbedding(nn.Module):
    """A10: pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i), w_i=exp(-ln(1e4)*2i/C); x + pe[:, :L]."""

    def __init__(self, embed_dim: int, max_seq_length: int = 32):
        super().__init__()
        position = torch.arange(max_seq_length).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embed_dim, 2) * (-math.log(10000.0) / embed_dim))
        pe = torch.zeros(1, max_seq_length, embed_dim)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return x + self.pe[:, : x.shape[1]]


class BasicTransformerBlock(nn.Module):
    """A7/A9 block: layer-norm variant only (the one the reference can execute).

    double_self_attention=True + positional_embeddings="sinusoidal" gives the motion-module
    block of A9.  The spatial block of the hot path is the subclass in oracle/i2v_adapter.py.
    """

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True,
                 norm_type="layer_norm", norm_eps=1e-5, positional_embeddings=None,
                 num_positional_embeddings=None, ff_inner_dim=None, ff_bias=True, attention_out_bias=True,
                 **_unused):
        super().__init__()
        if norm_type != "layer_norm":
            raise ValueError("only norm_type='layer_norm' is on the hot path (SURVEY 8b)")
        self.only_cross_attention = only_cross_attention
        self.pos_embed = (SinusoidalPositionalEmbedding(dim, max_seq_length=num_positional_embeddings)
                          if positional_embeddings == "sinusoidal" else None)
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias,
                               cross_attention_dim=cross_attention_dim if only_cross_attention else None,
                               out_bias=attention_out_bias)
        if cross_attention_dim is not None or double_self_attention:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
            self.attn2 = Attention(query_dim=dim,
                                   cross_attention_dim=cross_attention_dim if not double_self_attention else None,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                   bias=attention_bias, out_bias=attention_out_bias)
        else:
            self.norm2 = None
            self.attn2 = None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, inner_dim=ff_inner_dim,
                              bias=ff_bias)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, **_unused):
        n = self.norm1(hidden_states)
        if self.pos_embed is not None:
            n = self.pos_embed(n)
        a = self.attn1(n, encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None)
        hidden_states = a + hidden_states
        if self.attn2 is not None:
            n = self.norm2(hidden_states)
            if self.pos_embed is not None:
                n = self.pos_embed(n)
            hidden_states = self.attn2(n, encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


# --------------------------------------------------------------------------- A9
class TransformerTemporalModel(nn.Module):
    """A9.  Motion module; constructed at unet:232-244, 413-425, 607-619."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None,
                 num_layers=1, dropout=0.0, norm_num_groups=32, cross_attention_dim=None,
                 attention_bias=False, activation_fn="geglu", norm_elementwise_affine=True,
                 double_self_attention=True, positional_embeddings=None, num_positional_embeddings=None):
        super().__init__()
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                  attention_bias=attention_bias, double_self_attention=double_self_attention,
                                  norm_elementwise_affine=norm_elementwise_affine,
                                  positional_embeddings=positional_embeddings,
                                  num_positional_embeddings=num_positional_embeddings)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, num_frames: int = 1, return_dict=False, **_unused):
        batch_frames, channel, height, width = hidden_states.shape
        batch_size = batch_frames // num_frames
        residual = hidden_states
        h = hidden_states[None, :].reshape(batch_size, num_frames, channel, height, width)
        h = h.permute(0, 2, 1, 3, 4)
        h = self.norm(h)
        h = h.permute(0, 3, 4, 2, 1).reshape(batch_size * height * width, num_frames, channel)
        h = self.proj_in(h)
        for block in self.transformer_blocks:
            h = block(h, encoder_hidden_states=encoder_hidden_states)
        h = self.proj_out(h)
        h = (h[None, None, :].reshape(batch_size, height, width, num_frames, channel)
             .permute(0, 3, 4, 1, 2).contiguous())
        h = h.reshape(batch_frames, channel, height, width)
        return (h + residual,)


# --------------------------------------------------------------------------- A6
class ImageProjection(nn.Module):
    """A6.  Installed at unet:1284-1287: Linear(1024, 4*768) -> (B,4,768) -> LayerNorm(768)."""

    def __init__(self, image_embed_dim=768, cross_attention_dim=768, num_image_text_embeds=4):
        super().__init__()
        self.num_image_text_embeds = num_image_text_embeds
        self.image_embeds = nn.Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        b = image_embeds.shape[0]
        x = self.image_embeds(image_embeds).reshape(b, self.num_image_text_embeds, -1)
        return self.norm(x)


# --------------------------------------------------------------------------- A11
class DownBlockMotion(nn.Module):
    """A11.  diffusers DownBlockMotion as constructed at unet:54-68: [resnet -> motion] x layers, downsampler."""

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
            motion_modules.append(TransformerTemporalModel(
                num_attention_heads=temporal_num_attention_heads, in_channels=out_channels,
                norm_num_groups=resnet_groups, cross_attention_dim=temporal_cross_attention_dim,
                attention_bias=False, activation_fn="geglu", positional_embeddings="sinusoidal",
                num_positional_embeddings=temporal_max_seq_length,
                attention_head_dim=out_channels // temporal_num_attention_heads))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def forward(self, hidden_states, temb=None, scale: float = 1.0, num_frames: int = 1):
        output_states = ()
        for resnet, motion_module in zip(self.resnets, self.motion_modules):
            hidden_states = resnet(hidden_states, temb)
            hidden_states = motion_module(hidden_states, num_frames=num_frames)[0]
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class UpBlockMotion(nn.Module):
    """A11.  diffusers UpBlockMotion as constructed at unet:122-137."""

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
            motion_modules.append(TransformerTemporalModel(
                num_attention_heads=temporal_num_attention_heads, in_channels=out_channels,
                norm_num_groups=resnet_groups, cross_attention_dim=temporal_cross_attention_dim,
                attention_bias=False, activation_fn="geglu", positional_embeddings="sinusoidal",
                num_positional_embeddings=temporal_max_seq_length,
                attention_head_dim=out_channels // temporal_num_attention_heads))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)
        self.resolution_idx = resolution_idx

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, scale: float = 1.0,
                num_frames: int = 1):
        for resnet, motion_module in zip(self.resnets, self.motion_modules):
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb)
            hidden_states = motion_module(hidden_states, num_frames=num_frames)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class MotionAdapter(nn.Module):
    """Weight container with the state-dict layout of diffusers `MotionAdapter`,
    as consumed by `load_motion_modules` (unet:1028-1036)."""

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
                TransformerTemporalModel(in_channels=ch, norm_num_groups=motion_norm_num_groups,
                                         cross_attention_dim=None, activation_fn="geglu", attention_bias=False,
                                         num_attention_heads=motion_num_attention_heads,
                                         attention_head_dim=ch // motion_num_attention_heads,
                                         positional_embeddings="sinusoidal",
                                         num_positional_embeddings=motion_max_seq_length)
                for _ in range(n)])
            return m

        self.down_blocks = nn.ModuleList([mm(c, motion_layers_per_block) for c in block_out_channels])
        self.mid_block = mm(block_out_channels[-1], motion_mid_block_layers_per_block) if use_motion_mid_block else None
        self.up_blocks = nn.ModuleList([mm(c, motion_layers_per_block + 1) for c in reversed(block_out_channels)])


# --------------------------------------------------------------------------- A12
class DDIMScheduler:
    """A12.  pipe:755-757: DDIMScheduler(SD-1.5 config, clip_sample=False, timestep_spacing="linspace",
    steps_offset=1); eps-prediction, set_alpha_to_one=False."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                 timestep_spacing="linspace", prediction_type="epsilon"):
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
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset
        self.timestep_spacing = timestep_spacing
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        import numpy as np
        self.num_inference_steps = num_inference_steps
        if self.timestep_spacing == "linspace":
            ts = (np.linspace(0, self.num_train_timesteps - 1, num_inference_steps).round()[::-1]
                  .copy().astype(np.int64))
        elif self.timestep_spacing == "leading":
            step_ratio = self.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = ac[timesteps] ** 0.5
        sb = (1 - ac[timesteps]) ** 0.5
        sa = sa.flatten()
        sb = sb.flatten()
        while sa.dim() < original_samples.dim():
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, variance_noise=None):
        """diffusers DDIMScheduler.step, epsilon prediction, no clipping.  eta > 0 (pipe:550, 659-660
        `prepare_extra_step_kwargs`): sigma_t = eta sqrt((1 - a_prev) / (1 - a_t)) sqrt(1 - a_t / a_prev) is taken out of the
        direction term and added back as fresh Gaussian noise (drawn from `generator` unless `variance_noise` is given)."""
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        std = eta * (((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)) ** 0.5
        pred_x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        pred_dir = (1 - a_prev - std ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * pred_x0 + pred_dir
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return prev


class DDPMScheduler(DDIMScheduler):
    """Only `add_noise` with the default linear betas (1e-4 .. 0.02), for the reference's KAT
    (test/test_first_frame_pertubation.py:10,39)."""

    def __init__(self, num_train_timesteps=1000):
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=0.0001, beta_end=0.02,
                         beta_schedule="linear")


def gaussian_blur3(x: torch.Tensor, sigma: float) -> torch.Tensor:
    """torchvision GaussianBlur(kernel_size=3) for one sigma (pipe:112,648): separable 3-tap kernel,
    reflect padding.  torchvision draws sigma ~ U(0.1, 2.0) per call; the oracle takes it explicitly."""
    xs = torch.linspace(-1.0, 1.0, 3)
    pdf = torch.exp(-0.5 * (xs / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :]).to(x.dtype)
    c = x.shape[-3]
    w = k2.expand(c, 1, 3, 3)
    shp = x.shape
    x4 = x.reshape(-1, c, shp[-2], shp[-1])
    x4 = F.pad(x4, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(x4, w, groups=c).reshape(shp)
