"""Oracle restatement of diffusers `AutoencoderKL` (SD-1.5 VAE) as the reference uses it -- TEST INFRASTRUCTURE ONLY.

The reference decodes the final latents frame by frame (`decode_latents`, pipe:300-320: latents / scaling_factor ->
vae.decode(...).sample -> (B, F, 3, H, W) float) and encodes the condition image (pipe:626-627:
vae.encode(image).latent_dist.sample() * scaling_factor).  `AutoencoderKL` lives in the third-party package diffusers
(pinned 0.24.0, API level 0.25, absent offline): PARITY UNPINNED at that boundary, like the UNet (oracle/__init__.py);
this file restates its published architecture with diffusers' module / parameter names so that `vae/` checkpoints
(config.json + diffusion_pytorch_model.safetensors, pipe:754) load by key:

  Encoder: conv_in 3x3 -> DownEncoderBlock2D x 4 ([ResnetBlock2D(temb=None, eps 1e-6, groups 32)] x layers_per_block,
           Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) + conv 3x3 stride 2; none on the last block)
           -> UNetMidBlock2D (resnet, single-head Attention over H*W with its own GroupNorm, resnet)
           -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out 3x3 (2 x latent channels) ; then quant_conv 1x1
  DiagonalGaussianDistribution: mean, logvar = chunk(2); logvar clamped to [-30, 20]; sample = mean + exp(logvar / 2) eps
  Decoder: post_quant_conv 1x1 -> conv_in 3x3 -> UNetMidBlock2D -> UpDecoderBlock2D x 4 ([ResnetBlock2D] x
           (layers_per_block + 1), Upsample2D nearest x2 + conv 3x3; none on the last block)
           -> GroupNorm -> SiLU -> conv_out 3x3
  mid-block Attention (diffusers `Attention(C, heads = C / attention_head_dim = 1, bias=True, residual_connection=True,
           norm_num_groups=32, eps)` + AttnProcessor2_0): tokens = H*W pixels; GroupNorm over (C/32, H*W); q, k, v
           Linear(C, C) with bias; softmax(q k^T / sqrt(C)) v; Linear(C, C); + input.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .blocks import Downsample2D, ResnetBlock2D, Upsample2D


class VaeAttention(nn.Module):
    def __init__(self, channels: int, heads: int = 1, norm_num_groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.heads = heads
        self.group_norm = nn.GroupNorm(norm_num_groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, hidden_states, temb=None):
        residual = hidden_states
        b, c, h, w = hidden_states.shape
        x = hidden_states.view(b, c, h * w).transpose(1, 2)                      # (B, HW, C)
        x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        d = c // self.heads
        q, k, v = (t.view(b, -1, self.heads, d).transpose(1, 2) for t in (q, k, v))
        w_ = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(w_.softmax(dim=-1), v)
        o = o.transpose(1, 2).reshape(b, -1, c)
        o = self.to_out[1](self.to_out[0](o))
        o = o.transpose(-1, -2).reshape(b, c, h, w)
        return o + residual                                                       # residual_connection, rescale 1


class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, resnet_eps=1e-6, resnet_groups=32, attention_head_dim=None):
        super().__init__()
        attention_head_dim = attention_head_dim or in_channels
        res = lambda: ResnetBlock2D(in_channels, in_channels, temb_channels=None, eps=resnet_eps, groups=resnet_groups)
        self.attentions = nn.ModuleList([VaeAttention(in_channels, in_channels // attention_head_dim, resnet_groups,
                                                      resnet_eps)])
        self.resnets = nn.ModuleList([res(), res()])

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states)
        return self.resnets[1](hidden_states, temb)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_downsample=True,
                 downsample_padding=0):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=resnet_eps,
                          groups=resnet_groups) for i in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=downsample_padding, name="op")])
                             if add_downsample else None)

    def forward(self, hidden_states):
        for r in self.resnets:
            hidden_states = r(hidden_states, None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
        return hidden_states


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_upsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=resnet_eps,
                          groups=resnet_groups) for i in range(num_layers)])
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states):
        for r in self.resnets:
            hidden_states = r(hidden_states, None)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, double_z=True):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList()
        oc = block_out_channels[0]
        for i, c in enumerate(block_out_channels):
            ic, oc = oc, c
            self.down_blocks.append(DownEncoderBlock2D(ic, oc, num_layers=layers_per_block, resnet_groups=norm_num_groups,
                                                       add_downsample=i != len(block_out_channels) - 1))
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], resnet_groups=norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, sample):
        sample = self.conv_in(sample)
        for blk in self.down_blocks:
            sample = blk(sample)
        sample = self.mid_block(sample)
        return self.conv_out(self.conv_act(self.conv_norm_out(sample)))


class Decoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], 3, stride=1, padding=1)
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], resnet_groups=norm_num_groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        oc = rev[0]
        for i, c in enumerate(rev):
            prev, oc = oc, c
            self.up_blocks.append(UpDecoderBlock2D(prev, oc, num_layers=layers_per_block + 1,
                                                   resnet_groups=norm_num_groups, add_upsample=i != len(rev) - 1))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def forward(self, sample):
        sample = self.conv_in(sample)
        sample = self.mid_block(sample)
        for blk in self.up_blocks:
            sample = blk(sample)
        return self.conv_out(self.conv_act(self.conv_norm_out(sample)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters: torch.Tensor):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Config(dict):
    __getattr__ = dict.get


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block=2, latent_channels=4, norm_num_groups=32, sample_size=512,
                 scaling_factor=0.18215):
        super().__init__()
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups, sample_size=sample_size,
                              scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def encode(self, x):
        return _Out(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return _Out(sample=self.decoder(self.post_quant_conv(z)))
