"""AutoencoderKL (SD-1.5 VAE) on the HIP kernels: the two steps either side of the denoising loop.

The reference decodes the final latents frame by frame (`decode_latents`, pipe:300-320) and encodes the condition image
(pipe:626-627) with diffusers' `AutoencoderKL`.  This mirror keeps diffusers' module tree and state-dict keys
(`encoder.down_blocks.{i}.resnets.{j}...`, `decoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}`,
`quant_conv`, `post_quant_conv`), so a `vae/` checkpoint folder (config.json + diffusion_pytorch_model.safetensors,
pipe:754) loads by key, and runs every op on the kernels of the UNet path: token-major fp16 activations, GroupNorm+SiLU,
3x3 implicit-GEMM convolutions with fused bias / shortcut / residual, nearest-2x folded into the conv gather, the
encoder's stride-2 Downsample2D(padding=0) as the `asym_pad` gather.  The mid-block attention has ONE head of dim 512
(> the flash kernel's 160): QK^T and PV run as GEMMs around the row-softmax kernel, per image.
"""
from typing import Optional, Tuple

import torch
from torch import nn

from . import kernels as K
from ._lib import HipLibraryError
from .blocks import Downsample2D, HipModule, ResnetBlock2D, Upsample2D, pack_conv3x3, w16
from .checkpoint import PretrainedMixin

f16 = torch.float16


class VaeAttention(HipModule):
    """diffusers `Attention(C, heads, dim_head, bias=True, residual_connection=True, norm_num_groups, eps)` of
    UNetMidBlock2D with AttnProcessor2_0: GroupNorm over (C / G, H*W), q / k / v Linear with bias, softmax(q k^T /
    sqrt(d)) v, Linear, + input."""

    def __init__(self, channels: int, heads: int = 1, norm_num_groups: int = 32, eps: float = 1e-6):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the SD VAE mid block has a single attention head (attention_head_dim = channels)")
        self.channels, self.heads, self.groups, self.eps = channels, heads, norm_num_groups, eps
        self.group_norm = nn.GroupNorm(norm_num_groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def _pack(self):
        wo, bo = self.to_out[0].weight.float(), self.to_out[0].bias.float()
        return dict(g=w16(self.group_norm.weight), b=w16(self.group_norm.bias),
                    wqk=w16(torch.cat([self.to_q.weight, self.to_k.weight], dim=0)),
                    bqk=w16(torch.cat([self.to_q.bias, self.to_k.bias], dim=0)), wv=w16(self.to_v.weight),
                    wo=w16(wo),
                    # softmax rows sum to 1, so the value bias passes through the attention unchanged:
                    # (P (h Wv^T + bv)) Wo^T + bo = (P h Wv^T) Wo^T + (Wo bv + bo)
                    bo=w16(wo @ self.to_v.bias.float() + bo))

    def _fwd(self, x):
        p = self.packed()
        n, hh, ww, c = x.shape
        hw = hh * ww
        h = K.groupnorm(x, p["g"], p["b"], self.groups, self.eps)
        qk = K.gemm(h.view(-1, c), p["wqk"], p["bqk"])                       # [n * hw, 2c]
        ld = K.pad8(hw)
        # V^T and the score matrix are zero-filled so that the (at most 7) padded key columns contribute 0 * 0
        vt = K.project_vt(h.view(-1, c), p["wv"], hw, out=torch.zeros((n, c, ld), dtype=f16, device=x.device))
        o = torch.empty((n * hw, c), dtype=f16, device=x.device)
        scores = torch.zeros((hw, ld), dtype=f16, device=x.device)
        s_i = scores[:, :hw]
        for i in range(n):                                                   # per image: [hw, hw] scores
            rows = slice(i * hw, (i + 1) * hw)
            # 1 / sqrt(d) in the QK^T epilogue: the fp16 scores stay 22x further from overflow
            K.gemm(qk[rows, :c], qk[rows, c:], out=s_i, out_scale=float(c) ** -0.5)
            K.softmax_rows(s_i, 1.0, out=s_i)
            K.gemm(scores, vt[i], out=o[rows])
        out = K.gemm(o, p["wo"], p["bo"], residual=x.view(-1, c))
        return out.view(n, hh, ww, c)


class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, resnet_eps=1e-6, resnet_groups=32, attention_head_dim=None):
        super().__init__()
        attention_head_dim = attention_head_dim or in_channels
        res = lambda: ResnetBlock2D(in_channels, in_channels, temb_channels=None, eps=resnet_eps, groups=resnet_groups)
        self.attentions = nn.ModuleList([VaeAttention(in_channels, in_channels // attention_head_dim, resnet_groups,
                                                      resnet_eps)])
        self.resnets = nn.ModuleList([res(), res()])

    def _fwd(self, x):
        x = self.resnets[0]._fwd(x, None)
        x = self.attentions[0]._fwd(x)
        return self.resnets[1]._fwd(x, None)


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

    def _fwd(self, x):
        for r in self.resnets:
            x = r._fwd(x, None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d._fwd(x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32, add_upsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels=None, eps=resnet_eps,
                          groups=resnet_groups) for i in range(num_layers)])
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def _fwd(self, x):
        for r in self.resnets:
            x = r._fwd(x, None)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u._fwd(x)
        return x


class _ConvEnds(HipModule):
    """conv_in / GroupNorm + SiLU + conv_out shared by Encoder and Decoder."""

    def _pack(self):
        cin_pad = K.pad8(self.conv_in.in_channels)
        return dict(w_in=pack_conv3x3(self.conv_in.weight, cin_pad=cin_pad), b_in=w16(self.conv_in.bias),
                    g=w16(self.conv_norm_out.weight), be=w16(self.conv_norm_out.bias),
                    w_out=pack_conv3x3(self.conv_out.weight), b_out=w16(self.conv_out.bias), cin_pad=cin_pad)

    def packed(self):
        leaves = [self.conv_in.weight, self.conv_in.bias, self.conv_norm_out.weight, self.conv_norm_out.bias,
                  self.conv_out.weight, self.conv_out.bias]
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in leaves)
        if self._packed is None or key != self._packed_key:
            for p in leaves:
                if not p.is_cuda:
                    raise HipLibraryError(f"{type(self).__name__} has parameters on {p.device}: the HIP path has no "
                                          "CPU fallback")
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed


class Encoder(_ConvEnds):
    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, double_z=True):
        super().__init__()
        self.groups = norm_num_groups
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

    def _fwd(self, x):
        p = self.packed()
        x = K.conv3x3(x, p["w_in"], p["b_in"])
        for blk in self.down_blocks:
            x = blk._fwd(x)
        x = self.mid_block._fwd(x)
        x = K.groupnorm(x, p["g"], p["be"], self.groups, 1e-6, silu=True)
        return K.conv3x3(x, p["w_out"], p["b_out"])


class Decoder(_ConvEnds):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32):
        super().__init__()
        self.groups = norm_num_groups
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

    def _fwd(self, x):
        p = self.packed()
        x = K.conv3x3(x, p["w_in"], p["b_in"])
        x = self.mid_block._fwd(x)
        for blk in self.up_blocks:
            x = blk._fwd(x)
        x = K.groupnorm(x, p["g"], p["be"], self.groups, 1e-6, silu=True)
        return K.conv3x3(x, p["w_out"], p["b_out"], out_f32=True)      # the 3-channel image leaves in fp32


class DiagonalGaussianDistribution:
    """mean / logvar moments of the encoder (fp32, on the device); `sample` = mean + exp(logvar / 2) eps."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters                          # [N, 2 * latent, H, W] fp32
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if generator is not None else torch.device("cpu")
        eps = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32, device=gdev)
        return K.gaussian_sample(self.parameters, eps.to(self.parameters.device))

    def mode(self):
        return self.mean.contiguous()


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Config(dict):
    __getattr__ = dict.get


class AutoencoderKL(PretrainedMixin, HipModule):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block=2, latent_channels=4, norm_num_groups=32, sample_size=512,
                 scaling_factor=0.18215, **_unused):
        super().__init__()
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups, sample_size=sample_size,
                              scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @property
    def device(self):
        return self.quant_conv.weight.device

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def _pack(self):
        lc = self.config.latent_channels
        # 1x1 convs as GEMMs over channels padded to 8: the padded output channels are zero and double as the
        # zero-padded input channels of the decoder's conv_in
        wq = torch.zeros(K.pad8(2 * lc), K.pad8(2 * lc))
        wq[: 2 * lc, : 2 * lc] = self.quant_conv.weight.detach().float().reshape(2 * lc, 2 * lc).cpu()
        bq = torch.zeros(K.pad8(2 * lc))
        bq[: 2 * lc] = self.quant_conv.bias.detach().float().cpu()
        wp = torch.zeros(K.pad8(lc), K.pad8(lc))
        wp[:lc, :lc] = self.post_quant_conv.weight.detach().float().reshape(lc, lc).cpu()
        bp = torch.zeros(K.pad8(lc))
        bp[:lc] = self.post_quant_conv.bias.detach().float().cpu()
        dev = self.device
        return dict(wq=wq.to(dev, f16), bq=bq.to(dev, f16), wp=wp.to(dev, f16), bp=bp.to(dev, f16))

    def packed(self):
        leaves = [self.quant_conv.weight, self.quant_conv.bias, self.post_quant_conv.weight, self.post_quant_conv.bias]
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in leaves)
        if self._packed is None or key != self._packed_key:
            if not leaves[0].is_cuda:
                raise HipLibraryError(f"AutoencoderKL has parameters on {leaves[0].device}: the HIP path has no CPU fallback")
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        """x (N, 3, H, W) in [-1, 1] -> `.latent_dist` (DiagonalGaussianDistribution over (N, 4, H / 8, W / 8))."""
        if not x.is_cuda:
            raise HipLibraryError(f"input is on {x.device}: the HIP path has no CPU fallback")
        p = self.packed()
        lc2 = 2 * self.config.latent_channels
        t = K.nchw_to_tokens(x.float() if x.dtype not in (torch.float32, f16) else x, self.encoder.packed()["cin_pad"])
        m = self.encoder._fwd(t)                                               # [N, h, w, 2 * latent]
        n, hh, ww, _ = m.shape
        mp = torch.zeros((n * hh * ww, K.pad8(lc2)), dtype=f16, device=x.device) if K.pad8(lc2) != lc2 else None
        a = m.view(-1, lc2)
        if mp is not None:
            K.copy3d(a.view(1, -1, lc2), mp[:, :lc2].view(1, -1, lc2))
            a = mp
        q = K.gemm(a, p["wq"], p["bq"]).view(n, hh, ww, -1)                    # quant_conv
        return _Out(latent_dist=DiagonalGaussianDistribution(K.tokens_to_nchw(q, c=lc2, dtype=torch.float32)))

    @torch.no_grad()
    def decode(self, z: torch.Tensor):
        """z (N, 4, h, w) -> `.sample` (N, 3, 8h, 8w) fp32."""
        if not z.is_cuda:
            raise HipLibraryError(f"latents are on {z.device}: the HIP path has no CPU fallback")
        p = self.packed()
        lcp = K.pad8(self.config.latent_channels)
        t = K.nchw_to_tokens(z.float() if z.dtype not in (torch.float32, f16) else z, lcp)
        n, hh, ww, _ = t.shape
        t = K.gemm(t.view(-1, lcp), p["wp"], p["bp"]).view(n, hh, ww, lcp)     # post_quant_conv (padded channels stay 0)
        y = self.decoder._fwd(t)
        return _Out(sample=K.tokens_to_nchw(y, dtype=torch.float32))
