"""Oracle restatement of /root/reference/src/modules/i2v_adapter.py (TEST INFRASTRUCTURE).

Follows i2v:17-93 (I2VAdapterModule), i2v:95-234,252-314,351-354 (I2VAdapterTransformer2DModel,
continuous-input branch) and i2v:356-565 (I2VAdapterTransformerBlock, layer-norm branch).
Plain fp32 torch; same class names, ctor kwargs, attribute names and state-dict keys.
"""
from typing import Optional

import torch
from torch import nn

from .blocks import Attention, BasicTransformerBlock


def get_block(out_channel, block_depth, num_attention_heads, transformer_layers_per_block):
    """i2v:17-47 -- container tree `attentions[j].transformer_blocks[k].i2v_adapter`."""
    block = nn.Module()
    attn_blocks = []
    for _ in range(block_depth):
        attn_block = nn.Module()
        tbs = []
        for _ in range(transformer_layers_per_block):
            tb = nn.Module()
            tb.i2v_adapter = Attention(query_dim=out_channel, heads=num_attention_heads,
                                       dim_head=out_channel // num_attention_heads,
                                       cross_attention_dim=out_channel)
            tbs.append(tb)
        attn_block.transformer_blocks = nn.ModuleList(tbs)
        attn_blocks.append(attn_block)
    block.attentions = nn.ModuleList(attn_blocks)
    return block


class I2VAdapterModule(nn.Module):
    """i2v:49-93 -- the adapter checkpoint format (weights only, no arithmetic)."""

    def __init__(self, block_depth, block_out_channels, num_attention_heads,
                 transformer_layers_per_block: int = 1, mid_block_depth: int = 1):
        super().__init__()
        self.config = dict(block_depth=block_depth, block_out_channels=tuple(block_out_channels),
                           num_attention_heads=num_attention_heads,
                           transformer_layers_per_block=transformer_layers_per_block,
                           mid_block_depth=mid_block_depth)
        self.down_blocks = nn.ModuleList([
            get_block(c, block_depth, num_attention_heads, transformer_layers_per_block)
            for c in block_out_channels[:-1]])
        rev = list(reversed(block_out_channels[:-1]))
        up_blocks = nn.ModuleList([
            get_block(c, block_depth + 1, num_attention_heads, transformer_layers_per_block) for c in rev])
        self.up_blocks = nn.ModuleList([nn.Identity()]) + up_blocks
        self.mid_block = get_block(block_out_channels[-1], mid_block_depth, num_attention_heads,
                                   transformer_layers_per_block)

    def forward(self):
        pass


class I2VAdapterTransformerBlock(BasicTransformerBlock):
    """i2v:356-565.  BasicTransformerBlock + `self.i2v_adapter` (i2v:409-418)."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu",
                 attention_bias: bool = False, upcast_attention: bool = False, attention_out_bias: bool = True,
                 **kwargs):
        super().__init__(dim, num_attention_heads, attention_head_dim, dropout=dropout,
                         cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                         attention_bias=attention_bias, attention_out_bias=attention_out_bias, **kwargs)
        self.i2v_adapter = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                                     dropout=dropout, bias=attention_bias, cross_attention_dim=dim,
                                     out_bias=attention_out_bias)

    def forward(self, hidden_states, enable_cross_frame_attn: bool = False, num_frames: Optional[int] = None,
                attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, **_unused):
        batch_size = hidden_states.shape[0]
        n = self.norm1(hidden_states)                                              # i2v:444-445
        if self.pos_embed is not None:
            n = self.pos_embed(n)
        attn_output = self.attn1(                                                  # i2v:468-473
            n, encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None)
        if enable_cross_frame_attn:                                                # i2v:476-494
            if num_frames is None:
                raise ValueError('`num_frames` must be provided when `enable_cross_frame_attn` is True.')
            if batch_size % num_frames != 0:
                raise ValueError(
                    f'Batch size {batch_size} must be divisible by the number of frames {num_frames}.')
            first = n[0:batch_size:num_frames]                                     # i2v:484
            first = first.repeat_interleave(num_frames, dim=0)                     # 'b n d -> (b f) n d', i2v:485
            attn_output = attn_output + self.i2v_adapter(n, encoder_hidden_states=first)
        hidden_states = attn_output + hidden_states                                # i2v:501
        if self.attn2 is not None:                                                 # i2v:510-533
            n = self.norm2(hidden_states)
            if self.pos_embed is not None:
                n = self.pos_embed(n)
            hidden_states = self.attn2(n, encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states         # i2v:539,554,561
        return hidden_states


class I2VAdapterTransformer2DModel(nn.Module):
    """i2v:95-354, continuous-input branch (diffusers Transformer2DModel, A8)."""

    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88,
                 in_channels: Optional[int] = None, out_channels: Optional[int] = None, num_layers: int = 1,
                 dropout: float = 0.0, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 attention_bias: bool = False, activation_fn: str = "geglu", use_linear_projection: bool = False,
                 only_cross_attention: bool = False, double_self_attention: bool = False,
                 upcast_attention: bool = False, norm_type: str = "layer_norm",
                 norm_elementwise_affine: bool = True, norm_eps: float = 1e-5, attention_type: str = "default",
                 **_unused):
        super().__init__()
        if in_channels is None:
            raise ValueError("only the continuous-input branch (in_channels given) is on the hot path")
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner_dim)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            I2VAdapterTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, dropout=dropout,
                                       cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                       attention_bias=attention_bias, only_cross_attention=only_cross_attention,
                                       double_self_attention=double_self_attention,
                                       norm_type=norm_type, norm_elementwise_affine=norm_elementwise_affine,
                                       norm_eps=norm_eps)
            for _ in range(num_layers)])
        if use_linear_projection:
            self.proj_out = nn.Linear(inner_dim, in_channels)
        else:
            self.proj_out = nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0)

    def from_transformer2d_model(self, transformer2d_model):
        """i2v:171-182: copy spatial weights, initialise adapter from attn1, zero its to_out."""
        self.load_state_dict(transformer2d_model.state_dict(), strict=False)
        for mine, theirs in zip(self.transformer_blocks, transformer2d_model.transformer_blocks):
            mine.i2v_adapter.load_state_dict(theirs.attn1.state_dict())
            mine.i2v_adapter.to_out[0].weight.data.zero_()
            mine.i2v_adapter.to_out[0].bias.data.zero_()

    def forward(self, hidden_states, enable_cross_frame_attn: bool = False, encoder_hidden_states=None,
                num_frames: Optional[int] = None, attention_mask=None, encoder_attention_mask=None,
                return_dict: bool = True, **_unused):
        batch, _, height, width = hidden_states.shape
        residual = hidden_states
        h = self.norm(hidden_states)                                               # i2v:218
        if not self.use_linear_projection:
            h = self.proj_in(h)                                                    # i2v:220-224
            inner_dim = h.shape[1]
            h = h.permute(0, 2, 3, 1).reshape(batch, height * width, inner_dim)    # i2v:226
        else:
            inner_dim = h.shape[1]
            h = h.permute(0, 2, 3, 1).reshape(batch, height * width, inner_dim)
            h = self.proj_in(h)
        for block in self.transformer_blocks:                                      # i2v:285-295
            h = block(h, enable_cross_frame_attn=enable_cross_frame_attn, num_frames=num_frames,
                      encoder_hidden_states=encoder_hidden_states)
        if not self.use_linear_projection:
            h = h.reshape(batch, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()   # i2v:300
            h = self.proj_out(h)
        else:
            h = self.proj_out(h)
            h = h.reshape(batch, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        output = h + residual                                                      # i2v:314
        if not return_dict:
            return (output,)
        return _Out(output)


class _Out:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]
