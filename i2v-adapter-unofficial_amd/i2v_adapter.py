"""I2V-Adapter modules on the HIP kernels: drop-in for /root/reference/src/modules/i2v_adapter.py.

Same class names, constructor kwargs, attribute names, state-dict keys and error behaviour as the reference
(`I2VAdapterModule` i2v:49-93, `I2VAdapterTransformer2DModel` i2v:95-354, `I2VAdapterTransformerBlock`
i2v:356-565).  The arithmetic runs in libi2v_hip.so; there is no CPU path.

Cross-frame adapter attention (K1, i2v:475-494), MI355X-first: the reference repeats the frame-0 tokens F
times and runs an ordinary cross-attention (F redundant K/V projections, F copies of K/V in HBM).  Here the
frame-0 rows are gathered once per clip, K0 / V0^T are projected once, and the attention kernel maps every
query frame of a clip onto the same K0 / V0^T (kv_group = F).  The adapter's to_q shares one GEMM with attn1's
to_q | to_k, and both out-projections (+ both biases + the residual) are one dual-source GEMM.
"""
import os
from typing import Optional

import torch
from torch import nn

from . import kernels as K
from . import streams
from .blocks import (gn_proj_in, ff_tail_operands, precise_stream, Attention, FeedForward, HipModule, LazyPack, LnFoldPlan, _as_f16_matrix,
                     fold_layernorm, from_tokens, to_tokens, w16)
from ._lib import HipLibraryError
from .checkpoint import PretrainedMixin

f16 = torch.float16


def get_block(out_channel, block_depth, num_attention_heads, transformer_layers_per_block):
    """i2v:17-47."""
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


class I2VAdapterModule(PretrainedMixin, nn.Module):
    """i2v:49-93: the adapter checkpoint container (`down_blocks[].attentions[].transformer_blocks[].i2v_adapter`);
    `save_pretrained` / `from_pretrained` as the reference's ModelMixin container (pipe:741, unet:1088)."""

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


FUSED_TEXT_ATTN = os.environ.get("I2V_TEXT_FUSED", "1") != "0"
FUSED_ATTN_OUT = os.environ.get("I2V_ATTN_OUTP", "1") != "0"     # to_out + residual inside that launch (as blocks.FUSED_ATTN_OUT)
FUSED_LN_QKV = os.environ.get("I2V_QKV_FUSED", "1") != "0"       # LayerNorm 1 + q | k | q_adapter + V^T projection in one launch


class I2VAdapterTransformerBlock(HipModule):
    """i2v:356-565 (layer-norm branch)."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu",
                 num_embeds_ada_norm: Optional[int] = None, attention_bias: bool = False,
                 only_cross_attention: bool = False, double_self_attention: bool = False,
                 upcast_attention: bool = False, norm_elementwise_affine: bool = True,
                 norm_type: str = "layer_norm", norm_eps: float = 1e-5, final_dropout: bool = False,
                 attention_type: str = "default", positional_embeddings: Optional[str] = None,
                 num_positional_embeddings: Optional[int] = None, ff_inner_dim: Optional[int] = None,
                 ff_bias: bool = True, attention_out_bias: bool = True, **_unused):
        super().__init__()
        if norm_type != "layer_norm" or only_cross_attention or double_self_attention or positional_embeddings:
            raise NotImplementedError("only the layer-norm spatial block of the hot path is implemented "
                                      "(SURVEY 8b: ada-norm / GLIGEN / only_cross_attention branches are dropped)")
        self.dim, self.heads, self.dim_head = dim, num_attention_heads, attention_head_dim
        self.eps = norm_eps
        self.only_cross_attention = only_cross_attention
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias, out_bias=attention_out_bias)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                   bias=attention_bias, out_bias=attention_out_bias)
        else:
            self.norm2 = None
            self.attn2 = None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, inner_dim=ff_inner_dim,
                              bias=ff_bias)
        self.i2v_adapter = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                                     dropout=dropout, bias=attention_bias, cross_attention_dim=dim,
                                     out_bias=attention_out_bias)                           # i2v:409-418
        self._plan = LnFoldPlan()

    def _pack(self):
        a1, ad = self.attn1, self.i2v_adapter
        p = LazyPack(g1=w16(self.norm1.weight), b1=w16(self.norm1.bias), g3=w16(self.norm3.weight),
                     b3=w16(self.norm3.bias))
        # one GEMM: [attn1.to_q | attn1.to_k | i2v_adapter.to_q] (the first 2C rows alone when the adapter is off)
        p["w_qkq"] = w16(torch.cat([a1.to_q.weight, a1.to_k.weight, ad.to_q.weight], dim=0))
        p["w_v1"] = w16(a1.to_v.weight)
        p["w_k_ad"], p["w_v_ad"] = w16(ad.to_k.weight), w16(ad.to_v.weight)
        p["w_o1"], p["b_o1"] = w16(a1.to_out[0].weight), w16(a1.to_out[0].bias)
        # attn1.to_out and i2v_adapter.to_out as one GEMM over K = 2C with summed biases
        p["w_o_dual"] = w16(torch.cat([a1.to_out[0].weight, ad.to_out[0].weight], dim=1))
        p["b_o_dual"] = w16(a1.to_out[0].bias.float() + ad.to_out[0].bias.float())
        if self.attn2 is not None:
            p["g2"], p["b2"] = w16(self.norm2.weight), w16(self.norm2.bias)
            p["w_q2"] = w16(self.attn2.to_q.weight)
            p["w_o2"], p["b_o2"] = w16(self.attn2.to_out[0].weight), w16(self.attn2.to_out[0].bias)
            p["f_q2"] = fold_layernorm(self.attn2.to_q.weight, None, self.norm2.weight, self.norm2.bias)
            # the fused LayerNorm + to_q + text cross-attention kernel's operands (64^2 level of SD-1.5): built on first use
            # (thunks capture child modules and values, never `self`: no module -> _packed -> thunk -> module cycle)
            p.lazy("wq2_frag", lambda a2=self.attn2, heads=self.heads: K.pack_cross_q(a2.to_q.weight, heads))
            p.lazy("wo2_frag", lambda a2=self.attn2, heads=self.heads: K.pack_attn_out(a2.to_out[0].weight, a2.to_out[0].bias, heads))
            p.lazy("g2_f32", lambda n2=self.norm2: n2.weight.detach().float().contiguous())
            p.lazy("b2_f32", lambda n2=self.norm2: n2.bias.detach().float().contiguous())
        # LayerNorm folded into the consuming projections (i2v:444-445 -> q | k | q_adapter and V^T; i2v:510 -> attn2.to_q;
        # i2v:539 -> GEGLU): operands (W o gamma, row sums, W beta + b) of the LayerNorm-folded GEMM
        p["f_qkq"] = fold_layernorm(torch.cat([a1.to_q.weight, a1.to_k.weight, ad.to_q.weight], dim=0), None,
                                    self.norm1.weight, self.norm1.bias)
        p["f_v1"] = fold_layernorm(a1.to_v.weight, None, self.norm1.weight, self.norm1.bias)
        p["f_ff"] = self.ff.fold_norm(self.norm3)
        p.lazy("g3_f32", lambda n3=self.norm3: n3.weight.detach().float().contiguous())
        p.lazy("b3_f32", lambda n3=self.norm3: n3.bias.detach().float().contiguous())
        # the one-launch LayerNorm 1 + [q | k | q_adapter] + V^T projection's operands (64^2 level of SD-1.5), with and without
        # the adapter's query: built on first use
        if K.ln_qkv_supported(128, self.dim, 3 * self.dim, 128):
            p.lazy("g1_f32", lambda n1=self.norm1: n1.weight.detach().float().contiguous())
            p.lazy("b1_f32", lambda n1=self.norm1: n1.bias.detach().float().contiguous())
            p.lazy("w_lnqkv3", lambda a1=a1, ad=ad: K.pack_ln_qkv(torch.cat([a1.to_q.weight, a1.to_k.weight, ad.to_q.weight], dim=0), a1.to_v.weight))
            p.lazy("w_lnqkv2", lambda a1=a1: K.pack_ln_qkv(torch.cat([a1.to_q.weight, a1.to_k.weight], dim=0), a1.to_v.weight))
            # ... and the adapter's K0 | V0^T over the frame-0 rows (i2v:484-492) the same way
            p.lazy("w_lnkv_ad", lambda ad=ad: K.pack_ln_qkv(ad.to_k.weight, ad.to_v.weight))
        return p

    def _fold_ok(self, x, L, rows_qkq):
        """(LayerNorm 1, 2, 3 sites): is the fold implemented for the GEMMs that consume each of them?"""
        M = x.shape[0]

        def probe1():
            p = self.packed()
            wf, ws, cb = p["f_qkq"]
            wv, sv, cv = p["f_v1"]
            return (K.gemm(x, wf[:rows_qkq], cb[:rows_qkq], ln=(ws[:rows_qkq], self.eps), query_ln_support=True) and
                    K.project_vt(x, wv, L, bias=cv, ln=(sv, self.eps), query_ln_support=True))

        def probe2():
            wf, ws, cb = self.packed()["f_q2"]
            return K.gemm(x, wf, cb, ln=(ws, self.eps), query_ln_support=True)

        def probe3():
            p = self.packed()
            return p["f_ff"] is not None and self.ff.folded_supported(x, self.eps, p["f_ff"])

        return (self._plan.get((1, M, L, rows_qkq), probe1),
                self.attn2 is not None and self._plan.get((2, M), probe2), self._plan.get((3, M), probe3))

    def _fwd(self, x, n_img, L, enable_cross_frame_attn, num_frames, ctx_text, ctx_ip, cfg_expand=False, tail=None):
        """x [n_img * L, C] tokens; ctx_text [Bc, Lt, Dc] (+ ctx_ip [Bc, Li, Dc]) with n_img % Bc == 0.
        tail (FeedForward.tail_supported: the model's proj_out + residual): returns (result, applied) -- applied = it ran inside
        the fused feed-forward launch.
        cfg_expand: x holds ONE of the two identical CFG halves of the batch; the self- / cross-frame attention stage
        (i2v:444-501, no dependence on the prompt) runs on it once, and the result is duplicated in front of the text
        cross-attention, where the halves start to differ (returns 2 * n_img images)."""
        p = self.packed()
        c = self.dim
        if enable_cross_frame_attn:
            if num_frames is None:
                raise ValueError('`num_frames` must be provided when `enable_cross_frame_attn` is True.')
            if n_img % num_frames != 0:
                raise ValueError(f'Batch size {n_img} must be divisible by the number of frames {num_frames}.')
        rows_qkq = 3 * c if enable_cross_frame_attn else 2 * c                               # q1 | k1 [| q_adapter]
        fold1, fold2, fold3 = self._fold_ok(x, L, rows_qkq)
        # Where one launch cannot fill the chip (16 x 16 and 8 x 8 levels) the independent chains of this block run on
        # two streams (streams.fork): [q | k | q_adapter projection] beside [V^T projection, frame-0 gather -> K0 / V0^T],
        # then [attn1] beside [adapter attention]; joined before the dual out-projection.
        overlap = x.shape[0] <= streams.MAX_ROWS
        # LayerNorm 1 (i2v:444-445): folded into the q|k|q_ad and V^T GEMMs (never materialised) where the library
        # implements the fold for this level, else one pass that both chains read
        fused_qkv = FUSED_LN_QKV and "w_lnqkv3" in p and K.ln_qkv_supported(x.shape[0], c, rows_qkq, L)
        n = None if (fold1 or fused_qkv) else K.layernorm(x, p["g1"], p["b1"], self.eps)
        k0 = v0t = None
        if fused_qkv:          # LayerNorm 1, q | k | q_adapter and V^T in one launch (64^2 level of SD-1.5): x is read once
            proj, vt1 = K.ln_qkv(x, p["g1_f32"], p["b1_f32"], p["w_lnqkv3" if enable_cross_frame_attn else "w_lnqkv2"],
                                 n_qk=rows_qkq, rows_per_image=L, eps=self.eps)
        with streams.fork(overlap, x.device) as fk:
            with fk.side():
                if fused_qkv:
                    pass
                elif fold1:
                    wv, sv, cv = p["f_v1"]
                    vt1 = K.project_vt(x, wv, L, bias=cv, ln=(sv, self.eps))
                else:
                    vt1 = K.project_vt(n, p["w_v1"], L)
                if enable_cross_frame_attn and fused_qkv and K.ln_qkv_supported(n_img // num_frames * L, c, c, L):
                    # LayerNorm 1 of the frame-0 rows (read in place), K0 and V0^T in ONE launch instead of three
                    clips = n_img // num_frames
                    k0, v0t = K.ln_qkv(x, p["g1_f32"], p["b1_f32"], p["w_lnkv_ad"], n_qk=c, rows_per_image=L, eps=self.eps,
                                       images=clips, x_image_stride=num_frames * L * x.stride(0))
                elif enable_cross_frame_attn:
                    clips = n_img // num_frames
                    if n is None:
                        # LayerNorm 1 folded away: normalise just the frame-0 rows of every clip (1 / num_frames of the
                        # work), read IN PLACE from x as a batch of row blocks -- no gathered copy (i2v:484, no repeat)
                        first = K.layernorm(x.view(clips, num_frames * L, c)[:, :L], p["g1"], p["b1"], self.eps)
                    elif clips == 1:   # one clip: its frame-0 rows are the first L rows as they stand
                        first = n[:L]
                    else:
                        first = torch.empty((clips, L, c), dtype=f16, device=x.device)
                        K.copy3d(n.view(clips, num_frames * L, c)[:, :L], first)             # i2v:484 (no repeat)
                    f2d = first.reshape(-1, c)
                    k0 = K.gemm(f2d, p["w_k_ad"])
                    v0t = K.project_vt(f2d, p["w_v_ad"], L)
            if fused_qkv:
                pass
            elif fold1:
                wf, ws, cb = p["f_qkq"]
                proj = K.gemm(x, wf[:rows_qkq], cb[:rows_qkq], ln=(ws[:rows_qkq], self.eps))
            else:
                proj = K.gemm(n, p["w_qkq"][:rows_qkq])
        with streams.fork(overlap and enable_cross_frame_attn, x.device) as fk:
            if enable_cross_frame_attn:
                with fk.side():
                    o2 = K.attention(proj[:, 2 * c:], k0, v0t, batch_q=n_img, lq=L, lk=L, heads=self.heads,
                                     head_dim=self.dim_head, kv_group=num_frames, scale=self.dim_head ** -0.5)
            o1 = K.attention(proj[:, :c], proj[:, c:2 * c], vt1, batch_q=n_img, lq=L, lk=L, heads=self.heads,
                             head_dim=self.dim_head, scale=self.dim_head ** -0.5)            # i2v:468-473
        if enable_cross_frame_attn:
            x = K.gemm(o1, p["w_o_dual"], p["b_o_dual"], a2=o2, residual=x)                  # i2v:494,501
        else:
            x = K.gemm(o1, p["w_o1"], p["b_o1"], residual=x)                                 # i2v:501
        if cfg_expand:
            x, n_img = K.duplicate_batch(x), 2 * n_img
        if self.attn2 is not None:                                                           # i2v:510-533
            if ctx_text is None:
                raise ValueError("encoder_hidden_states is required by the cross-attention layer")
            if n_img % ctx_text.shape[0] != 0:
                raise ValueError(f"context batch {ctx_text.shape[0]} does not divide batch {n_img}")
            kv_group = n_img // ctx_text.shape[0]
            kv = self.attn2.context_kv(ctx_text, ctx_ip)
            _k, _vt, kip, _vtip, lt, li = kv
            use_ip = kip is not None and bool(self.attn2.ip_num_tokens)
            fused_out2 = False
            if FUSED_TEXT_ATTN and (not use_ip or li <= 16) and \
                    K.cross_attn_fused_supported(x.shape[0], c, self.heads, self.dim_head, lt, kv_group * L):
                # LayerNorm 2, to_q and the attention over the <= 80 context tokens (+ the IP-Adapter's image tokens) in one launch
                frag, frag_ip = self.attn2.context_fragments(ctx_text, kv)
                o = K.cross_attn_fused(x, p["g2_f32"], p["b2_f32"], p["wq2_frag"], frag, heads=self.heads, head_dim=self.dim_head,
                                       ctx_len=lt, rows_per_ctx=kv_group * L, eps=self.eps, scale=self.attn2.scale,
                                       ip_frag=frag_ip, ip_len=li if use_ip else 0, ip_scale=float(self.attn2.ip_scale),
                                       out_proj=p["wo2_frag"] if FUSED_ATTN_OUT else None)
                fused_out2 = FUSED_ATTN_OUT
            else:
                if fold2:
                    wf, ws, cb = p["f_q2"]
                    q = K.gemm(x, wf, cb, ln=(ws, self.eps))
                else:
                    n = K.layernorm(x, p["g2"], p["b2"], self.eps)
                    q = K.gemm(n, p["w_q2"])
                o = self.attn2._cross(q, ctx_text, ctx_ip, n_img, L, kv_group, kv=kv)
            if fused_out2:
                x = o
            else:
                x = K.gemm(o, p["w_o2"], p["b_o2"], residual=x)
        def ret(v, applied=False):
            return v if tail is None else (v, applied)
        if self.ff.fused_supported(x):
            if self.ff.tail_supported(x, tail):                                              # ... and i2v:298-314 with them
                return ret(self.ff._fwd_fused(x, p["g3_f32"], p["b3_f32"], self.eps, tail=tail), True)
            return ret(self.ff._fwd_fused(x, p["g3_f32"], p["b3_f32"], self.eps))            # i2v:539,554,561 in one launch
        if fold3:
            return ret(self.ff._fwd_folded(x, self.eps, p["f_ff"]))                          # i2v:539,554,561
        n = K.layernorm(x, p["g3"], p["b3"], self.eps)                                       # i2v:539
        return ret(self.ff._fwd(n, x))                                                       # i2v:554,561

    def _split_ctx(self, encoder_hidden_states):
        if encoder_hidden_states is None:
            return None, None
        ctx = _as_f16_matrix(encoder_hidden_states)
        nip = self.attn2.ip_num_tokens if self.attn2 is not None else 0
        if not nip:
            return ctx, None
        end = ctx.shape[1] - nip
        ct = torch.empty((ctx.shape[0], end, ctx.shape[2]), dtype=f16, device=ctx.device)
        ci = torch.empty((ctx.shape[0], nip, ctx.shape[2]), dtype=f16, device=ctx.device)
        K.copy3d(ctx[:, :end], ct)
        K.copy3d(ctx[:, end:], ci)
        return ct, ci

    def forward(self, hidden_states, enable_cross_frame_attn: bool = False, num_frames: Optional[int] = None,
                attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, timestep=None,
                cross_attention_kwargs=None, class_labels=None, added_cond_kwargs=None):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        x = _as_f16_matrix(hidden_states)
        b, l, c = x.shape
        ct, ci = self._split_ctx(encoder_hidden_states)
        out = self._fwd(x.view(-1, c), b, l, enable_cross_frame_attn, num_frames, ct, ci)
        return out.view(b, l, c).to(hidden_states.dtype)


class _Out:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class I2VAdapterTransformer2DModel(HipModule):
    """i2v:95-354, continuous-input branch.  In token-major layout the reference's two permute+reshape copies
    (i2v:226,300) are no-ops and the 1x1 proj_in / proj_out convolutions are plain GEMMs (proj_out fuses the
    `+ residual` of i2v:314)."""

    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88,
                 in_channels: Optional[int] = None, out_channels: Optional[int] = None, num_layers: int = 1,
                 dropout: float = 0.0, norm_num_groups: int = 32, cross_attention_dim: Optional[int] = None,
                 attention_bias: bool = False, sample_size: Optional[int] = None,
                 num_vector_embeds: Optional[int] = None, patch_size: Optional[int] = None,
                 activation_fn: str = "geglu", num_embeds_ada_norm: Optional[int] = None,
                 use_linear_projection: bool = False, only_cross_attention: bool = False,
                 double_self_attention: bool = False, upcast_attention: bool = False,
                 norm_type: str = "layer_norm", norm_elementwise_affine: bool = True, norm_eps: float = 1e-5,
                 attention_type: str = "default", caption_channels: int = None):
        super().__init__()
        if in_channels is None or num_vector_embeds is not None or patch_size is not None:
            raise NotImplementedError("only the continuous-input branch is on the hot path (SURVEY 8b)")
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels, self.inner_dim = in_channels, inner_dim
        self.groups = norm_num_groups
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
                                       double_self_attention=double_self_attention, norm_type=norm_type,
                                       norm_elementwise_affine=norm_elementwise_affine, norm_eps=norm_eps)
            for _ in range(num_layers)])
        if use_linear_projection:
            self.proj_out = nn.Linear(inner_dim, in_channels)
        else:
            self.proj_out = nn.Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0)

    def from_transformer2d_model(self, transformer2d_model):
        """i2v:171-182."""
        self.load_state_dict(transformer2d_model.state_dict(), strict=False)
        for mine, theirs in zip(self.transformer_blocks, transformer2d_model.transformer_blocks):
            mine.i2v_adapter.load_state_dict(theirs.attn1.state_dict())
            mine.i2v_adapter.to_out[0].weight.data.zero_()
            mine.i2v_adapter.to_out[0].bias.data.zero_()

    def _pack(self):
        p = LazyPack(g=w16(self.norm.weight), b=w16(self.norm.bias),
                     wi=w16(self.proj_in.weight.reshape(self.inner_dim, self.in_channels)), bi=w16(self.proj_in.bias),
                     wo=w16(self.proj_out.weight.reshape(self.in_channels, self.inner_dim)), bo=w16(self.proj_out.bias))
        # proj_out as the tail of the last block's fused feed-forward (the SD-1.5 64^2 width): built on first use
        p.lazy("tail", lambda po=self.proj_out, ci=self.inner_dim, co=self.in_channels: ff_tail_operands(po.weight, po.bias, ci, co))
        return p

    def packed(self):
        # only this module's own leaf parameters feed its pack (the transformer blocks pack themselves): a training step
        # that writes the adapter's to_q / to_out in place must not rebuild -- and re-allocate -- proj_in / proj_out's copies
        leaves = [self.norm.weight, self.norm.bias, self.proj_in.weight, self.proj_in.bias, self.proj_out.weight,
                  self.proj_out.bias]
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in leaves)
        if self._packed is None or key != self._packed_key:
            for p in leaves:
                if not p.is_cuda:
                    raise HipLibraryError(
                        f"{type(self).__name__} has parameters on {p.device}: the HIP path has no CPU fallback")
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed

    def _fwd(self, x, enable_cross_frame_attn, num_frames, ctx_text, ctx_ip, cfg_expand=False):
        """cfg_expand: x is one of the two identical CFG halves; the output has both (see the block's _fwd)."""
        p = self.packed()
        n_img, hh, ww, c = x.shape
        t = gn_proj_in(x, p["g"], p["b"], self.groups, 1e-6, p["wi"], p["bi"])             # i2v:218-226
        applied = False
        for j, blk in enumerate(self.transformer_blocks):                                    # i2v:285-295
            expand = cfg_expand and j == 0
            n_in = n_img
            if expand:
                n_img, x = 2 * n_img, K.duplicate_batch(x)                                   # the residual of proj_out
            if j + 1 == len(self.transformer_blocks):
                t, applied = blk._fwd(t, n_in, hh * ww, enable_cross_frame_attn, num_frames, ctx_text, ctx_ip,
                                      cfg_expand=expand, tail=(p["tail"], K.sview(x, -1, c), 0, 0))
            else:
                t = blk._fwd(t, n_in, hh * ww, enable_cross_frame_attn, num_frames, ctx_text, ctx_ip, cfg_expand=expand)
        if applied:                                                                          # i2v:298-314 ran inside the feed-forward
            return K.sview(t, n_img, hh, ww, c)
        out = K.gemm(t, p["wo"], p["bo"], residual=K.sview(x, -1, c), precise=precise_stream())   # i2v:298-314
        return K.sview(out, n_img, hh, ww, c)

    def forward(self, hidden_states, enable_cross_frame_attn: bool = False, encoder_hidden_states=None,
                num_frames: Optional[int] = None, timestep=None, added_cond_kwargs=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None,
                return_dict: bool = True):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the hot path (SURVEY 8b)")
        ct, ci = self.transformer_blocks[0]._split_ctx(encoder_hidden_states)
        out = from_tokens(self._fwd(to_tokens(hidden_states), enable_cross_frame_attn, num_frames, ct, ci),
                          hidden_states.dtype)
        if not return_dict:
            return (out,)
        return _Out(out)
