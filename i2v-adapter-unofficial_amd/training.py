"""Adapter training step, first vertical slice: forward + backward of ONE I2VAdapterTransformerBlock on the HIP
kernels, producing the gradients the reference's optimiser consumes.

The reference trains only `i2v_adapter.to_q` and `i2v_adapter.to_out` of every spatial block (`freeze_unet_params`,
unet:979-1026) with torch autograd over the whole frozen UNet (src/train_image_to_video.py:839-884: forward with
`enable_cross_frame_attn=True`, MSE without the first frame :848-856, backward, clip, step).  Gradients reach an early
adapter only THROUGH every later frozen layer, so a block's backward must deliver, besides its three parameter gradients,
the gradient with respect to its input hidden states.  This module does that for the block of i2v:420-565:

    n1 = LN1(x);  o1 = SDPA(n1 Wq, n1 Wk, n1 Wv);  oa = SDPA(n1 Wqa, n1[frame 0] Wka, n1[frame 0] Wva)      i2v:444-492
    x1 = x + o1 Wo^T + bo + oa Woa^T + boa                                                                    i2v:494-501
    x2 = x1 + SDPA(LN2(x1) Wq2, ctx Wk2, ctx Wv2) Wo2^T + bo2                                                 i2v:510-533
    x3 = x2 + GEGLU(LN3(x2) W1^T + b1) W2^T + b2                                                              i2v:539-561

Every product runs on the library: the forward on the inference kernels (un-fused here, because the backward needs the
intermediates), Linear dgrad = i2v_gemm_f16 over the transposed weight, Linear wgrad = i2v_gemm_f16 over channel-major copies
of the two activations (dW = dY^T X), attention backward = i2v_attention_bwd_f16 (dK0 / dV0 of the cross-frame attention
summed over the clip's frames inside the kernel), LayerNorm / GEGLU backward, bias column sums.  Activation gradients are
fp16 under the caller's loss scale (as the reference's fp16 mixed precision scales its loss); parameter gradients come back
un-scaled in fp32.  There is no CPU fallback.

Not here yet (the rest of f4): conv / GroupNorm / motion-module backward, the other blocks' chaining, the optimiser and the
RCCL all-reduce of the adapter gradients (~100 MB per step).
"""
import torch

from . import kernels as K
from ._lib import I2V_EPI_GEGLU
from .blocks import pack_geglu, w16

f16 = torch.float16
# dW = dY^T X sums M (tokens) products: the fp16 result of the GEMM is scaled down by this and scaled back in fp32
WGRAD_OUT_SCALE = 2.0 ** -6


def wgrad(dy, x, tokens_per_batch=None):
    """dW [N, K] (fp32) = dY^T X for dY [M, N], X [M, K]: one GEMM over the channel-major copies, contraction over M."""
    m = dy.shape[0]
    dyt = K.transpose_tokens(dy, m).view(dy.shape[1], -1)
    xt = K.transpose_tokens(x, m).view(x.shape[1], -1)
    return K.gemm(dyt, xt, out_scale=WGRAD_OUT_SCALE).float() / WGRAD_OUT_SCALE


class AdapterBlockTrainer:
    """forward(...) keeps what backward(...) needs; backward returns {"hidden_states": dL/dx (fp16, still loss-scaled),
    "i2v_adapter.to_q.weight", "i2v_adapter.to_out.0.weight", "i2v_adapter.to_out.0.bias": fp32, un-scaled}."""

    def __init__(self, block):
        if block.attn2 is None:
            raise NotImplementedError("the spatial block of the hot path has a text cross-attention")
        if block.attn2.ip_num_tokens:
            raise NotImplementedError("IP-Adapter image tokens in the training step are not implemented yet")
        self.block = block
        self._saved = None

    def _weights(self):
        b = self.block
        a1, ad, a2, ff = b.attn1, b.i2v_adapter, b.attn2, b.ff
        t = lambda w: w16(w.detach().t())
        w1, b1 = pack_geglu(ff.net[0].proj.weight.detach(), ff.net[0].proj.bias.detach())
        return dict(
            g1=w16(b.norm1.weight), be1=w16(b.norm1.bias), g2=w16(b.norm2.weight), be2=w16(b.norm2.bias),
            g3=w16(b.norm3.weight), be3=w16(b.norm3.bias),
            w_qkq=w16(torch.cat([a1.to_q.weight, a1.to_k.weight, ad.to_q.weight], dim=0)), w_v1=w16(a1.to_v.weight),
            w_k_ad=w16(ad.to_k.weight), w_v_ad=w16(ad.to_v.weight),
            w_o_dual=w16(torch.cat([a1.to_out[0].weight, ad.to_out[0].weight], dim=1)),
            b_o_dual=w16(a1.to_out[0].bias.float() + ad.to_out[0].bias.float()),
            w_q2=w16(a2.to_q.weight), w_k2=w16(a2.to_k.weight), w_v2=w16(a2.to_v.weight),
            w_o2=w16(a2.to_out[0].weight), b_o2=w16(a2.to_out[0].bias),
            w1=w1, b1=b1, w2=w16(ff.net[2].weight), b2=w16(ff.net[2].bias),
            # dgrad operands: dX = dY W  ==  gemm(dY, w = W^T)
            w2_t=t(ff.net[2].weight), w1_t=w16(w1.t()), w_o2_t=t(a2.to_out[0].weight), w_q2_t=t(a2.to_q.weight),
            w_o1_t=t(a1.to_out[0].weight), w_oa_t=t(ad.to_out[0].weight),
            w_qk1_t=w16(torch.cat([a1.to_q.weight.t(), a1.to_k.weight.t()], dim=1)),      # [C, 2C]: [dq1 | dk1] -> dn1
            w_vqa_t=w16(torch.cat([a1.to_v.weight.t(), ad.to_q.weight.t()], dim=1)),      # [dv1 | dqa] -> dn1
            w_kva_t=w16(torch.cat([ad.to_k.weight.t(), ad.to_v.weight.t()], dim=1)))      # [dk0 | dv0] -> dn1[frame 0]

    @torch.no_grad()
    def forward(self, x, n_img, L, num_frames, ctx_text):
        """x [n_img * L, C] fp16 tokens, ctx_text [Bc, Lt, Dc] fp16; returns x3 [n_img * L, C]."""
        b = self.block
        if n_img % num_frames != 0:
            raise ValueError(f"Batch size {n_img} must be divisible by the number of frames {num_frames}.")   # i2v:479-481
        w = self._weights()
        c, heads, d = b.dim, b.heads, b.dim_head
        clips = n_img // num_frames
        n1 = K.layernorm(x, w["g1"], w["be1"], b.eps)
        proj = K.gemm(n1, w["w_qkq"])                                                # [q1 | k1 | q_adapter]
        q1, k1, qa = proj[:, :c], proj[:, c:2 * c], proj[:, 2 * c:]
        v1 = K.gemm(n1, w["w_v1"])
        o1 = K.attention(q1, k1, K.transpose_tokens(v1, L), batch_q=n_img, lq=L, lk=L, heads=heads, head_dim=d)
        first = torch.empty((clips, L, c), dtype=f16, device=x.device)
        K.copy3d(n1.view(clips, num_frames * L, c)[:, :L], first)                    # i2v:484 (frame-0 tokens, no repeat)
        f2d = first.view(-1, c)
        k0, v0 = K.gemm(f2d, w["w_k_ad"]), K.gemm(f2d, w["w_v_ad"])
        oa = K.attention(qa, k0, K.transpose_tokens(v0, L), batch_q=n_img, lq=L, lk=L, heads=heads, head_dim=d,
                         kv_group=num_frames)
        x1 = K.gemm(o1, w["w_o_dual"], w["b_o_dual"], a2=oa, residual=x)
        n2 = K.layernorm(x1, w["g2"], w["be2"], b.eps)
        q2 = K.gemm(n2, w["w_q2"])
        bc, lt, dc = ctx_text.shape
        ctx2d = ctx_text.reshape(-1, dc).contiguous()
        kc, vc = K.gemm(ctx2d, w["w_k2"]), K.gemm(ctx2d, w["w_v2"])
        group2 = n_img // bc
        o2 = K.attention(q2, kc, K.transpose_tokens(vc, lt), batch_q=n_img, lq=L, lk=lt, heads=heads, head_dim=d,
                         kv_group=group2)
        x2 = K.gemm(o2, w["w_o2"], w["b_o2"], residual=x1)
        n3 = K.layernorm(x2, w["g3"], w["be3"], b.eps)
        h = K.gemm(n3, w["w1"], w["b1"])                                             # pre-activation, (value, gate) interleaved
        y = K.gemm(n3, w["w1"], w["b1"], epilogue=I2V_EPI_GEGLU)
        x3 = K.gemm(y, w["w2"], w["b2"], residual=x2)
        self._saved = dict(w=w, x=x, n1=n1, q1=q1, k1=k1, qa=qa, v1=v1, o1=o1, k0=k0, v0=v0, oa=oa, x1=x1, q2=q2, kc=kc,
                           vc=vc, o2=o2, x2=x2, h=h, n_img=n_img, L=L, F=num_frames, lt=lt, group2=group2)
        return x3

    @torch.no_grad()
    def backward(self, grad_out, loss_scale=1.0):
        """grad_out = loss_scale * dL/dx3, fp16 [n_img * L, C]."""
        s = self._saved
        if s is None:
            raise RuntimeError("backward() needs a forward() first")
        b, w = self.block, s["w"]
        c, heads, d = b.dim, b.heads, b.dim_head
        n_img, L, F = s["n_img"], s["L"], s["F"]
        clips = n_img // F
        # feed-forward (i2v:539-561)
        dy = K.gemm(grad_out, w["w2_t"])
        dh = K.geglu_bwd(s["h"], dy)
        dn3 = K.gemm(dh, w["w1_t"])
        g2 = K.layernorm_bwd(s["x2"], dn3, w["g3"], b.eps, add=grad_out)              # dL/dx2
        # text cross-attention (i2v:510-533); the context K / V are frozen: dQ only
        do2 = K.gemm(g2, w["w_o2_t"])
        dq2, _, _ = K.attention_bwd(s["q2"], s["kc"], s["vc"], s["o2"], do2, batch_q=n_img, lq=L, lk=s["lt"], heads=heads,
                                    head_dim=d, kv_group=s["group2"], need_dkv=False)
        dn2 = K.gemm(dq2, w["w_q2_t"])
        g1 = K.layernorm_bwd(s["x1"], dn2, w["g2"], b.eps, add=g2)                    # dL/dx1
        # self-attention + cross-frame adapter attention (i2v:444-501)
        do1, doa = K.gemm(g1, w["w_o1_t"]), K.gemm(g1, w["w_oa_t"])
        d_wout = wgrad(g1, s["oa"])                                                   # i2v_adapter.to_out.0.weight
        d_bout = K.colsum(g1)
        dq1, dk1, dv1 = K.attention_bwd(s["q1"], s["k1"], s["v1"], s["o1"], do1, batch_q=n_img, lq=L, lk=L, heads=heads,
                                        head_dim=d)
        dqa, dk0, dv0 = K.attention_bwd(s["qa"], s["k0"], s["v0"], s["oa"], doa, batch_q=n_img, lq=L, lk=L, heads=heads,
                                        head_dim=d, kv_group=F)                       # dK0 / dV0 summed over the frames
        d_wq = wgrad(dqa, s["n1"])                                                    # i2v_adapter.to_q.weight
        dn1 = K.gemm(dq1, w["w_qk1_t"], a2=dk1)
        dn1 = K.gemm(dv1, w["w_vqa_t"], a2=dqa, residual=dn1, out=dn1)
        for clip in range(clips):                                                     # frame-0 rows also fed K0 / V0
            rows = dn1[clip * F * L: clip * F * L + L]
            K.gemm(dk0[clip * L:(clip + 1) * L], w["w_kva_t"], a2=dv0[clip * L:(clip + 1) * L], residual=rows, out=rows)
        g0 = K.layernorm_bwd(s["x"], dn1, w["g1"], b.eps, add=g1)                     # dL/dx
        inv = 1.0 / float(loss_scale)
        return {"hidden_states": g0, "i2v_adapter.to_q.weight": d_wq * inv, "i2v_adapter.to_out.0.weight": d_wout * inv,
                "i2v_adapter.to_out.0.bias": d_bout * inv}


# ------------------------------------------------------------------------------------------------------------------------
# The frozen layers around the adapter blocks: input gradients only (no parameter of theirs trains, unet:979-1026), each
# from the library's kernels: conv dgrad = the conv kernel over the flipped, transposed weights; GroupNorm(+SiLU) backward;
# temporal attention backward = i2v_attention_bwd_f16 with batch = pixels, sequence = frames.
from ._lib import I2V_STORE_ROWPERM  # noqa: E402
from .blocks import pack_conv3x3  # noqa: E402


def conv_dgrad_weight(weight, cout_pad=None):
    """packed weights of the input-gradient convolution: W'[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]."""
    return pack_conv3x3(weight.detach().transpose(0, 1).flip(2, 3), cin_pad=cout_pad)


def _t(w):
    return w16(w.detach().t())


class ResnetTrainer:
    """ResnetBlock2D: out = conv2(silu(GN2(conv1(silu(GN1(x))) + temb))) + shortcut(x)."""

    def __init__(self, resnet):
        self.m = resnet
        if resnet.output_scale_factor != 1.0:
            raise NotImplementedError("output_scale_factor = 1 on the hot path")

    @torch.no_grad()
    def forward(self, x, temb_rows, x2=None):
        m, p = self.m, self.m.packed()
        n, hh, ww, c1 = x.shape
        a1 = K.groupnorm(x, p["g1"], p["b1"], m.groups, m.eps, x2=x2, silu=True)
        rpv = (n // temb_rows.shape[0]) * hh * ww
        h1 = K.conv3x3(a1, p["w1"], p["cb1"], rowvec=temb_rows, rows_per_vec=rpv)
        a2 = K.groupnorm(h1, p["g2"], p["b2"], m.groups, m.eps, silu=True)
        s = x
        if m.conv_shortcut is not None:
            a2d = None if x2 is None else x2.view(-1, x2.shape[3])
            s = K.gemm(x.view(-1, c1), p["ws"], p["bs"], a2=a2d).view(n, hh, ww, m.out_channels)
        self.saved = (x, x2, h1)
        return K.conv3x3(a2, p["w2"], p["cb2"], residual=s)

    @torch.no_grad()
    def backward(self, g):
        m, p = self.m, self.m.packed()
        x, x2, h1 = self.saved
        n, hh, ww, c1 = x.shape
        d_a2 = K.conv3x3(g, conv_dgrad_weight(m.conv2.weight))
        d_h1 = K.groupnorm_bwd(h1, d_a2, p["g2"], p["b2"], m.groups, m.eps, silu=True)
        d_a1 = K.conv3x3(d_h1, conv_dgrad_weight(m.conv1.weight))
        res = K.groupnorm_bwd(x, d_a1, p["g1"], p["b1"], m.groups, m.eps, x2=x2, silu=True)
        dx, dx2 = res if x2 is not None else (res, None)
        g2d = g.view(-1, m.out_channels)
        if m.conv_shortcut is not None:
            wst = _t(m.conv_shortcut.weight.reshape(m.out_channels, m.in_channels))          # [Cin, Cout]
            dx = K.gemm(g2d, wst[:c1].contiguous(), residual=dx.view(-1, c1)).view(n, hh, ww, c1)
            if x2 is not None:
                c2 = x2.shape[3]
                dx2 = K.gemm(g2d, wst[c1:].contiguous(), residual=dx2.view(-1, c2)).view(n, hh, ww, c2)
        else:
            dx = K.add(dx, g)
        return dx, dx2


class Transformer2DTrainer:
    """I2VAdapterTransformer2DModel (i2v:184-354): GroupNorm -> proj_in -> block -> proj_out + residual."""

    def __init__(self, t2d):
        if len(t2d.transformer_blocks) != 1 or t2d.use_linear_projection:
            raise NotImplementedError("one transformer block per Transformer2D, 1x1-conv projections (SD-1.5)")
        self.m = t2d
        self.block = AdapterBlockTrainer(t2d.transformer_blocks[0])

    @torch.no_grad()
    def forward(self, x, num_frames, ctx_text):
        m, p = self.m, self.m.packed()
        n, hh, ww, c = x.shape
        nrm = K.groupnorm(x, p["g"], p["b"], m.groups, 1e-6)
        t = K.gemm(nrm.view(-1, c), p["wi"], p["bi"])
        t = self.block.forward(t, n, hh * ww, num_frames, ctx_text)
        self.saved = x
        return K.gemm(t, p["wo"], p["bo"], residual=x.view(-1, c)).view(n, hh, ww, c)

    @torch.no_grad()
    def backward(self, g, loss_scale):
        m, p = self.m, self.m.packed()
        x = self.saved
        n, hh, ww, c = x.shape
        dt = K.gemm(g.view(-1, c), _t(p["wo"]))
        grads = self.block.backward(dt, loss_scale=loss_scale)
        dn = K.gemm(grads.pop("hidden_states"), _t(p["wi"])).view(n, hh, ww, c)
        dx = K.groupnorm_bwd(x, dn, p["g"], p["b"], m.groups, 1e-6)
        return K.add(dx, g), grads


class MotionModuleTrainer:
    """TransformerTemporalModel: clip-wide GroupNorm -> proj_in -> [LN + PE -> self-attention over the frames of
    a pixel -> + residual] x 2 -> LN -> GEGLU FF -> proj_out -> + residual, on rows in (batch, pixel, frame) order."""

    def __init__(self, mm):
        if len(mm.transformer_blocks) != 1:
            raise NotImplementedError("one temporal block per motion module (AnimateDiff v1.5)")
        self.m = mm

    @torch.no_grad()
    def forward(self, x, num_frames):
        m, p = self.m, self.m.packed()
        blk = m.transformer_blocks[0]
        q = blk.packed()
        n, hh, ww, c = x.shape
        hw, clips, F = hh * ww, n // num_frames, num_frames
        n_pixels = clips * hw
        nrm = K.groupnorm(x, p["g"], p["b"], m.groups, 1e-6, frames_per_stat=F)
        t = K.gemm(K.permute_rows(nrm.view(-1, c), clips, F, hw, True), p["wi"], p["bi"])
        stages = []
        for i in (1, 2):
            nl = K.layernorm(t, q[f"g{i}"], q[f"b{i}"], blk.eps, pe=q["pe"], pe_period=F)
            qk = K.gemm(nl, q[f"wqk{i}"])
            v = K.gemm(nl, q[f"wv{i}"])
            o = K.temporal_attention(qk[:, :c], qk[:, c:], K.transpose_tokens(v, F), n_pixels=n_pixels, frames=F,
                                     heads=blk.heads, head_dim=blk.dim_head, scale=blk.dim_head ** -0.5)
            stages.append((t, qk, v, o))
            t = K.gemm(o, q[f"wo{i}"], q[f"bo{i}"], residual=t)
        ff = blk.ff.packed()
        n3 = K.layernorm(t, q["g3"], q["b3"], blk.eps)
        h = K.gemm(n3, ff["w1"], ff["b1"])
        y = K.gemm(n3, ff["w1"], ff["b1"], epilogue=I2V_EPI_GEGLU)
        t3 = K.gemm(y, ff["w2"], ff["b2"], residual=t)
        self.saved = (x, stages, t, h)
        return K.gemm(t3, p["wo"], p["bo"], residual=x.view(-1, c), store=I2V_STORE_ROWPERM, frames=F, hw=hw).view(n, hh, ww, c)

    @torch.no_grad()
    def backward(self, g, num_frames):
        m, p = self.m, self.m.packed()
        blk = m.transformer_blocks[0]
        q, ff = blk.packed(), blk.ff.packed()
        x, stages, t2, h = self.saved
        n, hh, ww, c = x.shape
        hw, clips, F = hh * ww, n // num_frames, num_frames
        n_pixels = clips * hw
        gp = K.permute_rows(g.view(-1, c), clips, F, hw, True)
        dt = K.gemm(gp, _t(p["wo"]))                                                  # dL/dt3
        dy = K.gemm(dt, _t(ff["w2"]))
        dn3 = K.gemm(K.geglu_bwd(h, dy), w16(ff["w1"].t()))
        dt = K.layernorm_bwd(t2, dn3, q["g3"], blk.eps, add=dt)
        for i in (2, 1):
            t_in, qk, v, o = stages[i - 1]
            do = K.gemm(dt, _t(q[f"wo{i}"]))
            dq, dk, dv = K.attention_bwd(qk[:, :c], qk[:, c:], v, o, do, batch_q=n_pixels, lq=F, lk=F, heads=blk.heads,
                                         head_dim=blk.dim_head, scale=blk.dim_head ** -0.5)
            dnl = K.gemm(dq, w16(q[f"wqk{i}"].t()), a2=dk)                            # [dq | dk] [Wq ; Wk]
            dnl = K.gemm(dv, _t(q[f"wv{i}"]), residual=dnl, out=dnl)
            dt = K.layernorm_bwd(t_in, dnl, q[f"g{i}"], blk.eps, add=dt)
        dnp = K.gemm(dt, _t(p["wi"]))
        dn = K.permute_rows(dnp, clips, F, hw, False).view(n, hh, ww, c)
        dx = K.groupnorm_bwd(x, dn, p["g"], p["b"], m.groups, 1e-6, frames_per_stat=F)
        return K.add(dx, g)


class DownsampleTrainer:
    def __init__(self, d):
        if d.padding != 1:
            raise NotImplementedError("Downsample2D(padding=1) on the UNet path")
        self.m = d

    @torch.no_grad()
    def forward(self, x):
        p = self.m.packed()
        return K.conv3x3(x, p["w"], p["b"], stride=2)

    @torch.no_grad()
    def backward(self, g):
        return K.conv3x3(K.zero_insert2x(g), conv_dgrad_weight(self.m.conv.weight))


class UpsampleTrainer:
    def __init__(self, u):
        self.m = u

    @torch.no_grad()
    def forward(self, x):
        p = self.m.packed()
        return K.conv3x3(x, p["w"], p["b"], upsample=True)

    @torch.no_grad()
    def backward(self, g):
        return K.sum_pool2x(K.conv3x3(g, conv_dgrad_weight(self.m.conv.weight)))
