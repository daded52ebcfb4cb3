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
