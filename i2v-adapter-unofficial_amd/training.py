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

Below the block: the backward of the frozen layers around it (ResnetBlock2D, Transformer2D, the motion modules, the
samplers), `UNetAdapterTrainer` (the whole forward + backward of the reference's training step for the adapter parameters)
and `AdapterOptimizer` (clip + AdamW over flat fp32 buckets, ONE all-reduce of the adapter gradients over RCCL per step).
"""
import torch

from . import kernels as K
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


class _Memo:
    """Operand copies derived from weights (transposes for the dgrad GEMMs, flipped conv kernels, fused / concatenated
    projections), made once for the frozen layers.  (Rebuilt every step they were ~430 aten launches and 8 ms of a 148 ms step.)

    An entry is keyed by (tag, ids of its source tensors) and is valid while every source is the SAME live object with the
    same version counter, storage address, dtype and device -- an in-place write, a `.data` swap, `.to()` / `.half()` or a
    `load_state_dict(assign=True)` all rebuild it.  Sources are held by WEAK reference: when one dies (a packed tensor that
    its module rebuilt, a deleted model) the entry -- and the GPU memory of its copy -- goes with it, and a recycled id can
    never alias an old entry.  Anything derived from a trainable tensor (the adapter's to_q / to_out, registered with
    mark_trainable) is remade on every call and never kept."""

    def __init__(self):
        self.d = {}
        self.volatile = set()      # ids of the trainable tensors (registered by AdapterBlockTrainer)
        self._keep = []

    def mark_trainable(self, tensors):
        for t in tensors:
            self.volatile.add(id(t))
            self._keep.append(t)

    @staticmethod
    def _sig(t):
        return (t._version, t.data_ptr(), t.dtype, t.device)

    def get(self, tag, srcs, fn):
        import weakref
        if any(id(t) in self.volatile for t in srcs):   # derived from a trainable tensor: never kept
            return fn()
        key = (tag,) + tuple(id(t) for t in srcs)
        sig = tuple(self._sig(t) for t in srcs)
        hit = self.d.get(key)
        if hit is not None and hit[0] == sig and all(r() is t for r, t in zip(hit[1], srcs)):
            return hit[2]
        val = fn()
        drop = lambda _ref, key=key, d=self.d: d.pop(key, None)     # a source died: its entry (and the copy) goes
        self.d[key] = (sig, tuple(weakref.ref(t, drop) for t in srcs), val)
        return val

    def clear(self):
        self.d.clear()


_memo = _Memo()


class AdapterBlockTrainer:
    """forward(...) keeps what backward(...) needs; backward returns {"hidden_states": dL/dx (fp16, still loss-scaled),
    "i2v_adapter.to_q.weight", "i2v_adapter.to_out.0.weight", "i2v_adapter.to_out.0.bias": fp32, un-scaled}."""

    def __init__(self, block):
        if block.attn2 is None:
            raise NotImplementedError("the spatial block of the hot path has a text cross-attention")
        self.block = block
        self._saved = None
        ad = block.i2v_adapter
        _memo.mark_trainable([ad.to_q.weight, ad.to_out[0].weight, ad.to_out[0].bias])     # unet:1001-1006

    def _weights(self):
        b = self.block
        a1, ad, a2, ff = b.attn1, b.i2v_adapter, b.attn2, b.ff
        mm = lambda tag, srcs, fn: _memo.get(tag, tuple(srcs), fn)
        c16 = lambda prm: mm("w16", [prm], lambda: w16(prm))                     # fp16 contiguous copy of one parameter
        t = lambda prm: mm("t", [prm], lambda: w16(prm.detach().t()))           # its transpose (dgrad operand)
        pw, pb = ff.net[0].proj.weight, ff.net[0].proj.bias
        w1, b1 = mm("geglu", [pw, pb], lambda: pack_geglu(pw.detach(), pb.detach()))
        q1w, k1w, v1w, o1w, o1b = a1.to_q.weight, a1.to_k.weight, a1.to_v.weight, a1.to_out[0].weight, a1.to_out[0].bias
        qaw, kaw, vaw, oaw, oab = ad.to_q.weight, ad.to_k.weight, ad.to_v.weight, ad.to_out[0].weight, ad.to_out[0].bias
        return dict(
            g1=c16(b.norm1.weight), be1=c16(b.norm1.bias), g2=c16(b.norm2.weight), be2=c16(b.norm2.bias),
            g3=c16(b.norm3.weight), be3=c16(b.norm3.bias),
            w_qkq=mm("qkq", [q1w, k1w, qaw], lambda: w16(torch.cat([q1w, k1w, qaw], dim=0))), w_v1=c16(v1w),
            w_k_ad=c16(kaw), w_v_ad=c16(vaw),
            w_o_dual=mm("o_dual", [o1w, oaw], lambda: w16(torch.cat([o1w, oaw], dim=1))),
            b_o_dual=mm("b_o_dual", [o1b, oab], lambda: w16(o1b.float() + oab.float())),
            w_q2=c16(a2.to_q.weight), w_k2=c16(a2.to_k.weight), w_v2=c16(a2.to_v.weight),
            w_o2=c16(a2.to_out[0].weight), b_o2=c16(a2.to_out[0].bias),
            w1=w1, b1=b1, w2=c16(ff.net[2].weight), b2=c16(ff.net[2].bias),
            w_k_ip=c16(a2.to_k_ip.weight) if a2.ip_num_tokens else None,
            w_v_ip=c16(a2.to_v_ip.weight) if a2.ip_num_tokens else None,
            # dgrad operands: dX = dY W  ==  gemm(dY, w = W^T)
            w2_t=t(ff.net[2].weight), w1_t=mm("t", [w1], lambda: w16(w1.t())), w_o2_t=t(a2.to_out[0].weight),
            w_q2_t=t(a2.to_q.weight), w_o1_t=t(o1w), w_oa_t=t(oaw),
            w_qk1_t=mm("qk1_t", [q1w, k1w], lambda: w16(torch.cat([q1w.t(), k1w.t()], dim=1))),   # [C, 2C]: [dq1 | dk1] -> dn1
            w_vqa_t=mm("vqa_t", [v1w, qaw], lambda: w16(torch.cat([v1w.t(), qaw.t()], dim=1))),   # [dv1 | dqa] -> dn1
            w_kva_t=mm("kva_t", [kaw, vaw], lambda: w16(torch.cat([kaw.t(), vaw.t()], dim=1))))   # [dk0 | dv0] -> dn1[frame 0]

    @torch.no_grad()
    def forward(self, x, n_img, L, num_frames, ctx_text, ctx_ip=None):
        """x [n_img * L, C] fp16 tokens, ctx_text [Bc, Lt, Dc] (+ ctx_ip [Bc, 4, Dc]: the IP-Adapter image tokens, frozen
        decoupled cross-attention of unet:1263-1279) fp16; returns x3 [n_img * L, C]."""
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
        # (return_lse: the forward kernel writes each row's log2-sum-exp beside O; the backward sweeps take it from the tape
        #  instead of recomputing Q K^T for it)
        o1, lse1 = K.attention(q1, k1, K.transpose_tokens(v1, L), batch_q=n_img, lq=L, lk=L, heads=heads, head_dim=d,
                               return_lse=True)
        first = torch.empty((clips, L, c), dtype=f16, device=x.device)
        K.copy3d(n1.view(clips, num_frames * L, c)[:, :L], first)                    # i2v:484 (frame-0 tokens, no repeat)
        f2d = first.view(-1, c)
        k0, v0 = K.gemm(f2d, w["w_k_ad"]), K.gemm(f2d, w["w_v_ad"])
        oa, lsea = K.attention(qa, k0, K.transpose_tokens(v0, L), batch_q=n_img, lq=L, lk=L, heads=heads, head_dim=d,
                               kv_group=num_frames, return_lse=True)
        x1 = K.gemm(o1, w["w_o_dual"], w["b_o_dual"], a2=oa, residual=x)
        n2 = K.layernorm(x1, w["g2"], w["be2"], b.eps)
        q2 = K.gemm(n2, w["w_q2"])
        bc, lt, dc = ctx_text.shape
        ctx2d = ctx_text.reshape(-1, dc).contiguous()
        kc, vc = K.gemm(ctx2d, w["w_k2"]), K.gemm(ctx2d, w["w_v2"])
        group2 = n_img // bc
        o2, lse2 = K.attention(q2, kc, K.transpose_tokens(vc, lt), batch_q=n_img, lq=L, lk=lt, heads=heads, head_dim=d,
                               kv_group=group2, return_lse=True)
        ip = None
        if ctx_ip is not None and b.attn2.ip_num_tokens:
            li = ctx_ip.shape[1]
            ip2d = ctx_ip.reshape(-1, dc).to(f16).contiguous()
            kip, vip = K.gemm(ip2d, w["w_k_ip"]), K.gemm(ip2d, w["w_v_ip"])
            vipt = K.transpose_tokens(vip, li)
            kw = dict(batch_q=n_img, lq=L, lk=li, heads=heads, head_dim=d, kv_group=group2)
            o_ip, lse_ip = K.attention(q2, kip, vipt, return_lse=True, **kw)          # kept alone: its backward needs it
            o_sum = torch.empty_like(o2)
            K.copy3d(o2.view(1, -1, c), o_sum.view(1, -1, c))
            K.attention(q2, kip, vipt, out=o_sum, accumulate=True, acc_scale=float(b.attn2.ip_scale), **kw)
            ip = dict(k=kip, v=vip, o=o_ip, li=li, scale=float(b.attn2.ip_scale), lse=lse_ip)
            o2_text, o2 = o2, o_sum
        else:
            o2_text = o2
        x2 = K.gemm(o2, w["w_o2"], w["b_o2"], residual=x1)
        n3 = K.layernorm(x2, w["g3"], w["be3"], b.eps)
        h = K.gemm(n3, w["w1"], w["b1"])                                             # pre-activation, (value, gate) interleaved
        y = K.geglu(h)                                                               # (kept for the backward: one GEMM, not two)
        x3 = K.gemm(y, w["w2"], w["b2"], residual=x2)
        self._saved = dict(w=w, x=x, n1=n1, q1=q1, k1=k1, qa=qa, v1=v1, o1=o1, k0=k0, v0=v0, oa=oa, x1=x1, q2=q2, kc=kc,
                           vc=vc, o2=o2_text, ip=ip, x2=x2, h=h, n_img=n_img, L=L, F=num_frames, lt=lt, group2=group2,
                           lse1=lse1, lsea=lsea, lse2=lse2)
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
                                    head_dim=d, kv_group=s["group2"], need_dkv=False, lse=s["lse2"])
        if s["ip"] is not None:       # + ip_scale * softmax(q K_ip^T) V_ip: the same query, frozen K / V; dq is linear in dO
            ipb = s["ip"]
            dq_ip, _, _ = K.attention_bwd(s["q2"], ipb["k"], ipb["v"], ipb["o"], do2, batch_q=n_img, lq=L, lk=ipb["li"],
                                          heads=heads, head_dim=d, kv_group=s["group2"], need_dkv=False, lse=ipb["lse"])
            dn2 = K.gemm(dq2, w["w_q2_t"], residual=K.gemm(dq_ip, w["w_q2_t"], out_scale=ipb["scale"]))
        else:
            dn2 = K.gemm(dq2, w["w_q2_t"])
        g1 = K.layernorm_bwd(s["x1"], dn2, w["g2"], b.eps, add=g2)                    # dL/dx1
        # self-attention + cross-frame adapter attention (i2v:444-501)
        do1, doa = K.gemm(g1, w["w_o1_t"]), K.gemm(g1, w["w_oa_t"])
        d_wout = wgrad(g1, s["oa"])                                                   # i2v_adapter.to_out.0.weight
        d_bout = K.colsum(g1)
        dq1, dk1, dv1 = K.attention_bwd(s["q1"], s["k1"], s["v1"], s["o1"], do1, batch_q=n_img, lq=L, lk=L, heads=heads,
                                        head_dim=d, lse=s["lse1"])
        dqa, dk0, dv0 = K.attention_bwd(s["qa"], s["k0"], s["v0"], s["oa"], doa, batch_q=n_img, lq=L, lk=L, heads=heads,
                                        head_dim=d, kv_group=F, lse=s["lsea"])        # dK0 / dV0 summed over the frames
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
    """packed weights of the input-gradient convolution: W'[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx] (made once per weight
    version: the convolutions are frozen)."""
    return _memo.get(("conv_dgrad", cout_pad), (weight,),
                     lambda: pack_conv3x3(weight.detach().transpose(0, 1).flip(2, 3), cin_pad=cout_pad))


def _t(w):
    return _memo.get("t", (w,), lambda: w16(w.detach().t()))


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
            sw = m.conv_shortcut.weight
            wst = _memo.get("shortcut_t", (sw,), lambda: w16(sw.detach().reshape(m.out_channels, m.in_channels).t()))   # [Cin, Cout]
            dx = K.gemm(g2d, _memo.get(("shortcut_t_lo", c1), (sw,), lambda: wst[:c1].contiguous()),
                        residual=dx.view(-1, c1)).view(n, hh, ww, c1)
            if x2 is not None:
                c2 = x2.shape[3]
                dx2 = K.gemm(g2d, _memo.get(("shortcut_t_hi", c1), (sw,), lambda: wst[c1:].contiguous()),
                             residual=dx2.view(-1, c2)).view(n, hh, ww, c2)
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
    def forward(self, x, num_frames, ctx_text, ctx_ip=None):
        m, p = self.m, self.m.packed()
        n, hh, ww, c = x.shape
        nrm = K.groupnorm(x, p["g"], p["b"], m.groups, 1e-6)
        t = K.gemm(nrm.view(-1, c), p["wi"], p["bi"])
        t = self.block.forward(t, n, hh * ww, num_frames, ctx_text, ctx_ip)
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


def _unit_affine(c, device):
    """(ones, zeros) fp16 [c]: a norm kernel run with them returns x-hat, the factor of the gain gradient."""
    key = (c, str(device))
    hit = _unit_affine.cache.get(key)
    if hit is None:
        hit = _unit_affine.cache[key] = (torch.ones((c,), dtype=f16, device=device), torch.zeros((c,), dtype=f16, device=device))
    return hit


_unit_affine.cache = {}


class MotionModuleTrainer:
    """TransformerTemporalModel: clip-wide GroupNorm -> proj_in -> [LN + PE -> self-attention over the frames of
    a pixel -> + residual] x 2 -> LN -> GEGLU FF -> proj_out -> + residual, on rows in (batch, pixel, frame) order.

    train_weights=True is `--update_motion_modules` (train_image_to_video.py:452, 669 -> unet:984-999 with
    freeze_animatediff=False: every parameter of every motion module trains): `backward` then also returns the gradient of
    all 26 tensors of the module keyed by their state-dict names.  Weight gradients are dY^T X GEMMs over the channel-major
    copies (`wgrad`), bias / shift gradients column sums, gain gradients sum dY o x-hat (`K.colsum_prod`); the normalised
    inputs are recomputed from the kept layer inputs instead of being stored."""

    def __init__(self, mm, train_weights=False):
        if len(mm.transformer_blocks) != 1:
            raise NotImplementedError("one temporal block per motion module (AnimateDiff v1.5)")
        self.m = mm
        self.train_weights = bool(train_weights)

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
        y = K.geglu(h)
        t3 = K.gemm(y, ff["w2"], ff["b2"], residual=t)
        self.saved = (x, stages, t, h, t3 if self.train_weights else None)
        return K.gemm(t3, p["wo"], p["bo"], residual=x.view(-1, c), store=I2V_STORE_ROWPERM, frames=F, hw=hw).view(n, hh, ww, c)

    @torch.no_grad()
    def backward(self, g, num_frames, loss_scale=1.0):
        """d / d input; with train_weights also {state-dict name: fp32 gradient / loss_scale} as the second value."""
        m, p = self.m, self.m.packed()
        blk = m.transformer_blocks[0]
        q, ff = blk.packed(), blk.ff.packed()
        x, stages, t2, h, t3 = self.saved
        n, hh, ww, c = x.shape
        hw, clips, F = hh * ww, n // num_frames, num_frames
        n_pixels = clips * hw
        tw = self.train_weights
        inv = 1.0 / float(loss_scale)
        grads = {}
        ones, zeros = _unit_affine(c, x.device)

        def linear(name, dy, xin, bias=True):
            grads[f"{name}.weight"] = wgrad(dy, xin) * inv
            if bias:
                grads[f"{name}.bias"] = K.colsum(dy) * inv

        def norm(name, dy, xhat):
            grads[f"{name}.weight"] = K.colsum_prod(dy, xhat) * inv
            grads[f"{name}.bias"] = K.colsum(dy) * inv

        gp = K.permute_rows(g.view(-1, c), clips, F, hw, True)
        if tw:
            linear("proj_out", gp, t3)
        dt = K.gemm(gp, _t(p["wo"]))                                                  # dL/dt3
        dy = K.gemm(dt, _t(ff["w2"]))
        dh = K.geglu_bwd(h, dy)
        if tw:
            linear("transformer_blocks.0.ff.net.2", dt, K.geglu(h))
            n3 = K.layernorm(t2, q["g3"], q["b3"], blk.eps)
            inner = dh.shape[1] // 2
            # the packed projection interleaves (value_i, gate_i) rows (`pack_geglu`); the reference keeps [values ; gates]
            unpack = lambda t_: torch.cat([t_[0::2], t_[1::2]], dim=0)
            grads["transformer_blocks.0.ff.net.0.proj.weight"] = unpack(wgrad(dh, n3)) * inv
            grads["transformer_blocks.0.ff.net.0.proj.bias"] = unpack(K.colsum(dh)) * inv
            assert grads["transformer_blocks.0.ff.net.0.proj.weight"].shape[0] == 2 * inner
        dn3 = K.gemm(dh, _t(ff["w1"]))
        if tw:
            norm("transformer_blocks.0.norm3", dn3, K.layernorm(t2, ones, zeros, blk.eps))
        dt = K.layernorm_bwd(t2, dn3, q["g3"], blk.eps, add=dt)
        for i in (2, 1):
            t_in, qk, v, o = stages[i - 1]
            do = K.gemm(dt, _t(q[f"wo{i}"]))
            dq, dk, dv = K.attention_bwd(qk[:, :c], qk[:, c:], v, o, do, batch_q=n_pixels, lq=F, lk=F, heads=blk.heads,
                                         head_dim=blk.dim_head, scale=blk.dim_head ** -0.5)
            dnl = K.gemm(dq, _t(q[f"wqk{i}"]), a2=dk)                            # [dq | dk] [Wq ; Wk]
            dnl = K.gemm(dv, _t(q[f"wv{i}"]), residual=dnl, out=dnl)
            if tw:
                a = f"transformer_blocks.0.attn{i}"
                linear(f"{a}.to_out.0", dt, o)
                nl = K.layernorm(t_in, q[f"g{i}"], q[f"b{i}"], blk.eps, pe=q["pe"], pe_period=F)
                linear(f"{a}.to_q", dq, nl, bias=False)
                linear(f"{a}.to_k", dk, nl, bias=False)
                linear(f"{a}.to_v", dv, nl, bias=False)
                norm(f"transformer_blocks.0.norm{i}", dnl, K.layernorm(t_in, ones, zeros, blk.eps))
            dt = K.layernorm_bwd(t_in, dnl, q[f"g{i}"], blk.eps, add=dt)
        dnp = K.gemm(dt, _t(p["wi"]))
        dn = K.permute_rows(dnp, clips, F, hw, False).view(n, hh, ww, c)
        if tw:
            cin = x.shape[3]
            o1, z1 = _unit_affine(cin, x.device)
            nrm = K.groupnorm(x, p["g"], p["b"], m.groups, 1e-6, frames_per_stat=F)
            linear("proj_in", dt, K.permute_rows(nrm.view(-1, cin), clips, F, hw, True))
            norm("norm", dn.view(-1, cin), K.groupnorm(x, o1, z1, m.groups, 1e-6, frames_per_stat=F).view(-1, cin))
        dx = K.groupnorm_bwd(x, dn, p["g"], p["b"], m.groups, 1e-6, frames_per_stat=F)
        self.saved = None
        dx = K.add(dx, g)
        return (dx, grads) if tw else dx


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


# ------------------------------------------------------------------------------------------------------------------------
# The whole training step of the reference loop (src/train_image_to_video.py:839-884) for the adapter parameters.
class UNetAdapterTrainer:
    """forward(...) = UNetMotionCrossFrameAttnModel.forward with enable_cross_frame_attn=True (unet:1289-1451) in the un-fused
    training form, keeping every layer's input; backward(target, ...) = the loss without the first frame
    (train_image_to_video.py:848-856) seeded at the prediction and walked back through every frozen layer, collecting
    d loss / d {i2v_adapter.to_q.weight, to_out.0.weight, to_out.0.bias} of all spatial blocks (unet:979-1026) as un-scaled
    fp32 tensors keyed by their state-dict names."""

    def __init__(self, unet, update_motion_modules=False):
        """update_motion_modules: `--update_motion_modules` (train_image_to_video.py:452, 669): the 21 motion modules train
        too (unet:984-999 with freeze_animatediff=False) and `backward` also returns their parameters' gradients."""
        from .blocks import DownBlockMotion, UpBlockMotion  # noqa: F401  (attention-free blocks: resnet -> motion)
        self.unet = unet
        self.update_motion_modules = bool(update_motion_modules)
        names = {m: n for n, m in unet.named_modules()}
        self.tape_layout = []          # (kind, trainer, extra) in forward order; built once, replayed per step
        self._mk = dict(resnet=ResnetTrainer, t2d=Transformer2DTrainer, motion=MotionModuleTrainer, down=DownsampleTrainer,
                        up=UpsampleTrainer)
        self._names = names
        self._trainers = {}

    def _tr(self, kind, module):
        t = self._trainers.get(module)
        if t is None:
            kw = dict(train_weights=self.update_motion_modules) if kind == "motion" else {}
            t = self._trainers[module] = self._mk[kind](module, **kw)
        return t

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None):
        """sample (B, F, C, H, W) on the GPU; added_cond_kwargs={"image_embeds": ...} when the IP-Adapter is installed (the
        reference's training step passes them, train_image_to_video.py:843); returns the prediction as tokens fp32
        [B * F, H, W, 8] (4 channels + zeros)."""
        u = self.unet
        b, F, c, hh, ww = sample.shape
        p = u.packed()
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.to(device=sample.device, dtype=torch.float32).reshape(-1).expand(b).contiguous()
        temb_act = K.silu(u._embed_time(t))
        ctx = encoder_hidden_states.to(f16).contiguous()
        ctx_ip = u._project_image_embeds(added_cond_kwargs)                      # unet:1346-1352 (frozen ImageProjection)
        rows = lambda r: K.gemm(temb_act, r.packed()["wt"], r.packed()["bt"])
        tape, res, res_ids = [], [], []
        x = K.conv3x3(K.nchw_to_tokens(sample.reshape(b * F, c, hh, ww), p["cin_pad"]), p["w_in"], p["b_in"])
        push = lambda v: (tape.append(("push", None, len(res_ids))), res.append(v), res_ids.append(len(res_ids)))

        def layer(x, resnet, attn, motion, skip=None, skip_id=None):
            r = self._tr("resnet", resnet)
            x = r.forward(x, rows(resnet), x2=skip)
            tape.append(("resnet", r, skip_id))
            if attn is not None:
                a = self._tr("t2d", attn)
                x = a.forward(x, F, ctx, ctx_ip)
                tape.append(("t2d", a, self._names[attn]))
            if motion is not None:
                mm = self._tr("motion", motion)
                x = mm.forward(x, F)
                tape.append(("motion", mm, self._names[motion]))
            return x

        push(x)
        for blk in u.down_blocks:
            attns = getattr(blk, "attentions", [None] * len(blk.resnets))
            for resnet, attn, motion in zip(blk.resnets, attns, blk.motion_modules):
                x = layer(x, resnet, attn, motion)
                push(x)
            if blk.downsamplers is not None:
                for d in blk.downsamplers:
                    tr = self._tr("down", d)
                    x = tr.forward(x)
                    tape.append(("down", tr, None))
                push(x)
        mid = u.mid_block
        x = layer(x, mid.resnets[0], None, None)
        for attn, resnet, motion in zip(mid.attentions, mid.resnets[1:], mid.motion_modules):
            a = self._tr("t2d", attn)
            x = a.forward(x, F, ctx, ctx_ip)
            tape.append(("t2d", a, self._names[attn]))
            mm = self._tr("motion", motion)
            x = mm.forward(x, F)
            tape.append(("motion", mm, self._names[motion]))
            x = layer(x, resnet, None, None)
        for blk in u.up_blocks:
            attns = getattr(blk, "attentions", [None] * len(blk.resnets))
            for resnet, attn, motion in zip(blk.resnets, attns, blk.motion_modules):
                skip, sid = res.pop(), res_ids.pop()
                x = layer(x, resnet, attn, motion, skip=skip, skip_id=sid)
            if blk.upsamplers is not None:
                for up in blk.upsamplers:
                    tr = self._tr("up", up)
                    x = tr.forward(x)
                    tape.append(("up", tr, None))
        a = K.groupnorm(x, p["g_out"], p["be_out"], u.config.norm_num_groups, u.config.norm_eps, silu=True)
        # conv_out padded to 8 output channels (the head's tokens are [.., 8] fp16); frozen: the padded weight, its packed
        # form and (in backward) its input-gradient form are made once per weight version, not once per step
        cw, cb = u.conv_out.weight, u.conv_out.bias

        def _pad_w():
            w = torch.zeros((8,) + tuple(cw.shape[1:]), dtype=cw.dtype, device=cw.device)
            w[: cw.shape[0]] = cw.detach()
            return w

        def _pad_b():
            bb = torch.zeros((8,), dtype=f16, device=cb.device)
            bb[: cb.shape[0]] = cb.detach().to(f16)
            return bb

        w_out = _memo.get("conv_out_pad8", (cw,), _pad_w)
        # (fp32 out of the accumulators: the loss is F.mse_loss(model_pred.float(), target.float()), train_image_to_video.py:848)
        y = K.conv3x3(a, _memo.get("conv_out_pad8_packed", (cw,), lambda: pack_conv3x3(w_out)),
                      _memo.get("conv_out_bias_pad8", (cb,), _pad_b), out_f32=True)
        self.saved = dict(tape=tape, x_last=x, y=y, F=F, w_out=w_out)
        return y

    @torch.no_grad()
    def backward(self, target, loss_scale=2.0 ** 12):
        """target (B, F, C, H, W) (the noise, train_image_to_video.py:830-831).  Returns (loss, {name: fp32 gradient})."""
        u, s = self.unet, self.saved
        p = u.packed()
        y, F = s["y"], s["F"]
        n, hh, ww, _ = y.shape
        b, c = n // F, target.shape[2]
        tgt = torch.zeros((n, hh * ww, 8), dtype=torch.float32, device=y.device)          # fp32 target tokens, 4 + 4 zero channels
        tgt[:, :, :c] = target.to(device=y.device, dtype=torch.float32).reshape(n, c, hh * ww).transpose(1, 2)
        count = float(b * (F - 1) * c * hh * ww)
        seed, rowsq = K.masked_mse_grad_f32(y.view(n, hh * ww, 8), tgt, F, 2.0 * loss_scale / count)
        loss = rowsq.sum() / count         # (the row sums on the GPU in fp32; their total is plumbing)
        g = K.conv3x3(seed.view(n, hh, ww, 8), conv_dgrad_weight(s["w_out"], cout_pad=8))
        g = K.groupnorm_bwd(s["x_last"], g, p["g_out"], p["be_out"], u.config.norm_num_groups, u.config.norm_eps, silu=True)
        grads, dskip = {}, {}
        tape = s["tape"]
        first_t2d = min(i for i, op in enumerate(tape) if op[0] == "t2d")
        for i in range(len(tape) - 1, first_t2d - 1, -1):          # nothing trains in front of the first transformer
            kind, tr, extra = tape[i]
            if kind == "push":
                if extra in dskip:
                    g = K.add(g, dskip.pop(extra))
            elif kind == "resnet":
                g, dx2 = tr.backward(g)
                if extra is not None:
                    dskip[extra] = dx2
            elif kind == "t2d":
                g, pg = tr.backward(g, loss_scale)
                for k, v in pg.items():
                    grads[f"{extra}.transformer_blocks.0.{k}"] = v
            elif kind == "motion":
                if self.update_motion_modules:
                    g, pg = tr.backward(g, F, loss_scale)
                    for k, v in pg.items():
                        grads[f"{extra}.{k}"] = v
                else:
                    g = tr.backward(g, F)
            else:
                g = tr.backward(g)
        self.saved = None
        return loss, grads


class AdapterOptimizer:
    """AdamW + clip_grad_norm_ on the adapter parameters (train_image_to_video.py:716-724, 876-882), data-parallel:
    fp32 master copies, gradients and both moments live in FLAT buckets in state-dict order; `step(grads)` copies the
    gradients into the bucket, sums it over the ranks with ONE all-reduce (RCCL over xGMI under backend "nccl"; the mean is
    folded into the update's gradient coefficient), takes the global norm and applies the update with two kernels, then
    writes the fp16 parameters back.  No per-parameter launches, no host round trip for the clip coefficient."""

    def __init__(self, unet, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0, process_group=None,
                 gradient_accumulation_steps=1, use_ema=False, ema_decay=0.9999, update_motion_modules=False):
        """update_motion_modules: the motion modules' parameters join the buckets (`--update_motion_modules`,
        train_image_to_video.py:452, 669; unet:984-999).  gradient_accumulation_steps: `accelerator.accumulate` (train_image_to_video.py:486, 785) -- `step` adds each
        micro-batch's gradients / N into the bucket and updates on every N-th call.  use_ema: an exponential moving average of
        the TRAINED parameters in a fourth flat bucket (`--use_ema`, :673-677, 888-889; diffusers EMAModel's schedule: decay_t =
        min(ema_decay, (1 + t) / (10 + t)) with t = updates - 1, 0 on the first).  (The reference builds its EMA over a
        `UNet2DConditionModel`'s parameter list and steps it with the motion UNet's, :674-677 vs :889 -- lists that do not match;
        the EMA here covers the parameters that train, read back with `ema_state_dict()`.)"""
        self.unet, self.group = unet, process_group
        self.accum_steps, self._micro = int(gradient_accumulation_steps), 0
        if self.accum_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        self.use_ema, self.ema_decay, self.ema_updates = bool(use_ema), float(ema_decay), 0
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.update_motion_modules = bool(update_motion_modules)
        self.names = [n for n, _ in unet.named_parameters()
                      if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n                    # unet:1001-1006
                      or (self.update_motion_modules and ".motion_modules." in n)]                   # unet:984-999
        params = dict(unet.named_parameters())
        self.params = [params[n] for n in self.names]
        self.offsets, off = {}, 0
        for n, prm in zip(self.names, self.params):
            self.offsets[n] = (off, prm.numel())
            off += prm.numel()
        dev = self.params[0].device
        self.master = torch.cat([prm.detach().float().reshape(-1) for prm in self.params]).contiguous()
        self.grad = torch.zeros_like(self.master)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.master), torch.zeros_like(self.master)
        self.norm_sq = torch.zeros((1,), dtype=torch.float32, device=dev)
        self._partials = torch.zeros((1024,), dtype=torch.float32, device=dev)
        # device-side state of the guarded step: steps actually applied (the bias correction uses THIS count) and whether
        # the last step was skipped because its gradients held inf / NaN (fp16 activation gradients under a static scale)
        self.applied_steps = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.found_inf = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.step_count = 0            # steps requested (applied + skipped)
        self._micro_grad = torch.zeros_like(self.master) if self.accum_steps > 1 else None
        self.ema = self.master.clone() if self.use_ema else None

    def last_step_skipped(self) -> bool:
        """did the last `step` find inf / NaN gradients and leave everything unchanged?  (one host read; the trainer halves
        its loss scale on True, as accelerate's GradScaler does for the reference's fp16 run)"""
        return bool(self.found_inf.item())

    def fill_gradients(self, grads):
        for n in self.names:
            off, cnt = self.offsets[n]
            self.grad[off: off + cnt].copy_(grads[n].reshape(-1))

    def reduce_gradients(self):
        """sum the bucket over the data-parallel ranks; returns the divisor that turns the sum into the mean."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
            return dist.get_world_size(self.group)
        return 1

    def ema_state_dict(self):
        """{name: fp32 EMA tensor} of the trained parameters (what `ema_unet.copy_to(unet.parameters())` would load,
        train_image_to_video.py:904-907, 946-947)."""
        if self.ema is None:
            raise ValueError("use_ema=False")
        return {n: self.ema[off: off + cnt].view_as(prm).clone() for n, prm, (off, cnt) in
                ((n, prm, self.offsets[n]) for n, prm in zip(self.names, self.params))}

    @torch.no_grad()
    def step(self, grads):
        """one micro-batch's gradients.  Returns True when an optimiser step was taken (accelerate's `sync_gradients`), False
        when the gradients were only accumulated."""
        if self.accum_steps > 1:
            # accumulate(): the N micro-batch losses are averaged, i.e. every gradient enters with weight 1 / N
            bucket, self.grad = self.grad, self._micro_grad
            self.fill_gradients(grads)
            self.grad = bucket
            if self._micro == 0:
                self.grad.zero_()
            K.axpby(self.grad, self._micro_grad, 1.0, 1.0 / self.accum_steps)
            self._micro += 1
            if self._micro < self.accum_steps:
                return False
            self._micro = 0
        else:
            self.fill_gradients(grads)
        world = self.reduce_gradients()
        self.step_count += 1
        # one call: fixed-order gradient norm (identical on every rank: the clip coefficients cannot drift apart), overflow
        # check, clip + AdamW.  With inf / NaN in the bucket nothing moves -- masters, moments, bias-correction step -- and
        # the fp16 parameters below are rewritten with their old values.
        K.adamw_guarded_step(self.master, self.grad, self.exp_avg, self.exp_avg_sq, lr=self.lr, betas=self.betas,
                             eps=self.eps, weight_decay=self.wd, grad_coef=1.0 / world, max_norm=self.max_norm,
                             partials=self._partials, norm_sq=self.norm_sq, applied_steps=self.applied_steps,
                             found_inf=self.found_inf)
        for n, prm in zip(self.names, self.params):
            off, cnt = self.offsets[n]
            prm.copy_(self.master[off: off + cnt].view_as(prm))   # (on the parameter itself: its version counter moves)
        if self.use_ema:
            # (a step skipped for inf / NaN gradients leaves the masters as they were: the EMA then averages the same value)
            self.ema_updates += 1
            t = max(0, self.ema_updates - 1)
            decay = 0.0 if t <= 0 else min(self.ema_decay, (1.0 + t) / (10.0 + t))
            K.axpby(self.ema, self.master, decay, 1.0 - decay)
        return True
