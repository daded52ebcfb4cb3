"""SURVEY 8 f4, first vertical slice: the backward of one I2VAdapterTransformerBlock on the HIP kernels against torch
autograd over the fp32 CPU oracle block -- the gradients the reference's optimiser receives for the block's trainable
parameters (unet:979-1026: i2v_adapter.to_q / to_out) and the gradient it passes on to the earlier layers, under the
reference's loss (train_image_to_video.py:848-856: MSE without the first frame of each clip).

Kernel-level cases first (attention backward incl. the kv_group reduction of dK0 / dV0, LayerNorm / GEGLU backward, the
transposes and sums), then the block.  Tolerance: 2.5e-3 (r4: 2x the measured) of the largest reference gradient entry (fp16 operands, fp32
accumulation; the P / dS operands of the attention backward are fp16 like the forward's P)."""
import pytest
import torch

from tests.parity import compare, randomize_adapter_out_, round_fp16_

pytestmark = pytest.mark.gpu
GRAD_REL_TOL = 2.5e-3     # measured 2.5e-4 .. 1.2e-3 of the largest reference gradient entry


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def h(t):
    return t.half().float()


@pytest.mark.parametrize("d,lq,lk,group,need_dkv", [(40, 256, 256, 1, True), (40, 128, 128, 4, True), (80, 64, 64, 2, True),
                                                    (160, 64, 64, 1, True), (40, 96, 77, 2, False), (64, 32, 160, 1, True),
                                                    (40, 512, 640, 1, True), (80, 576, 512, 2, True),
                                                    # a kv_group dealt to 8 / 4 workgroups per key block (kv_partitions: fp32 partials + sum)
                                                    (40, 256, 640, 8, True), (64, 128, 512, 4, True)])
def test_attention_backward_vs_autograd(dev, d, lq, lk, group, need_dkv):
    """dQ, dK, dV of softmax(q k^T / sqrt d) v against autograd; kv_group > 1 = several query batches share one K / V (the
    cross-frame attention: dK / dV are the sums over the group); lk = 77 is the text context (dQ only, masked tail);
    sequences >= 512 take the two-tiles-per-wave forms of both sweeps; with a kv_group and few key blocks the group's query batches
    are dealt to several workgroups per key block (kernels.dkv_partitions)."""
    K = pkg().kernels
    heads, bkv = 2, 2
    bq, C = bkv * group, heads * d
    g = torch.Generator().manual_seed(d + lq + group)
    q = h(torch.randn(bq, lq, C, generator=g)).requires_grad_()
    k = h(torch.randn(bkv, lk, C, generator=g)).requires_grad_()
    v = h(torch.randn(bkv, lk, C, generator=g)).requires_grad_()
    do = h(torch.randn(bq, lq, C, generator=g) * 0.5)
    split = lambda t, b, l: t.view(b, l, heads, d).transpose(1, 2)
    kk = split(k, bkv, lk).repeat_interleave(group, dim=0)
    vv = split(v, bkv, lk).repeat_interleave(group, dim=0)
    o = torch.nn.functional.scaled_dot_product_attention(split(q, bq, lq), kk, vv).transpose(1, 2).reshape(bq, lq, C)
    o.backward(do)
    to = lambda t: t.detach().half().to(dev).reshape(-1, C)
    qd, kd, vd = to(q), to(k), to(v)
    od = K.attention(qd, kd, K.transpose_tokens(vd, lk), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    compare(od, o.reshape(-1, C), rel=3e-3, name="attention forward (training layout)")
    dq, dk, dv = K.attention_bwd(qd, kd, vd, od, to(do), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group,
                                 need_dkv=need_dkv)
    compare(dq, q.grad.reshape(-1, C), rel=GRAD_REL_TOL, name=f"dQ d={d} lq={lq} lk={lk} group={group}")
    if group >= 4 and lk >= 512:
        assert K.dkv_partitions(bq, group, heads, d, lq, lk) > 1
    if need_dkv:
        compare(dk, k.grad.reshape(-1, C), rel=GRAD_REL_TOL, name=f"dK d={d} group={group}")
        compare(dv, v.grad.reshape(-1, C), rel=GRAD_REL_TOL, name=f"dV d={d} group={group}")
        dq2, dk2, dv2 = K.attention_bwd(qd, kd, vd, od, to(do), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d,
                                        kv_group=group)
        assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)      # no atomics: run-to-run identical
    else:
        assert dk is None and dv is None


@pytest.mark.parametrize("lq,lk", [(70, 77), (640, 520)])
def test_attention_lse(dev, lq, lk):
    K = pkg().kernels
    g = torch.Generator().manual_seed(3)
    bq, heads, d = 3, 4, 40
    q, k = h(torch.randn(bq, lq, heads * d, generator=g)), h(torch.randn(bq, lk, heads * d, generator=g))
    s = torch.einsum("blhd,bmhd->bhlm", q.view(bq, lq, heads, d), k.view(bq, lk, heads, d)) * d ** -0.5
    ref = torch.logsumexp(s, dim=-1) * 1.4426950408889634
    got = K.attention_lse(q.half().to(dev).view(-1, heads * d), k.half().to(dev).view(-1, heads * d), batch_q=bq, lq=lq, lk=lk,
                          heads=heads, head_dim=d)
    compare(got, ref, rel=1e-3, name="attention log2-sum-exp")
    # the same written by the forward pass itself (i2v_attn_params.lse, ABI 6), every head_dim class of the UNet
    for d2, heads2 in ((40, 4), (80, 2), (160, 2), (64, 2)):
        C2 = heads2 * d2
        q2, k2 = h(torch.randn(bq, lq, C2, generator=g) * 1.5), h(torch.randn(bq, lk, C2, generator=g) * 1.5)
        v2 = h(torch.randn(bq, lk, C2, generator=g))
        s2 = torch.einsum("blhd,bmhd->bhlm", q2.view(bq, lq, heads2, d2), k2.view(bq, lk, heads2, d2)) * d2 ** -0.5
        to = lambda t: t.half().to(dev).reshape(-1, C2)
        o_plain = K.attention(to(q2), to(k2), K.transpose_tokens(to(v2), lk), batch_q=bq, lq=lq, lk=lk, heads=heads2, head_dim=d2)
        o_fused, lse_fused = K.attention(to(q2), to(k2), K.transpose_tokens(to(v2), lk), batch_q=bq, lq=lq, lk=lk, heads=heads2,
                                         head_dim=d2, return_lse=True)
        assert torch.equal(o_plain, o_fused)
        compare(lse_fused, torch.logsumexp(s2, dim=-1) * 1.4426950408889634, rel=1e-3, name=f"fused log2-sum-exp d={d2}")
        with pytest.raises(ValueError, match="accumulate"):
            K.attention(to(q2), to(k2), K.transpose_tokens(to(v2), lk), batch_q=bq, lq=lq, lk=lk, heads=heads2, head_dim=d2,
                        out=o_plain, accumulate=True, return_lse=True)


@pytest.mark.parametrize("d,lq,lk,group,amp", [(40, 192, 256, 2, 3.0), (80, 64, 128, 1, 3.0), (40, 512, 640, 1, 2.5)])
def test_attention_backward_large_logits(dev, d, lq, lk, group, amp):
    """the log-sum-exp recompute and both backward sweeps with logits of 20 - 50 (nearly one-hot rows)"""
    K = pkg().kernels
    heads, bkv = 2, 2
    bq, C = bkv * group, heads * d
    g = torch.Generator().manual_seed(d + lq + int(10 * amp))
    q = h(torch.randn(bq, lq, C, generator=g) * amp).requires_grad_()
    k = h(torch.randn(bkv, lk, C, generator=g) * amp).requires_grad_()
    v = h(torch.randn(bkv, lk, C, generator=g)).requires_grad_()
    do = h(torch.randn(bq, lq, C, generator=g) * 0.5)
    split = lambda t, b, l: t.view(b, l, heads, d).transpose(1, 2)
    kk = split(k, bkv, lk).repeat_interleave(group, dim=0)
    vv = split(v, bkv, lk).repeat_interleave(group, dim=0)
    o = torch.nn.functional.scaled_dot_product_attention(split(q, bq, lq), kk, vv).transpose(1, 2).reshape(bq, lq, C)
    o.backward(do)
    to = lambda t: t.detach().half().to(dev).reshape(-1, C)
    qd, kd, vd = to(q), to(k), to(v)
    s = torch.einsum("blhd,bmhd->bhlm", q.detach().view(bq, lq, heads, d), kk.transpose(1, 2)) * d ** -0.5
    lse = K.attention_lse(qd, kd, batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    compare(lse, torch.logsumexp(s, dim=-1) * 1.4426950408889634, rel=2e-3, name="log2-sum-exp, large logits")
    od = K.attention(qd, kd, K.transpose_tokens(vd, lk), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    compare(od, o.reshape(-1, C), rel=6e-4 * amp * amp, name="attention forward, large logits")
    dq, dk, dv = K.attention_bwd(qd, kd, vd, od, to(do), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group)
    # ... and with the log-sum-exp the forward pass wrote (what the trainer does)
    od2, lse_f = K.attention(qd, kd, K.transpose_tokens(vd, lk), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group,
                             return_lse=True)
    compare(lse_f, torch.logsumexp(s, dim=-1) * 1.4426950408889634, rel=2e-3, name="fused log2-sum-exp, large logits")
    dq_f, dk_f, dv_f = K.attention_bwd(qd, kd, vd, od2, to(do), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d, kv_group=group,
                                       lse=lse_f)
    rel = GRAD_REL_TOL * amp * amp
    compare(dq_f, q.grad.reshape(-1, C), rel=rel, name="dQ from the forward's log-sum-exp")
    compare(dk_f, k.grad.reshape(-1, C), rel=rel, name="dK from the forward's log-sum-exp")
    compare(dv_f, v.grad.reshape(-1, C), rel=rel, name="dV from the forward's log-sum-exp")
    compare(dq, q.grad.reshape(-1, C), rel=rel, name="dQ, large logits")
    compare(dk, k.grad.reshape(-1, C), rel=rel, name="dK, large logits")
    compare(dv, v.grad.reshape(-1, C), rel=rel, name="dV, large logits")


def test_trainer_operand_cache_follows_the_trainable_weights(dev):
    """the copies derived from weights (transposes, fused projections, flipped conv kernels) are made once for the frozen
    layers; whatever depends on the adapter's to_q / to_out is remade on every step, also when the optimiser writes the
    parameter through `.data` (no version bump)."""
    p = pkg()
    from i2v_adapter_unofficial_amd.training import AdapterBlockTrainer
    torch.manual_seed(3)
    dim, heads = 64, 2
    m = p.I2VAdapterTransformerBlock(dim, heads, dim // heads, dropout=0.0, cross_attention_dim=48)
    m = m.to(device=dev, dtype=torch.float16).eval()
    tr = AdapterBlockTrainer(m)
    w0 = tr._weights()
    qa_before = w0["w_qkq"][2 * dim:].clone()
    m.i2v_adapter.to_q.weight.data.mul_(2.0)                  # what an optimiser step through .data does
    m.i2v_adapter.to_out[0].weight.data.add_(1.0)
    w1 = tr._weights()
    assert torch.equal(w1["w_qkq"][2 * dim:], (qa_before.float() * 2).half())
    assert torch.equal(w1["w_oa_t"], m.i2v_adapter.to_out[0].weight.detach().t().contiguous())
    assert torch.equal(w1["w_o_dual"][:, dim:], m.i2v_adapter.to_out[0].weight.detach())
    assert w1["w2_t"] is w0["w2_t"] and w1["w_qk1_t"] is w0["w_qk1_t"] and w1["w1"] is w0["w1"]      # frozen: made once
    m.ff.net[2].weight.mul_(0.5) if not m.ff.net[2].weight.requires_grad else m.ff.net[2].weight.data.mul_(0.5)
    with torch.no_grad():
        m.attn1.to_q.weight.mul_(0.5)                          # an in-place write bumps the version: the copy is remade
    w2 = tr._weights()
    assert w2["w_qk1_t"] is not w0["w_qk1_t"]
    assert torch.equal(w2["w_qk1_t"][:, :dim], m.attn1.to_q.weight.detach().t().contiguous())


def test_geglu_forward_from_stored_preactivation(dev):
    """i2v_geglu_f16 against torch and against the GEMM's fused GEGLU epilogue on the same operands"""
    K = pkg().kernels
    g = torch.Generator().manual_seed(6)
    rows, C, inner = 300, 64, 128
    x = h(torch.randn(rows, C, generator=g) * 2)
    w = h(torch.randn(2 * inner, C, generator=g) / 8)
    b = h(torch.randn(2 * inner, generator=g))
    xd, wd, bd = x.half().to(dev), w.half().to(dev), b.half().to(dev)
    hd = K.gemm(xd, wd, bd)
    y = K.geglu(hd)
    hf = hd.float().cpu()
    compare(y, hf[:, 0::2] * torch.nn.functional.gelu(hf[:, 1::2]), rel=1e-3, name="geglu forward")
    compare(y, K.gemm(xd, wd, bd, epilogue=K.I2V_EPI_GEGLU).float().cpu(), rel=2e-3, name="geglu forward vs the fused epilogue")
    big = torch.tensor([[30.0, 40.0, -30.0, -40.0, 5.0, -20.0, 0.0, 0.0] * 2], device=dev).half()      # saturated gates
    yb = K.geglu(big).float().cpu()
    assert torch.isfinite(yb).all() and torch.allclose(yb[0, :4], torch.tensor([1200.0, 0.0, 0.0, 0.0]), atol=1e-3)


def test_layernorm_geglu_backward_and_sums(dev):
    K = pkg().kernels
    g = torch.Generator().manual_seed(5)
    rows, C = 70, 320
    x = (h(torch.randn(rows, C, generator=g)) * 2 + 0.5).requires_grad_()
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.1 * torch.randn(C, generator=g))
    dn, add = h(torch.randn(rows, C, generator=g)), h(torch.randn(rows, C, generator=g))
    torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5).backward(dn)
    got = K.layernorm_bwd(x.detach().half().to(dev), dn.half().to(dev), gamma.half().to(dev), 1e-5, add=add.half().to(dev))
    compare(got, x.grad + add, rel=2e-3, name="LayerNorm backward (+ residual gradient)")
    # GEGLU: h interleaved (value, gate)
    inner = 64
    val, gate = h(torch.randn(rows, inner, generator=g)).requires_grad_(), h(torch.randn(rows, inner, generator=g) * 2).requires_grad_()
    dy = h(torch.randn(rows, inner, generator=g))
    (val * torch.nn.functional.gelu(gate)).backward(dy)
    hh = torch.stack([val.detach(), gate.detach()], dim=2).reshape(rows, 2 * inner)
    dh = K.geglu_bwd(hh.half().to(dev), dy.half().to(dev)).float().cpu().view(rows, inner, 2)
    compare(dh[..., 0], val.grad, rel=2e-3, name="GEGLU backward (value)")
    compare(dh[..., 1], gate.grad, rel=2e-3, name="GEGLU backward (gate)")
    # transposes / sums / loss seed
    t = h(torch.randn(3 * 77, 48, generator=g))
    tt = K.transpose_tokens(t.half().to(dev), 77).float().cpu()
    assert tt.shape == (3, 48, 80) and torch.equal(tt[:, :, :77], t.view(3, 77, 48).transpose(1, 2)) and (tt[:, :, 77:] == 0).all()
    compare(K.colsum(dn.half().to(dev)), dn.sum(0), rel=1e-3, name="column sums")
    big = h(torch.randn(70000, 72, generator=g))                # 274 row blocks: the order of their sums is fixed
    bd = big.half().to(dev)
    c1, c2 = K.colsum(bd[:, :70]), K.colsum(bd[:, :70])
    assert torch.equal(c1, c2) and torch.allclose(c1.double().cpu(), big[:, :70].double().sum(0), rtol=1e-5, atol=1e-3)
    p1, p2 = K.colsum_prod(bd[:, :70], bd[:, 2:]), K.colsum_prod(bd[:, :70], bd[:, 2:])
    assert torch.equal(p1, p2) and torch.allclose(p1.double().cpu(), (big[:, :70].double() * big[:, 2:].double()).sum(0), rtol=1e-5, atol=1e-3)
    y, tg = h(torch.randn(8, 16, 32, generator=g)), h(torch.randn(8, 16, 32, generator=g))
    gr = K.masked_mse_grad(y.half().to(dev), tg.half().to(dev), frames=4, coef=0.25).float().cpu()
    ref = 0.25 * (y - tg)
    ref[0::4] = 0
    compare(gr, ref, rel=1e-3, name="masked MSE seed gradient")
    assert (gr[0] == 0).all() and (gr[4] == 0).all()
    # the fp32 form (the loss of train_image_to_video.py:848 is taken on .float() operands): fp16 seed, fp32 row sums of squares
    y32 = torch.randn(8, 16, 32, generator=g)
    t32 = torch.randn(8, 16, 32, generator=g)
    g32, rowsq = K.masked_mse_grad_f32(y32.to(dev), t32.to(dev), frames=4, coef=0.25)
    d = (y32 - t32).double()
    d[0::4] = 0
    assert g32.dtype == torch.float16 and torch.equal(g32.float().cpu(), (0.25 * d.float()).half().float())
    assert torch.allclose(rowsq.double().cpu().view(8, 16), (d * d).sum(-1), rtol=1e-6, atol=0)


@pytest.mark.parametrize("dim,heads,L,frames,clips", [(320, 8, 256, 4, 2), (128, 4, 64, 2, 1)])
def test_adapter_block_backward_vs_autograd(dev, dim, heads, L, frames, clips):
    """One spatial block (C = 320, head_dim 40: the 64 x 64 level's block at a 16 x 16 token grid): loss = MSE over the
    tokens of frames >= 1 (train_image_to_video.py:848-856) on the block output; d loss / d i2v_adapter.to_q.weight,
    d / d to_out.0.{weight, bias} and d / d hidden_states against torch autograd on the fp32 oracle block."""
    from oracle.i2v_adapter import I2VAdapterTransformerBlock as O
    p = pkg()
    from i2v_adapter_unofficial_amd.training import AdapterBlockTrainer
    torch.manual_seed(11)
    o = O(dim, heads, dim // heads, dropout=0.0, cross_attention_dim=96)
    randomize_adapter_out_(o)
    with torch.no_grad():
        for n in (o.norm1, o.norm2, o.norm3):
            n.weight.add_(0.1 * torch.randn_like(n.weight))
            n.bias.add_(0.1 * torch.randn_like(n.bias))
    o = round_fp16_(o).eval()
    m = p.I2VAdapterTransformerBlock(dim, heads, dim // heads, dropout=0.0, cross_attention_dim=96)
    m.load_state_dict(o.state_dict())
    m = m.to(device=dev, dtype=torch.float16).eval()
    g = torch.Generator().manual_seed(12)
    n_img = frames * clips
    x = h(torch.randn(n_img, L, dim, generator=g)).requires_grad_()
    ctx = h(torch.randn(n_img, 77, 96, generator=g))
    target = h(torch.randn(n_img, L, dim, generator=g))
    for prm in o.parameters():
        prm.requires_grad_(False)
    train = [o.i2v_adapter.to_q.weight, o.i2v_adapter.to_out[0].weight, o.i2v_adapter.to_out[0].bias]
    for prm in train:
        prm.requires_grad_(True)
    out = o(x, enable_cross_frame_attn=True, num_frames=frames, encoder_hidden_states=ctx)
    mask = torch.ones_like(out)
    mask.view(clips, frames, L, dim)[:, 0] = 0
    loss = ((out - target) ** 2 * mask).sum() / mask.sum()
    loss.backward()

    tr = AdapterBlockTrainer(m)
    xd = x.detach().half().to(dev).view(-1, dim)
    y = tr.forward(xd, n_img, L, frames, ctx.half().to(dev))
    compare(y.view(n_img, L, dim), out, rel=3e-3, name="training forward of the block")
    loss_scale = 2.0 ** 14
    coef = 2.0 * loss_scale / mask.sum().item()
    seed = p.kernels.masked_mse_grad(y.view(n_img, L, dim).contiguous(), target.half().to(dev), frames, coef)
    grads = tr.backward(seed.view(-1, dim), loss_scale=loss_scale)
    compare(grads["hidden_states"].float() / loss_scale, x.grad.reshape(-1, dim), rel=GRAD_REL_TOL,
            name=f"d loss / d hidden_states (C={dim})")
    compare(grads["i2v_adapter.to_q.weight"], train[0].grad, rel=GRAD_REL_TOL, name=f"d loss / d i2v_adapter.to_q.weight (C={dim})")
    compare(grads["i2v_adapter.to_out.0.weight"], train[1].grad, rel=GRAD_REL_TOL,
            name=f"d loss / d i2v_adapter.to_out.0.weight (C={dim})")
    compare(grads["i2v_adapter.to_out.0.bias"], train[2].grad, rel=GRAD_REL_TOL, name=f"d loss / d i2v_adapter.to_out.0.bias (C={dim})")
    again = tr.backward(seed.view(-1, dim), loss_scale=loss_scale)
    assert all(torch.equal(grads[k], again[k]) for k in grads)      # (the bias sums too: fixed-order block sums, no atomics)


# ---------------------------------------------------------------------------------------------- the frozen layers around
def _nchw_to_tok(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _pair(oracle_cls, hip_cls, dev, seed, *args, jitter=True, **kwargs):
    torch.manual_seed(seed)
    o = oracle_cls(*args, **kwargs)
    if jitter:
        with torch.no_grad():
            for mod in o.modules():
                if isinstance(mod, (torch.nn.GroupNorm, torch.nn.LayerNorm)):
                    mod.weight.add_(0.1 * torch.randn_like(mod.weight))
                    mod.bias.add_(0.1 * torch.randn_like(mod.bias))
    o = round_fp16_(o).eval()
    m = hip_cls(*args, **kwargs)
    m.load_state_dict(o.state_dict())
    for prm in o.parameters():
        prm.requires_grad_(False)
    return o, m.to(device=dev, dtype=torch.float16).eval()


@pytest.mark.parametrize("c1,c2,hw,fps,silu", [(64, 0, 16, 1, True), (320, 0, 32, 1, False), (64, 32, 8, 1, True), (64, 0, 8, 4, False)])
def test_groupnorm_backward(dev, c1, c2, hw, fps, silu):
    """spatial and clip-wide statistics, SiLU fused, channel-concatenated input (the up blocks' skip, unet:478)."""
    K = pkg().kernels
    g = torch.Generator().manual_seed(c1 + c2 + hw)
    n, C, groups = 8, c1 + c2, 16
    x = (h(torch.randn(n, C, hw, hw, generator=g)) * 1.5 + 3.0 * torch.randn(1, C, 1, 1, generator=g)).half().float().requires_grad_()
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.2 * torch.randn(C, generator=g))
    dy = h(torch.randn(n, C, hw, hw, generator=g))
    xin = x.view(n // fps, fps, C, hw, hw).permute(0, 2, 1, 3, 4) if fps > 1 else x
    y = torch.nn.functional.group_norm(xin, groups, gamma, beta, 1e-5)
    if fps > 1:
        y = y.permute(0, 2, 1, 3, 4).reshape(n, C, hw, hw)
    (torch.nn.functional.silu(y) if silu else y).backward(dy)
    xt = _nchw_to_tok(x.detach()).half().to(dev)
    kw = dict(silu=silu, frames_per_stat=fps)
    if c2:
        dx1, dx2 = K.groupnorm_bwd(xt[..., :c1].contiguous(), _nchw_to_tok(dy).half().to(dev), gamma.half().to(dev),
                                   beta.half().to(dev), groups, 1e-5, x2=xt[..., c1:].contiguous(), **kw)
        got = torch.cat([dx1, dx2], dim=3)
    else:
        got = K.groupnorm_bwd(xt, _nchw_to_tok(dy).half().to(dev), gamma.half().to(dev), beta.half().to(dev), groups, 1e-5, **kw)
    compare(got, _nchw_to_tok(x.grad), rel=GRAD_REL_TOL, name=f"GroupNorm backward C={c1}+{c2} fps={fps} silu={silu}")


@pytest.mark.parametrize("cin,cout,skip", [(64, 64, 0), (64, 128, 0), (96, 64, 32)])
def test_resnet_backward(dev, cin, cout, skip):
    from oracle.blocks import ResnetBlock2D as O
    from i2v_adapter_unofficial_amd.training import ResnetTrainer
    o, m = _pair(O, pkg().blocks.ResnetBlock2D, dev, 21, cin, cout, temb_channels=128, eps=1e-5, groups=16)
    g = torch.Generator().manual_seed(22)
    n, hw = 4, 16
    x = h(torch.randn(n, cin, hw, hw, generator=g)).requires_grad_()
    temb = h(torch.randn(2, 128, generator=g))
    dy = h(torch.randn(n, cout, hw, hw, generator=g))
    out = o(x, temb.repeat_interleave(2, dim=0))
    out.backward(dy)
    K = pkg().kernels
    tr = ResnetTrainer(m)
    p = m.packed()
    rows = K.gemm(K.silu(temb.half().to(dev)), p["wt"], p["bt"])
    xt = _nchw_to_tok(x.detach()).half().to(dev)
    x1, x2 = (xt[..., :cin - skip].contiguous(), xt[..., cin - skip:].contiguous()) if skip else (xt, None)
    y = tr.forward(x1, rows, x2=x2)
    compare(y, _nchw_to_tok(out), rel=3e-3, name="resnet training forward")
    dx, dx2 = tr.backward(_nchw_to_tok(dy).half().to(dev))
    got = torch.cat([dx, dx2], dim=3) if skip else dx
    compare(got, _nchw_to_tok(x.grad), rel=GRAD_REL_TOL, name=f"ResnetBlock2D backward {cin}->{cout} skip={skip}")


def test_samplers_backward(dev):
    from oracle.blocks import Downsample2D as OD, Upsample2D as OU
    from i2v_adapter_unofficial_amd.training import DownsampleTrainer, UpsampleTrainer
    g = torch.Generator().manual_seed(31)
    for ocls, hcls, tcls, scale in ((OD, pkg().blocks.Downsample2D, DownsampleTrainer, 0.5), (OU, pkg().blocks.Upsample2D, UpsampleTrainer, 2)):
        o, m = _pair(ocls, hcls, dev, 32, 64)
        x = h(torch.randn(3, 64, 16, 16, generator=g)).requires_grad_()
        out = o(x)
        dy = h(torch.randn(out.shape, generator=g))
        out.backward(dy)
        tr = tcls(m)
        y = tr.forward(_nchw_to_tok(x.detach()).half().to(dev))
        compare(y, _nchw_to_tok(out), rel=3e-3, name=f"{ocls.__name__} forward")
        compare(tr.backward(_nchw_to_tok(dy).half().to(dev)), _nchw_to_tok(x.grad), rel=GRAD_REL_TOL, name=f"{ocls.__name__} backward")


@pytest.mark.parametrize("frames", [4, 16])
def test_motion_module_backward(dev, frames):
    from oracle.blocks import TransformerTemporalModel as O
    from i2v_adapter_unofficial_amd.training import MotionModuleTrainer
    kw = dict(num_attention_heads=4, attention_head_dim=16, in_channels=64, norm_num_groups=16, attention_bias=False,
              activation_fn="geglu", positional_embeddings="sinusoidal", num_positional_embeddings=32)
    o, m = _pair(O, pkg().blocks.TransformerTemporalModel, dev, 41, **kw)
    g = torch.Generator().manual_seed(42)
    n, hw = 2 * frames, 8
    x = h(torch.randn(n, 64, hw, hw, generator=g)).requires_grad_()
    dy = h(torch.randn(n, 64, hw, hw, generator=g))
    out = o(x, num_frames=frames)[0]
    out.backward(dy)
    tr = MotionModuleTrainer(m)
    y = tr.forward(_nchw_to_tok(x.detach()).half().to(dev), frames)
    compare(y, _nchw_to_tok(out), rel=3e-3, name="motion module training forward")
    compare(tr.backward(_nchw_to_tok(dy).half().to(dev), frames), _nchw_to_tok(x.grad), rel=GRAD_REL_TOL,
            name=f"TransformerTemporalModel backward F={frames}")


@pytest.mark.parametrize("frames", [4, 16])
def test_motion_module_weight_gradients(dev, frames):
    """`--update_motion_modules` (train_image_to_video.py:452, 669; unet:984-999): the gradient of every one of the 26
    tensors of a motion module -- GroupNorm, proj_in, three LayerNorms, q / k / v / out of both temporal attentions, the
    GEGLU feed-forward, proj_out -- against torch autograd on the fp32 oracle, plus the column-sum-of-products kernel."""
    from oracle.blocks import TransformerTemporalModel as O
    from i2v_adapter_unofficial_amd.training import MotionModuleTrainer
    K = pkg().kernels
    kw = dict(num_attention_heads=4, attention_head_dim=16, in_channels=64, norm_num_groups=16, attention_bias=False,
              activation_fn="geglu", positional_embeddings="sinusoidal", num_positional_embeddings=32)
    o, m = _pair(O, pkg().blocks.TransformerTemporalModel, dev, 43, **kw)
    for prm in o.parameters():
        prm.requires_grad_(True)
    g = torch.Generator().manual_seed(44)
    n, hw = 2 * frames, 8
    x = h(torch.randn(n, 64, hw, hw, generator=g)).requires_grad_()
    dy = h(torch.randn(n, 64, hw, hw, generator=g))
    out = o(x, num_frames=frames)[0]
    out.backward(dy)
    tr = MotionModuleTrainer(m, train_weights=True)
    tr.forward(_nchw_to_tok(x.detach()).half().to(dev), frames)
    scale = 4.0
    dx, grads = tr.backward((_nchw_to_tok(dy) * scale).half().to(dev), frames, loss_scale=scale)
    compare(dx, _nchw_to_tok(x.grad) * scale, rel=GRAD_REL_TOL, name=f"motion module (training weights) d input F={frames}")
    ref = {k: v.grad for k, v in o.named_parameters()}
    assert set(grads) == set(ref) and len(ref) == 26, sorted(set(grads) ^ set(ref))
    for k in sorted(ref):
        assert grads[k].dtype == torch.float32 and grads[k].shape == ref[k].shape, k
        compare(grads[k], ref[k], rel=GRAD_REL_TOL, name=f"motion module F={frames}: d / d {k}")
    # the kernel under the gain gradients on a ragged shape (rows not a multiple of its 256-row blocks, 70 columns of 72)
    a = h(torch.randn(1000, 72, generator=g)).to(dev).half()
    b = h(torch.randn(1000, 72, generator=g)).to(dev).half()
    got = K.colsum_prod(a[:, :70], b[:, :70])
    want = (a[:, :70].double() * b[:, :70].double()).sum(0)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        K.colsum_prod(a, b[:, :70])


def test_transformer2d_backward(dev):
    from oracle.i2v_adapter import I2VAdapterTransformer2DModel as O
    from i2v_adapter_unofficial_amd.training import Transformer2DTrainer
    o, m = _pair(O, pkg().I2VAdapterTransformer2DModel, dev, 51, 4, 16, in_channels=64, num_layers=1, cross_attention_dim=48,
                 norm_num_groups=16)
    randomize_adapter_out_(o)
    m.load_state_dict(o.state_dict())
    m = m.to(device=dev, dtype=torch.float16)
    ad = o.transformer_blocks[0].i2v_adapter
    train = [ad.to_q.weight, ad.to_out[0].weight, ad.to_out[0].bias]
    for prm in train:
        prm.requires_grad_(True)
    g = torch.Generator().manual_seed(52)
    frames, n, hw = 4, 8, 8
    x = h(torch.randn(n, 64, hw, hw, generator=g)).requires_grad_()
    ctx = h(torch.randn(n, 7, 48, generator=g))
    dy = h(torch.randn(n, 64, hw, hw, generator=g))
    out = o(x, enable_cross_frame_attn=True, encoder_hidden_states=ctx, num_frames=frames, return_dict=False)[0]
    out.backward(dy)
    tr = Transformer2DTrainer(m)
    y = tr.forward(_nchw_to_tok(x.detach()).half().to(dev), frames, ctx.half().to(dev))
    compare(y, _nchw_to_tok(out), rel=3e-3, name="Transformer2D training forward")
    dx, grads = tr.backward(_nchw_to_tok(dy).half().to(dev), loss_scale=1.0)
    compare(dx, _nchw_to_tok(x.grad), rel=GRAD_REL_TOL, name="Transformer2D backward: d / d input")
    for key, ref in zip(("i2v_adapter.to_q.weight", "i2v_adapter.to_out.0.weight", "i2v_adapter.to_out.0.bias"), train):
        compare(grads[key], ref.grad, rel=GRAD_REL_TOL, name=f"Transformer2D backward: {key}")


# ---------------------------------------------------------------------------------------------- the whole training step
@pytest.mark.parametrize("ip,motion", [(False, False), (True, False), (False, True)])
def test_unet_training_step_vs_autograd(dev, ip, motion):
    """The reference's step (train_image_to_video.py:839-884) on the reduced UNet (SD-1.5 topology, narrow channels): forward
    with enable_cross_frame_attn=True, loss = MSE without the first frame, backward through every layer -- the gradient of
    ALL 16 x 3 trainable adapter tensors (unet:979-1026) and the loss against torch autograd on the fp32 oracle UNet."""
    from tests.parity import hip_unet_from_oracle, host_threads, oracle_small_unet, small_unet_inputs
    from i2v_adapter_unofficial_amd.training import UNetAdapterTrainer
    host_threads()
    from tests.parity import small_ip_state_dict
    ou = oracle_small_unet(seed=77, ip=ip)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou) if ip else None)
    for prm in ou.parameters():
        prm.requires_grad_(False)
    ou.freeze_unet_params() if hasattr(ou, "freeze_unet_params") else None
    # motion=True: `--update_motion_modules` -> freeze_unet_params(freeze_animatediff=False) (unet:984-999): + 21 x 26 tensors
    train = {n: prm for n, prm in ou.named_parameters() if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n
             or (motion and ".motion_modules." in n)}
    for prm in train.values():
        prm.requires_grad_(True)
    assert len(train) == 16 * 3 + (21 * 26 if motion else 0)
    hu.freeze_unet_params(freeze_animatediff=not motion)
    assert {n for n, prm in hu.named_parameters() if prm.requires_grad} == set(train)
    inp = small_unet_inputs(b=2, f=4, hw=16)
    t = torch.tensor([481, 481])
    g = torch.Generator().manual_seed(78)
    target = h(torch.randn(inp["sample"].shape, generator=g))
    added = {"image_embeds": inp["image_embeds"]} if ip else None
    added_d = {"image_embeds": inp["image_embeds"].half().to(dev)} if ip else None
    pred = ou(inp["sample"], t, True, inp["ctx"], added_cond_kwargs=added).sample
    mask = torch.ones_like(pred)
    mask[:, 0] = 0
    loss = ((pred.float() - target) ** 2 * mask).sum() / mask.sum()
    loss.backward()

    tr = UNetAdapterTrainer(hu, update_motion_modules=motion)
    y = tr.forward(inp["sample"].half().to(dev), t.to(dev), inp["ctx"].half().to(dev), added_cond_kwargs=added_d)
    got_pred = y[..., :4].float().cpu().permute(0, 3, 1, 2).reshape(pred.shape)
    compare(got_pred, pred, rel=6e-3, name="training forward of the reduced UNet")
    got_loss, grads = tr.backward(target.to(dev), loss_scale=2.0 ** 12)
    assert abs(got_loss.item() - loss.item()) <= 5e-4 * abs(loss.item()), (got_loss.item(), loss.item())   # measured <= 7e-6
    assert set(grads) == set(train)
    worst = 0.0
    for name, prm in train.items():
        err, scale = compare(grads[name], prm.grad, rel=1.2e-2, name=f"UNet step: d loss / d {name}")
        worst = max(worst, err / scale)
    print(f"UNet training step: loss {got_loss.item():.6f} vs {loss.item():.6f}, worst gradient error {worst:.2e} of max")


def test_update_motion_modules_optimizer_step(dev):
    """`--update_motion_modules`: the optimiser's buckets cover the adapter AND the motion modules (unet:984-1006), one step
    moves every such parameter by at most lr (1 + weight decay |p|) (AdamW's first step is lr sign(g)), leaves every frozen
    parameter bit-identical, and the next forward runs on the new weights (the packed / transposed operand copies follow
    the parameters' version counters)."""
    from tests.parity import hip_unet_from_oracle, oracle_small_unet, small_unet_inputs
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer, UNetAdapterTrainer
    hu = hip_unet_from_oracle(oracle_small_unet(seed=79), dev)
    hu.freeze_unet_params(freeze_animatediff=False)
    lr = 1e-3
    opt = AdapterOptimizer(hu, lr=lr, max_grad_norm=1.0, update_motion_modules=True)
    assert set(opt.names) == {n for n, prm in hu.named_parameters() if prm.requires_grad}
    assert len(opt.names) == 16 * 3 + 21 * 26
    before = {n: prm.detach().clone() for n, prm in hu.named_parameters()}
    inp = small_unet_inputs(b=2, f=4, hw=16)
    t = torch.tensor([300, 300]).to(dev)
    target = torch.randn(inp["sample"].shape, generator=torch.Generator().manual_seed(80)).to(dev)
    tr = UNetAdapterTrainer(hu, update_motion_modules=True)
    y0 = tr.forward(inp["sample"].half().to(dev), t, inp["ctx"].half().to(dev)).clone()
    _, grads = tr.backward(target, loss_scale=2.0 ** 10)
    assert opt.step(grads) and not opt.last_step_skipped()
    moved = 0
    for n, prm in hu.named_parameters():
        d = (prm.detach().float() - before[n].float()).abs()
        if n in opt.offsets:
            bound = lr * (1.0 + 1e-2 * before[n].float().abs()) + 9.8e-4 * before[n].float().abs()    # + one fp16 rounding
            assert (d <= bound * 1.001 + 1e-7).all(), n
            moved += int(d.max().item() > 0)
        else:
            assert d.max().item() == 0.0, f"frozen parameter {n} moved"
    assert moved >= 0.9 * len(opt.names), moved
    y1 = tr.forward(inp["sample"].half().to(dev), t, inp["ctx"].half().to(dev))
    assert (y1.float() - y0.float()).abs().max().item() > 0
    tr.backward(target, loss_scale=2.0 ** 10)
    # the fp32 masters moved DOWN the gradient: sum g . (p_new - p_old) < 0 over the adapter and over the motion modules
    for part in (".i2v_adapter.", ".motion_modules."):
        dot = sum((grads[n].double().reshape(-1) * (opt.master[o: o + c].double() - before[n].double().reshape(-1))).sum().item()
                  for n, (o, c) in opt.offsets.items() if part in n)
        assert dot < 0, (part, dot)


def test_adamw_clip_matches_torch(dev):
    """AdapterOptimizer (flat fp32 buckets, clip + AdamW in two kernels) against torch.optim.AdamW + clip_grad_norm_ on
    the same parameters / gradients for three steps (train_image_to_video.py:716-724, 876-882 defaults)."""
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer
    p = pkg()
    m = p.I2VAdapterTransformerBlock(64, 4, 16, cross_attention_dim=32)

    class Holder(torch.nn.Module):
        def __init__(self, blk):
            super().__init__()
            self.encoder_hid_proj = None
            self.blocks = torch.nn.ModuleList([blk])
    hold = Holder(m).to(dev).half()
    ref_params = [torch.nn.Parameter(prm.detach().float().cpu().clone()) for n, prm in hold.named_parameters()
                  if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n]
    ref_opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    opt = AdapterOptimizer(hold, lr=1e-3, max_grad_norm=1.0)
    assert len(opt.names) == 3 == len(ref_params)
    g = torch.Generator().manual_seed(1)
    for step in range(3):
        grads = {n: torch.randn(prm.shape, generator=g) * (3.0 if step == 0 else 0.01) for n, prm in zip(opt.names, ref_params)}
        for prm, n in zip(ref_params, opt.names):
            prm.grad = grads[n].clone()
        torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
        ref_opt.step()
        opt.step({n: v.to(dev) for n, v in grads.items()})
    for n, prm in zip(opt.names, ref_params):
        off, cnt = opt.offsets[n]
        compare(opt.master[off: off + cnt].view_as(prm), prm, rel=1e-5, name=f"AdamW master {n}")
    # the fp16 parameters are the rounded masters: equal to the reference's rounding up to one fp16 ulp where a master sits on
    # a rounding boundary (the bias correction 1 - beta^step is evaluated on the device since the guarded step counts there)
    got = dict(hold.named_parameters())
    for n, prm in zip(opt.names, ref_params):
        off, cnt = opt.offsets[n]
        assert torch.equal(got[n].detach().cpu(), opt.master[off: off + cnt].view_as(prm).half().cpu())
        d = (got[n].detach().float().cpu() - prm.detach().half().float()).abs()
        assert (d <= prm.detach().abs() * 2.0 ** -10 + 1e-7).all()


def test_adamw_step_with_overflowed_gradients_is_skipped(dev):
    """one inf in the fp16-scaled gradients must not poison the weights.  The guarded step leaves parameters,
    both moments and the bias-correction step untouched and raises the device flag (what accelerate's GradScaler does for the
    reference's `--mixed_precision fp16` run, train_image_to_video.py:306-308); the next finite step then equals the step a
    torch optimiser takes that never saw the bad one.  The gradient norm is summed in a fixed order: bit-identical run to run."""
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer
    p = pkg()
    m = p.I2VAdapterTransformerBlock(64, 4, 16, cross_attention_dim=32)

    class Holder(torch.nn.Module):
        def __init__(self, blk):
            super().__init__()
            self.encoder_hid_proj = None
            self.blocks = torch.nn.ModuleList([blk])
    hold = Holder(m).to(dev).half()
    ref_params = [torch.nn.Parameter(prm.detach().float().cpu().clone()) for n, prm in hold.named_parameters()
                  if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n]
    ref_opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    opt = AdapterOptimizer(hold, lr=1e-3, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(5)
    draw = lambda s: {n: torch.randn(prm.shape, generator=g) * s for n, prm in zip(opt.names, ref_params)}

    def ref_step(grads):
        for prm, n in zip(ref_params, opt.names):
            prm.grad = grads[n].clone()
        torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
        ref_opt.step()

    g1 = draw(0.5)
    ref_step(g1)
    opt.step({n: v.to(dev) for n, v in g1.items()})
    assert not opt.last_step_skipped() and opt.applied_steps.item() == 1
    before = [t.clone() for t in (opt.master, opt.exp_avg, opt.exp_avg_sq)]
    params_before = {n: prm.detach().clone() for n, prm in hold.named_parameters()}
    for bad_value in (float("inf"), float("nan")):
        bad = {n: v.clone() for n, v in draw(0.5).items()}
        bad[opt.names[1]].view(-1)[7] = bad_value
        opt.step({n: v.to(dev) for n, v in bad.items()})
        assert opt.last_step_skipped() and opt.applied_steps.item() == 1 and opt.step_count >= 2
        for t, b in zip((opt.master, opt.exp_avg, opt.exp_avg_sq), before):
            assert torch.equal(t, b), "a skipped step must not touch masters or moments"
        for n, prm in hold.named_parameters():
            assert torch.equal(prm.detach(), params_before[n])
    g2 = draw(0.01)
    ref_step(g2)
    opt.step({n: v.to(dev) for n, v in g2.items()})
    assert not opt.last_step_skipped() and opt.applied_steps.item() == 2
    for n, prm in zip(opt.names, ref_params):
        off, cnt = opt.offsets[n]
        compare(opt.master[off: off + cnt].view_as(prm), prm, rel=1e-5, name=f"AdamW master after a skipped step: {n}")
    # fixed-order norm: the same bucket gives the same bits every time
    norms = []
    for _ in range(3):
        opt.fill_gradients({n: v.to(dev) for n, v in g1.items()})
        pkg().kernels.adamw_guarded_step(opt.master.clone(), opt.grad, opt.exp_avg.clone(), opt.exp_avg_sq.clone(), lr=0.0,
                               betas=opt.betas, eps=opt.eps, weight_decay=0.0, grad_coef=1.0, max_norm=1.0,
                               partials=opt._partials, norm_sq=opt.norm_sq, applied_steps=opt.applied_steps.clone(),
                               found_inf=opt.found_inf.clone())
        norms.append(opt.norm_sq.clone())
    assert torch.equal(norms[0], norms[1]) and torch.equal(norms[1], norms[2])
    want = sum((v.double() ** 2).sum() for v in g1.values()).item()
    assert abs(norms[0].item() - want) <= 1e-5 * want


def test_training_steps_do_not_grow_memory(dev):
    """forward + backward + optimiser step, repeated -- the in-place parameter writes of the optimiser must
    neither re-pack frozen weights nor leave operand copies behind: the memo's size and the allocated bytes are constant from
    the third cycle on."""
    from tests.parity import hip_unet_from_oracle, oracle_small_unet, small_unet_inputs
    from i2v_adapter_unofficial_amd import training
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer, UNetAdapterTrainer
    hu = hip_unet_from_oracle(oracle_small_unet(seed=79), dev)
    inp = small_unet_inputs(b=1, f=4, hw=16)
    t = torch.tensor([481])
    target = h(torch.randn(inp["sample"].shape, generator=torch.Generator().manual_seed(80)))
    sample, ctx, tgt = inp["sample"].half().to(dev), inp["ctx"][:1].half().to(dev), target.to(dev)
    tr, opt = UNetAdapterTrainer(hu), AdapterOptimizer(hu, lr=1e-4)
    t2d = hu.down_blocks[0].attentions[0]
    packs, sizes, mem, losses = [], [], [], []
    for step in range(6):
        tr.forward(sample, t.to(dev), ctx)
        loss, grads = tr.backward(tgt, loss_scale=2.0 ** 10)
        opt.step(grads)
        del grads
        torch.cuda.synchronize()
        packs.append(t2d.packed()["wo"].data_ptr())
        sizes.append(len(training._memo.d))
        mem.append(torch.cuda.memory_allocated())
        losses.append(loss.item())
    assert not opt.last_step_skipped()
    assert len(set(packs)) == 1, "Transformer2D re-packed proj_in / proj_out although only the adapter's weights moved"
    assert len(set(sizes[2:])) == 1, f"operand memo keeps growing: {sizes}"
    assert len(set(mem[2:])) == 1, f"allocated bytes keep growing: {mem}"
    assert all(l == l for l in losses) and losses[-1] != losses[0], "the steps must actually move the weights"


def test_gradient_accumulation_and_ema(dev):
    """`accelerator.accumulate` (train_image_to_video.py:486, 785) and `--use_ema` (:673-677, 888-889) on the flat buckets: N
    micro-batches accumulate into one update that equals torch's step on the MEAN gradient; the EMA follows diffusers
    EMAModel's schedule decay_t = min(decay, (1 + t) / (10 + t)), t = updates - 1 (0 on the first update)."""
    from i2v_adapter_unofficial_amd.training import AdapterOptimizer
    p = pkg()
    m = p.I2VAdapterTransformerBlock(64, 4, 16, cross_attention_dim=32)

    class Holder(torch.nn.Module):
        def __init__(self, blk):
            super().__init__()
            self.encoder_hid_proj = None
            self.blocks = torch.nn.ModuleList([blk])
    hold = Holder(m).to(dev).half()
    ref_params = [torch.nn.Parameter(prm.detach().float().cpu().clone()) for n, prm in hold.named_parameters()
                  if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n]
    ref_opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    opt = AdapterOptimizer(hold, lr=1e-3, max_grad_norm=1.0, gradient_accumulation_steps=3, use_ema=True, ema_decay=0.95)
    ema_ref = [prm.detach().clone() for prm in ref_params]
    g = torch.Generator().manual_seed(9)
    updates = 0
    for it in range(12):
        grads = {n: torch.randn(prm.shape, generator=g) * 0.05 for n, prm in zip(opt.names, ref_params)}
        for prm, n in zip(ref_params, opt.names):
            prm.grad = grads[n] / 3 if prm.grad is None else prm.grad + grads[n] / 3
        stepped = opt.step({n: v.to(dev) for n, v in grads.items()})
        assert stepped == (it % 3 == 2)
        if stepped:
            torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
            ref_opt.step()
            ref_opt.zero_grad(set_to_none=True)
            updates += 1
            t = max(0, updates - 1)
            d = 0.0 if t <= 0 else min(0.95, (1.0 + t) / (10.0 + t))
            for e, prm in zip(ema_ref, ref_params):
                e.mul_(d).add_(prm.detach(), alpha=1.0 - d)
    assert updates == 4 and opt.applied_steps.item() == 4
    ema = opt.ema_state_dict()
    for n, prm, e in zip(opt.names, ref_params, ema_ref):
        off, cnt = opt.offsets[n]
        compare(opt.master[off: off + cnt].view_as(prm), prm, rel=1e-5, name=f"accumulated AdamW master {n}")
        compare(ema[n], e, rel=1e-5, name=f"EMA {n}")
