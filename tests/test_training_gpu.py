"""SURVEY 8 f4, first vertical slice: the backward of one I2VAdapterTransformerBlock on the HIP kernels against torch
autograd over the fp32 CPU oracle block -- the gradients the reference's optimiser receives for the block's trainable
parameters (unet:979-1026: i2v_adapter.to_q / to_out) and the gradient it passes on to the earlier layers, under the
reference's loss (train_image_to_video.py:848-856: MSE without the first frame of each clip).

Kernel-level cases first (attention backward incl. the kv_group reduction of dK0 / dV0, LayerNorm / GEGLU backward, the
transposes and sums), then the block.  Tolerance: 5e-3 of the largest reference gradient entry (fp16 operands, fp32
accumulation; the P / dS operands of the attention backward are fp16 like the forward's P)."""
import pytest
import torch

from tests.parity import compare, randomize_adapter_out_, round_fp16_

pytestmark = pytest.mark.gpu
GRAD_REL_TOL = 5.0e-3


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def h(t):
    return t.half().float()


@pytest.mark.parametrize("d,lq,lk,group,need_dkv", [(40, 256, 256, 1, True), (40, 128, 128, 4, True), (80, 64, 64, 2, True),
                                                    (160, 64, 64, 1, True), (40, 96, 77, 2, False), (64, 32, 160, 1, True)])
def test_attention_backward_vs_autograd(dev, d, lq, lk, group, need_dkv):
    """dQ, dK, dV of softmax(q k^T / sqrt d) v against autograd; kv_group > 1 = several query batches share one K / V (the
    cross-frame attention: dK / dV are the sums over the group); lk = 77 is the text context (dQ only, masked tail)."""
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
    if need_dkv:
        compare(dk, k.grad.reshape(-1, C), rel=GRAD_REL_TOL, name=f"dK d={d} group={group}")
        compare(dv, v.grad.reshape(-1, C), rel=GRAD_REL_TOL, name=f"dV d={d} group={group}")
        dq2, dk2, dv2 = K.attention_bwd(qd, kd, vd, od, to(do), batch_q=bq, lq=lq, lk=lk, heads=heads, head_dim=d,
                                        kv_group=group)
        assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)      # no atomics: run-to-run identical
    else:
        assert dk is None and dv is None


def test_attention_lse(dev):
    K = pkg().kernels
    g = torch.Generator().manual_seed(3)
    bq, heads, d, lq, lk = 3, 4, 40, 70, 77
    q, k = h(torch.randn(bq, lq, heads * d, generator=g)), h(torch.randn(bq, lk, heads * d, generator=g))
    s = torch.einsum("blhd,bmhd->bhlm", q.view(bq, lq, heads, d), k.view(bq, lk, heads, d)) * d ** -0.5
    ref = torch.logsumexp(s, dim=-1) * 1.4426950408889634
    got = K.attention_lse(q.half().to(dev).view(-1, heads * d), k.half().to(dev).view(-1, heads * d), batch_q=bq, lq=lq, lk=lk,
                          heads=heads, head_dim=d)
    compare(got, ref, rel=1e-3, name="attention log2-sum-exp")


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
    y, tg = h(torch.randn(8, 16, 32, generator=g)), h(torch.randn(8, 16, 32, generator=g))
    gr = K.masked_mse_grad(y.half().to(dev), tg.half().to(dev), frames=4, coef=0.25).float().cpu()
    ref = 0.25 * (y - tg)
    ref[0::4] = 0
    compare(gr, ref, rel=1e-3, name="masked MSE seed gradient")
    assert (gr[0] == 0).all() and (gr[4] == 0).all()


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
    assert all(torch.equal(grads[k], again[k]) for k in grads if k != "i2v_adapter.to_out.0.bias")   # (bias: fp32 atomics)
