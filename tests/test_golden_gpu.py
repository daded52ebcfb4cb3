"""GPU parity against the committed golden vectors: (1) the reference-authored BasicAttention outputs
(K1 cross-frame / K2 self / K3 text attention shapes) through the HIP Attention path, (2) the oracle outputs
stored in tests/golden/oracle_outputs.safetensors (no oracle execution on the GPU box needed)."""
import glob
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

from tests.parity import (REL_TOL_UNET, compare, hip_unet_from_oracle, oracle_small_unet, round_fp16_,
                          small_unet_inputs)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ref_attention_*.safetensors"))))
def test_hip_attention_vs_reference_golden(dev, path):
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd import kernels as K
    t = load_file(path)
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
    heads, d, frames, clips = int(meta["heads"]), int(meta["head_dim"]), int(meta["frames"]), int(meta["clips"])
    c = heads * d
    a = pkg.Attention(c, cross_attention_dim=t["to_k"].shape[1], heads=heads, dim_head=d)
    a.load_state_dict({"to_q.weight": t["to_q"], "to_k.weight": t["to_k"], "to_v.weight": t["to_v"],
                       "to_out.0.weight": t["to_out_w"], "to_out.0.bias": t["to_out_b"]})
    a = a.to(dev).half()
    x = t["x"].half().to(dev)
    if meta["kind"] == "self":
        y = a(x)
    elif meta["kind"] == "text":
        y = a(x, encoder_hidden_states=t["ctx"].half().to(dev))
    else:
        # K1 the MI355X way: frame-0 tokens gathered once per clip, K0 / V0^T projected once, kv_group = frames
        p = a.packed()
        n, L = x.shape[0], x.shape[1]
        first = torch.empty((clips, L, c), dtype=torch.float16, device=dev)
        K.copy3d(x.view(clips, frames * L, c)[:, :L], first)
        q = K.gemm(x.view(-1, c), p["wq"])
        k0 = K.gemm(first.view(-1, c), p["wk"])
        v0t = K.project_vt(first.view(-1, c), p["wv"], L)
        o = K.attention(q, k0, v0t, batch_q=n, lq=L, lk=L, heads=heads, head_dim=d, kv_group=frames)
        y = K.gemm(o, p["wo"], p["bo"]).view(n, L, c)
    compare(y, t["y"], rel=1.5e-3, name=os.path.basename(path))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ref_transformer_block_*.safetensors"))))
def test_hip_block_vs_reference_basic_transformer_block(dev, path):
    """the HIP I2VAdapterTransformerBlock (adapter off, feed-forward zeroed) reproduces the reference-authored
    BasicTransformerBlock (src/modules/attention.py:64-77) on its own weights: LN -> attn -> +x, LN -> cross-attn -> +x."""
    import i2v_adapter_unofficial_amd as pkg
    from tests.test_oracle import load_ref_block_into
    t = load_file(path)
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
    heads, d = int(meta["heads"]), int(meta["head_dim"])
    torch.manual_seed(0)
    blk = load_ref_block_into(pkg.I2VAdapterTransformerBlock(heads * d, heads, d, cross_attention_dim=t["ctx"].shape[-1]), t)
    blk = blk.to(dev).half().eval()
    y = blk(t["x"].half().to(dev), enable_cross_frame_attn=False, encoder_hidden_states=t["ctx"].half().to(dev))
    compare(y, t["y"], rel=1.5e-3, name=os.path.basename(path))


def test_hip_timestep_embedding_vs_reference_positional_emb(dev):
    from i2v_adapter_unofficial_amd import kernels as K
    t = load_file(os.path.join(GOLD, "ref_positional_emb.safetensors"))
    for c in (320, 32):
        e = K.timestep_embedding(t["t"][:, 0].contiguous().to(dev), c).float().cpu()
        ref = torch.cat([t[f"emb{c}"][:, c // 2:], t[f"emb{c}"][:, : c // 2]], dim=1)      # [sin | cos] -> [cos | sin]
        assert (e - ref).abs().max().item() <= 1.5e-3, (c, (e - ref).abs().max())           # fp16 output of |v| <= 1


def test_hip_conv_groupnorm_vs_reference_resblock(dev):
    """bias-free conv3x3 -> GroupNorm(8) and the 1x1 res_conv of the reference-authored ResBlock
    (src/modules/resnet.py:19-72) through the HIP conv / GroupNorm / GEMM kernels."""
    from i2v_adapter_unofficial_amd import kernels as K
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
    t = load_file(os.path.join(GOLD, "ref_resblock_conv_gn.safetensors"))
    x = K.nchw_to_tokens(t["x"].to(dev))
    y = K.conv3x3(x, pack_conv3x3(t["conv_w"]).to(dev))
    compare(K.tokens_to_nchw(y, dtype=torch.float32), t["y_conv"], rel=1e-3, name="ResBlock conv1[0]")
    z = K.groupnorm(y, t["gn_w"].half().to(dev), t["gn_b"].half().to(dev), 8, 1e-5)
    compare(K.tokens_to_nchw(z, dtype=torch.float32), t["y_conv_gn"], rel=1.5e-3, name="ResBlock conv1[0:2]")
    co, ci = t["res_w"].shape[:2]
    r = K.gemm(x.view(-1, ci), t["res_w"].reshape(co, ci).half().to(dev), t["res_b"].half().to(dev))
    compare(K.tokens_to_nchw(r.view(x.shape[0], x.shape[1], x.shape[2], co), dtype=torch.float32), t["y_res"], rel=1.2e-3,
            name="ResBlock res_conv")


def test_hip_motion_module_vs_reference_video_transformer(dev):
    """the HIP TransformerTemporalModel -- entry GroupNorm writing rows in (b, pixel, frame) order, LayerNorm-folded
    q|k / V^T projections, the temporal-attention kernel, out-projections, the ROWPERM store back to (b, frame, pixel)
    -- against the reference-authored VideoTransformer temporal path (src/modules/attention.py:79-131) on its own
    weights (set-up: tests/test_oracle.py temporal_model_from_video_transformer_fixture)."""
    import i2v_adapter_unofficial_amd as pkg
    from tests.test_oracle import temporal_model_from_video_transformer_fixture
    path = os.path.join(GOLD, "ref_video_transformer_temporal.safetensors")
    t = load_file(path)
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
    m, frames = temporal_model_from_video_transformer_fixture(pkg.TransformerTemporalModel, t, meta)
    m = m.to(dev).half().eval()
    with torch.no_grad():
        y = m(t["x"].to(dev), num_frames=frames)[0].float().cpu() - t["x"]
    # the module returns x + block(x) in fp16 (|.| up to ~8): its rounding alone is 2e-3 absolute
    compare(y, t["y"], rel=2.5e-3, name="HIP motion module vs reference VideoTransformer temporal path")


def test_hip_kernels_compose_reference_resblock_forward(dev):
    """ResBlock.forward of the reference (src/modules/resnet.py:63-72), whole, composed from the HIP kernels: conv3x3 ->
    GroupNorm -> GELU (GEMM epilogue), + emb_layer(t) as a per-image row vector (Linear -> SiLU -> Linear), conv3x3 ->
    GroupNorm -> GELU, + 1x1 res_conv(x) through the fused residual: the temb-add / shortcut composition of the hot
    path's ResnetBlock2D on reference-authored numbers."""
    from i2v_adapter_unofficial_amd import kernels as K
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
    t = load_file(os.path.join(GOLD, "ref_resblock_forward.safetensors"))
    d16 = lambda v: v.half().to(dev)
    x = K.nchw_to_tokens(t["x"].to(dev))                                   # [3, 12, 12, 64]
    n, hh, ww, ci = x.shape
    cm, co = t["conv1.0.weight"].shape[0], t["conv2.0.weight"].shape[0]
    eye = lambda c: torch.eye(c, dtype=torch.float16, device=dev)

    def gn_gelu(v, w, b):      # GroupNorm(8) kernel, then GELU as the epilogue of an identity GEMM
        z = K.groupnorm(v, d16(w), d16(b), 8, 1e-5)
        c = z.shape[-1]
        return K.gemm(z.view(-1, c), eye(c), epilogue=K.I2V_EPI_GELU)

    h1 = gn_gelu(K.conv3x3(x, pack_conv3x3(t["conv1.0.weight"]).to(dev)), t["conv1.1.weight"], t["conv1.1.bias"])
    e = K.gemm(d16(t["timesteps"]), d16(t["emb_layer.0.weight"]), d16(t["emb_layer.0.bias"]))
    e = K.gemm(K.silu(e), d16(t["emb_layer.2.weight"]), d16(t["emb_layer.2.bias"]))          # [3, cm]
    h1 = K.gemm(h1, eye(cm), rowvec=e, rows_per_vec=hh * ww).view(n, hh, ww, cm)             # + emb[..., None, None]
    h2 = gn_gelu(K.conv3x3(h1, pack_conv3x3(t["conv2.0.weight"]).to(dev)), t["conv2.1.weight"], t["conv2.1.bias"])
    out = K.gemm(x.view(-1, ci), d16(t["res_conv.weight"].reshape(co, ci)), d16(t["res_conv.bias"]), residual=h2)
    compare(K.tokens_to_nchw(out.view(n, hh, ww, co), dtype=torch.float32), t["y"], rel=2.5e-3,
            name="HIP kernels composing the reference ResBlock.forward")


def test_hip_unet_vs_committed_oracle_outputs(dev):
    gold = load_file(os.path.join(GOLD, "oracle_outputs.safetensors"))
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    inp = small_unet_inputs()
    with torch.no_grad():
        y = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
        y2 = hu(inp["sample"].to(dev), inp["timestep"].to(dev), False, inp["ctx"].to(dev)).sample
    compare(y, gold["unet_y"], rel=REL_TOL_UNET, name="unet_y")
    compare(y2, gold["unet_y_no_cross_frame"], rel=REL_TOL_UNET, name="unet_y_no_cross_frame")


def test_hip_block_vs_committed_oracle_outputs(dev):
    import i2v_adapter_unofficial_amd as pkg
    from oracle.i2v_adapter import I2VAdapterTransformerBlock as O
    gold = load_file(os.path.join(GOLD, "oracle_outputs.safetensors"))
    torch.manual_seed(101)
    o = round_fp16_(O(256, 8, 32, dropout=0.0, cross_attention_dim=512, activation_fn="gelu"))
    m = pkg.I2VAdapterTransformerBlock(256, 8, 32, dropout=0.0, cross_attention_dim=512, activation_fn="gelu")
    m.load_state_dict(o.state_dict())
    m = m.to(dev).half()
    g = torch.Generator().manual_seed(102)
    x = torch.randn(8, 64, 256, generator=g).half()
    ctx = torch.randn(8, 77, 512, generator=g).half()
    y = m(x.to(dev), enable_cross_frame_attn=True, num_frames=4, encoder_hidden_states=ctx.to(dev))
    compare(y, gold["block_y_cross_frame"], name="block_y_cross_frame")
    y = m(x.to(dev), enable_cross_frame_attn=False, encoder_hidden_states=ctx.to(dev))
    compare(y, gold["block_y_plain"], name="block_y_plain")


@pytest.mark.parametrize("name", ["d40", "d8"])
def test_hip_block_backward_vs_reference_autograd(dev, name):
    """The training step's block backward (training.AdapterBlockTrainer: LayerNorm backward, MFMA attention backward, Linear
    dgrad / wgrad, bias sums) against torch autograd through the REFERENCE-AUTHORED BasicTransformerBlock
    (src/modules/attention.py:64-77; tests/golden/ref_grads_transformer_block_*: weights and inputs of the forward fixture,
    an output gradient dy, d / d x and the gradients of attn1.to_out).  The HIP block reproduces that block with its
    feed-forward off and its adapter = a copy of attn1's q / k / v with a ZERO to_out and one frame per clip (the cross-frame
    attention is then attn1 itself and contributes nothing forward): its input gradient must equal the reference's, and the
    gradients it reports for i2v_adapter.to_out must equal the reference's for attn1.to_out (same dY, same attention output)."""
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd.training import AdapterBlockTrainer
    from tests.test_oracle import load_ref_block_into
    fwd_path = os.path.join(GOLD, f"ref_transformer_block_{name}.safetensors")
    t, gr = load_file(fwd_path), load_file(os.path.join(GOLD, f"ref_grads_transformer_block_{name}.safetensors"))
    with safe_open(fwd_path, framework="pt") as f:
        meta = f.metadata()
    heads, d = int(meta["heads"]), int(meta["head_dim"])
    c = heads * d
    torch.manual_seed(0)
    blk = load_ref_block_into(pkg.I2VAdapterTransformerBlock(c, heads, d, cross_attention_dim=t["ctx"].shape[-1]), t)
    with torch.no_grad():
        blk.i2v_adapter.to_q.weight.copy_(blk.attn1.to_q.weight)
        blk.i2v_adapter.to_k.weight.copy_(blk.attn1.to_k.weight)
        blk.i2v_adapter.to_v.weight.copy_(blk.attn1.to_v.weight)
        blk.i2v_adapter.to_out[0].weight.zero_()
        blk.i2v_adapter.to_out[0].bias.zero_()
    blk = blk.to(dev).half().eval()
    n_img, L, _ = t["x"].shape
    tr = AdapterBlockTrainer(blk)
    y = tr.forward(t["x"].half().to(dev).view(-1, c), n_img, L, 1, t["ctx"].half().to(dev))
    compare(y.view(n_img, L, c), t["y"], rel=1.5e-3, name=f"training forward vs reference block {name}")
    grads = tr.backward(gr["dy"].half().to(dev).view(-1, c), loss_scale=1.0)
    compare(grads["hidden_states"].view(n_img, L, c), gr["dx"], rel=1.8e-3, name=f"d / d x vs reference autograd {name}")
    compare(grads["i2v_adapter.to_out.0.weight"], gr["d_attn1_to_out_weight"], rel=1.5e-3, name=f"to_out weight gradient vs reference {name}")
    compare(grads["i2v_adapter.to_out.0.bias"], gr["d_attn1_to_out_bias"], rel=1e-3, name=f"to_out bias gradient vs reference {name}")
