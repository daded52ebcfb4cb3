"""CPU tests of the oracle (test infrastructure): pinned against the reference-authored BasicAttention golden
vectors, torch functional ops, the reference's shape contracts and its add_noise known-answer test, and against
its own committed outputs (regression)."""
import glob
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors import safe_open
from safetensors.torch import load_file

from tests.parity import oracle_small_unet, round_fp16_, small_unet_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load_ref(path):
    with safe_open(path, framework="pt") as f:
        meta = f.metadata()
    return load_file(path), meta


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ref_attention_*.safetensors"))))
def test_attention_matches_reference_basic_attention(path):
    """oracle Attention == reference src/modules/attention.py:26-62 on K1 / K2 / K3 shaped inputs."""
    from oracle.blocks import Attention
    t, meta = _load_ref(path)
    heads, d, frames = int(meta["heads"]), int(meta["head_dim"]), int(meta["frames"])
    c = heads * d
    ctx_dim = t["to_k"].shape[1]
    a = Attention(c, cross_attention_dim=ctx_dim, heads=heads, dim_head=d)
    a.load_state_dict({"to_q.weight": t["to_q"], "to_k.weight": t["to_k"], "to_v.weight": t["to_v"],
                       "to_out.0.weight": t["to_out_w"], "to_out.0.bias": t["to_out_b"]})
    x = t["x"]
    if meta["kind"] == "self":
        ctx = None
    elif meta["kind"] == "cross_frame":
        ctx = x[0:x.shape[0]:frames].repeat_interleave(frames, dim=0)
    else:
        ctx = t["ctx"]
    with torch.no_grad():
        y = a(x, encoder_hidden_states=ctx)
    assert torch.allclose(y, t["y"], atol=2e-6, rtol=1e-5), (y - t["y"]).abs().max()


def test_golden_fixture_count():
    assert len(glob.glob(os.path.join(GOLD, "ref_attention_*.safetensors"))) == 6
    assert len(glob.glob(os.path.join(GOLD, "ref_transformer_block_*.safetensors"))) == 2
    for f in ("oracle_outputs", "ref_positional_emb", "ref_resblock_conv_gn", "ref_video_transformer_temporal",
              "ref_resblock_forward"):
        assert os.path.exists(os.path.join(GOLD, f + ".safetensors"))


def load_ref_block_into(block, t):
    """weights of the reference-authored BasicTransformerBlock (src/modules/attention.py:64-77: attn1, attn2, norm1,
    norm2; `to_out` is a Sequential there, index 0 = the Linear as in diffusers) into a block of this repo, and the
    GEGLU feed-forward switched off (ff.net.2 = 0 => ff(norm3(x)) + x == x)."""
    sd = {k: v for k, v in t.items() if k.split(".")[0] in ("attn1", "attn2", "norm1", "norm2")}
    missing, unexpected = block.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.split(".")[0] in ("norm3", "ff", "i2v_adapter", "pos_embed") for k in missing), missing
    with torch.no_grad():
        block.ff.net[2].weight.zero_()
        block.ff.net[2].bias.zero_()
    return block


def temporal_model_from_video_transformer_fixture(cls, t, meta):
    """A TransformerTemporalModel of this repo set up so that `model(x, num_frames) - x` must equal the
    reference-authored VideoTransformer's temporal path on the fixture's weights (tests/golden/
    ref_video_transformer_temporal: src/modules/attention.py:79-131, `(b t) s c -> (b s) t c`, BasicTransformerBlock over
    the frame axis, back):
      * transformer block <- the reference block's attn1 / attn2 / norm1 / norm2; feed-forward off (ff.net.2 = 0);
        positional table zeroed (the reference path under test adds none);
      * proj_in = proj_out = identity;
      * the entry GroupNorm made the identity ON THIS INPUT: with ONE clip its statistics are per group only, so
        gamma = sqrt(var_g + eps), beta = mean_g (fp64 on the host) give GN(x) = x up to the rounding of the affine."""
    heads, d, frames = int(meta["heads"]), int(meta["head_dim"]), int(meta["frames"])
    c = heads * d
    torch.manual_seed(0)
    m = cls(num_attention_heads=heads, attention_head_dim=d, in_channels=c, norm_num_groups=32,
            attention_bias=False, activation_fn="geglu", positional_embeddings="sinusoidal",
            num_positional_embeddings=32).eval()
    load_ref_block_into(m.transformer_blocks[0], {k: v.float() for k, v in t.items()})
    x = t["x"].double()                                                  # (frames, c, h, w): one clip
    xg = x.permute(1, 0, 2, 3).reshape(32, -1)                            # group g = channels [16 g, 16 g + 16)
    mean, var = xg.mean(dim=1), xg.var(dim=1, unbiased=False)
    with torch.no_grad():
        m.norm.weight.copy_((var + 1e-6).sqrt().repeat_interleave(c // 32))
        m.norm.bias.copy_(mean.repeat_interleave(c // 32))
        for lin in (m.proj_in, m.proj_out):
            lin.weight.copy_(torch.eye(c))
            lin.bias.zero_()
        m.transformer_blocks[0].pos_embed.pe.zero_()
    return m, frames


def test_temporal_model_matches_reference_video_transformer():
    """the oracle's motion-module reshapes + double self-attention over frames (A9) == the reference's VideoTransformer
    temporal path on its weights."""
    from oracle.blocks import TransformerTemporalModel
    t, meta = _load_ref(os.path.join(GOLD, "ref_video_transformer_temporal.safetensors"))
    m, frames = temporal_model_from_video_transformer_fixture(TransformerTemporalModel, t, meta)
    with torch.no_grad():
        y = m(t["x"], num_frames=frames)[0] - t["x"]
    err = (y - t["y"]).abs().max().item()
    assert err <= 2e-5 * t["y"].abs().max().item(), err


def test_functional_restatement_of_reference_resblock_forward():
    """ResBlock.forward (src/modules/resnet.py:63-72) restated with the torch functional ops the HIP composition test
    mirrors (tests/test_golden_gpu.py): pins that reading of the reference."""
    t, meta = _load_ref(os.path.join(GOLD, "ref_resblock_forward.safetensors"))
    y = resblock_forward_functional(t, F.conv2d, lambda v, w, b: F.group_norm(v, 8, w, b, 1e-5), F.gelu, F.silu, F.linear)
    assert torch.allclose(y, t["y"], atol=5e-6, rtol=1e-5), (y - t["y"]).abs().max()


def resblock_forward_functional(t, conv2d, group_norm, gelu, silu, linear):
    x = t["x"]
    hmid = gelu(group_norm(conv2d(x, t["conv1.0.weight"], None, padding=1), t["conv1.1.weight"], t["conv1.1.bias"]))
    emb = linear(silu(linear(t["timesteps"], t["emb_layer.0.weight"], t["emb_layer.0.bias"])),
                 t["emb_layer.2.weight"], t["emb_layer.2.bias"])
    hmid = hmid + emb[..., None, None]
    out = gelu(group_norm(conv2d(hmid, t["conv2.0.weight"], None, padding=1), t["conv2.1.weight"], t["conv2.1.bias"]))
    return out + conv2d(x, t["res_conv.weight"], t["res_conv.bias"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ref_transformer_block_*.safetensors"))))
def test_block_composition_matches_reference_basic_transformer_block(path):
    """LayerNorm -> attention -> residual, twice (i2v:444-445, 468-473, 501, 510-533) == the reference-authored
    BasicTransformerBlock on the same weights, for the oracle's plain block and its adapter subclass."""
    from oracle.blocks import BasicTransformerBlock
    from oracle.i2v_adapter import I2VAdapterTransformerBlock
    t, meta = _load_ref(path)
    heads, d = int(meta["heads"]), int(meta["head_dim"])
    ctx_dim = t["ctx"].shape[-1]
    for cls in (BasicTransformerBlock, I2VAdapterTransformerBlock):
        torch.manual_seed(0)
        blk = load_ref_block_into(cls(heads * d, heads, d, cross_attention_dim=ctx_dim).eval(), t)
        with torch.no_grad():
            y = blk(t["x"], encoder_hidden_states=t["ctx"])
        assert torch.allclose(y, t["y"], atol=5e-6, rtol=1e-5), (cls.__name__, (y - t["y"]).abs().max())


def test_sinusoid_matches_reference_positional_emb():
    """oracle Timesteps (A1: [cos | sin], unet:763) and the motion-module table (A10) use the frequencies of the
    reference-authored positional_emb (src/modules/util.py:4-8: [sin | cos], 1 / 10000^(2 i / C))."""
    from oracle.blocks import SinusoidalPositionalEmbedding, Timesteps
    t = load_file(os.path.join(GOLD, "ref_positional_emb.safetensors"))
    for c in (320, 32):
        ref = t[f"emb{c}"]
        e = Timesteps(c, True, 0)(t["t"][:, 0])
        assert torch.allclose(e[:, : c // 2], ref[:, c // 2:], atol=2e-4)      # cos half (fp32 sin/cos of t <= 999)
        assert torch.allclose(e[:, c // 2:], ref[:, : c // 2], atol=2e-4)      # sin half
    pe = SinusoidalPositionalEmbedding(32, 32).pe[0]                            # interleaved sin / cos, same frequencies
    ref = t["emb32"]
    for row, pos in ((0, 0), (1, 1), (2, 2), (3, 16)):
        assert torch.allclose(pe[pos, 0::2], ref[row, :16], atol=1e-5) and torch.allclose(pe[pos, 1::2], ref[row, 16:], atol=1e-5)


def test_add_noise_known_answer():
    """reference test/test_first_frame_pertubation.py:17-39: with noise[:, 0] = 0 the first frame is exactly
    x * sqrt(alphas_cumprod[t]) (torch.eq), DDPMScheduler(1000) default linear betas."""
    from oracle.blocks import DDPMScheduler
    torch.manual_seed(0)
    sch = DDPMScheduler(num_train_timesteps=1000)
    ts = torch.randint(low=1, high=1000, size=(4,))
    x = torch.randn(4, 16, 4, 32, 32)
    noise = torch.randn(x.shape)
    noise[:, 0] = 0
    noised = sch.add_noise(x, noise, ts)
    sa = (sch.alphas_cumprod[ts] ** 0.5).flatten()
    while sa.dim() < x[:, 0].dim():
        sa = sa.unsqueeze(-1)
    assert torch.all(torch.eq(noised[:, 0], x[:, 0] * sa))
    gold = load_file(os.path.join(GOLD, "oracle_outputs.safetensors"))["ddpm_sqrt_alphas_cumprod"]
    assert torch.equal(sch.alphas_cumprod ** 0.5, gold)
    assert abs(float(sch.betas[0]) - 1e-4) < 1e-10 and abs(float(sch.betas[-1]) - 0.02) < 1e-8


def test_ddim_schedule_and_step():
    from oracle.blocks import DDIMScheduler
    s = DDIMScheduler()
    s.set_timesteps(25)
    ts = s.timesteps.tolist()
    assert ts[0] == 999 and ts[-1] == 0 and len(ts) == 25 and ts[1] == 957      # round(linspace(0, 999, 25))[::-1]
    assert abs(float(s.betas[0]) - 0.00085) < 1e-9 and abs(float(s.betas[-1]) - 0.012) < 1e-8
    x, eps = torch.randn(2, 3), torch.randn(2, 3)
    t = 957
    prev = t - 1000 // 25
    a_t, a_p = s.alphas_cumprod[t], s.alphas_cumprod[prev]
    x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
    assert torch.allclose(s.step(eps, t, x), a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps)
    # last step: t_prev < 0 -> final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one=False)
    x0 = (x - (1 - s.alphas_cumprod[0]).sqrt() * eps) / s.alphas_cumprod[0].sqrt()
    assert torch.allclose(s.step(eps, 0, x), s.alphas_cumprod[0].sqrt() * x0 + (1 - s.alphas_cumprod[0]).sqrt() * eps)


def test_functional_ops_agree():
    """independent per-op references: explicit-softmax SDPA vs F.scaled_dot_product_attention, GEGLU, timesteps."""
    from oracle.blocks import Attention, GEGLU, SinusoidalPositionalEmbedding, Timesteps
    torch.manual_seed(1)
    a = Attention(64, heads=4, dim_head=16)
    x = torch.randn(3, 10, 64)
    q, k, v = (a._split(f(x)) for f in (a.to_q, a.to_k, a.to_v))
    assert torch.allclose(a._sdpa(q, k, v), F.scaled_dot_product_attention(q, k, v), atol=1e-6)
    gg = GEGLU(8, 16)
    y = gg.proj(torch.ones(2, 8))
    assert torch.allclose(gg(torch.ones(2, 8)), y[:, :16] * F.gelu(y[:, 16:]))
    e = Timesteps(320, True, 0)(torch.tensor([3.0]))
    assert e.shape == (1, 320) and abs(float(e[0, 0]) - torch.cos(torch.tensor(3.0)).item()) < 1e-6   # [cos | sin]
    assert abs(float(e[0, 160]) - torch.sin(torch.tensor(3.0)).item()) < 1e-6
    pe = SinusoidalPositionalEmbedding(8, 32).pe
    assert pe.shape == (1, 32, 8) and float(pe[0, 0, 1]) == 1.0 and float(pe[0, 0, 0]) == 0.0


def test_reference_shape_contracts():
    """the reference's own (shape-only) tests, at reduced batch: test_i2v_adapter.py:11-124,
    test_unet_motion_cross_frame_attn.py:18-149."""
    from oracle.i2v_adapter import I2VAdapterTransformer2DModel, I2VAdapterTransformerBlock
    from oracle.unet_motion_cross_frame_attn import CrossFrameAttnDownBlockMotion
    torch.manual_seed(0)
    with torch.no_grad():
        blk = I2VAdapterTransformerBlock(256, 8, 32, dropout=0.0, cross_attention_dim=512, activation_fn="gelu")
        assert blk(torch.randn(8, 64, 256), enable_cross_frame_attn=True, num_frames=4,
                   encoder_hidden_states=torch.randn(8, 77, 512)).shape == (8, 64, 256)
        with pytest.raises(ValueError):
            blk(torch.randn(8, 64, 256), enable_cross_frame_attn=True, encoder_hidden_states=torch.randn(8, 77, 512))
        with pytest.raises(ValueError):
            blk(torch.randn(6, 64, 256), enable_cross_frame_attn=True, num_frames=4,
                encoder_hidden_states=torch.randn(6, 77, 512))
        t2d = I2VAdapterTransformer2DModel(8, 8, in_channels=64, out_channels=64, num_layers=1, cross_attention_dim=128,
                                           norm_num_groups=32)
        assert t2d(torch.randn(4, 64, 8, 8), enable_cross_frame_attn=True, num_frames=2,
                   encoder_hidden_states=torch.randn(4, 77, 128), return_dict=False)[0].shape == (4, 64, 8, 8)
        db = CrossFrameAttnDownBlockMotion(in_channels=64, out_channels=128, temb_channels=512, cross_attention_dim=768,
                                           num_layers=2, num_attention_heads=8)
        y, states = db(torch.randn(4, 64, 8, 8), temb=torch.randn(4, 512), enable_cross_frame_attn=True,
                       encoder_hidden_states=torch.randn(4, 77, 768), num_frames=2)
        assert y.shape == (4, 128, 4, 4) and len(states) == 3 and states[0].shape == (4, 128, 8, 8)


def test_oracle_regression_pins():
    gold = load_file(os.path.join(GOLD, "oracle_outputs.safetensors"))
    inp = small_unet_inputs()
    ou = oracle_small_unet()
    with torch.no_grad():
        y = ou(inp["sample"], inp["timestep"], True, inp["ctx"]).sample
        y2 = ou(inp["sample"], inp["timestep"], False, inp["ctx"]).sample
    assert torch.allclose(y, gold["unet_y"], atol=1e-5, rtol=1e-4)
    assert torch.allclose(y2, gold["unet_y_no_cross_frame"], atol=1e-5, rtol=1e-4)
    assert (y - y2).abs().max() > 1e-3          # the adapter branch contributes (non-zero to_out)


def test_unet_assembly_keys_and_ip_order():
    """state-dict key layout and IP-Adapter key ids 1, 3, ..., 31 in attn_processors order."""
    ou = oracle_small_unet()
    keys = set(ou.state_dict())
    for k in ["conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.time_emb_proj.bias",
              "down_blocks.0.attentions.1.transformer_blocks.0.i2v_adapter.to_out.0.bias",
              "down_blocks.3.motion_modules.1.transformer_blocks.0.pos_embed.pe",
              "up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k.weight",
              "up_blocks.0.motion_modules.2.transformer_blocks.0.ff.net.0.proj.weight",
              "mid_block.attentions.0.proj_in.weight", "up_blocks.3.resnets.2.conv_shortcut.weight",
              "down_blocks.2.downsamplers.0.conv.bias", "up_blocks.2.upsamplers.0.conv.weight", "conv_norm_out.weight"]:
        assert k in keys, k
    assert not any(k.startswith("down_blocks.3.attentions") or k.startswith("up_blocks.0.attentions") for k in keys)
    names = [n for n in ou.attn_processor_names() if n.endswith("attn2.processor") and "motion_modules" not in n]
    assert len(names) == 16 and names[0].startswith("down_blocks.0") and names[-1].startswith("mid_block")
    assert names[6].startswith("up_blocks.1")
    adapter = ou.obtain_i2v_adapter_modules()
    assert len(adapter.state_dict()) == 16 * 5


def test_ddim_step_with_eta_matches_the_published_update():
    """diffusers DDIMScheduler.step with eta (pipe:550, 659-660): x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev -
    sigma^2) eps + sigma z, sigma = eta sqrt((1 - a_prev) / (1 - a_t)) sqrt(1 - a_t / a_prev); eta = 1 is the DDPM posterior
    (its variance is beta_tilde), eta = 0 the deterministic update; the product scheduler's coefficient table agrees."""
    import sys, os
    from oracle.blocks import DDIMScheduler as O
    sch = O()
    sch.set_timesteps(25)
    g = torch.Generator().manual_seed(0)
    x, eps, z = (torch.randn(2, 3, 4, 8, 8, generator=g) for _ in range(3))
    t = int(sch.timesteps[3])
    a_t, a_p = sch.alphas_cumprod[t], sch.alphas_cumprod[t - 1000 // 25]
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    assert torch.equal(sch.step(eps, t, x, eta=0.0), a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps)
    for eta in (0.3, 1.0):
        sig = eta * ((1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) ** 0.5
        want = a_p ** 0.5 * x0 + (1 - a_p - sig ** 2) ** 0.5 * eps + sig * z
        assert torch.allclose(sch.step(eps, t, x, eta=eta, variance_noise=z), want, atol=1e-6)
    # eta = 1: sigma^2 is the DDPM posterior variance beta_tilde = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
    assert abs(float(((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))) - float(sig ** 2)) < 1e-6
    # the same draw from a generator
    a = sch.step(eps, t, x, eta=0.5, generator=torch.Generator().manual_seed(9))
    b = sch.step(eps, t, x, eta=0.5, variance_noise=torch.randn(x.shape, generator=torch.Generator().manual_seed(9)))
    assert torch.equal(a, b)
    import i2v_adapter_unofficial_amd as pkg
    ps = pkg.DDIMScheduler()
    ps.set_timesteps(25)
    co = ps.step_coefficients(ps.timesteps, 0.3)
    sg = ps.step_sigmas(ps.timesteps, 0.3)
    s3 = 0.3 * float(((1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) ** 0.5)
    assert abs(sg[3] - s3) < 1e-7 and abs(float(co[3, 3]) - float((1 - a_p - s3 ** 2) ** 0.5)) < 1e-6
    assert torch.equal(ps.step_coefficients(ps.timesteps), ps.step_coefficients(ps.timesteps, 0.0))
