"""Generate the golden vectors under tests/golden/ (TEST INFRASTRUCTURE; run in the build container only).

    python -m oracle.make_golden

Two families:

1. `ref_attention_*.safetensors` -- produced by IMPORTING the reference-authored
   /root/reference/src/modules/attention.py:26-62 `BasicAttention` (the same op as diffusers
   `Attention` + `AttnProcessor2_0`: bias-free q/k/v, SDPA, biased out-projection) on seeded inputs, in the three
   forms the hot path uses it: K2 self-attention, K1 cross-frame (context = frame-0 tokens of each clip repeated
   over the frames, i2v:484-485) and K3 text cross-attention.  These pin the oracle's `Attention` (and, on the GPU,
   the HIP attention path) to reference-authored code.  The reference's Python never leaves this container; only
   these tensors do.

2. `oracle_*.safetensors` -- outputs of the oracle itself on seeded weights / inputs (weights are NOT stored: they
   are re-created from the recorded seed with torch's default initialisers), as regression pins of the restatement
   and as fixtures for the HIP parity tests: transformer block (reference test shape), down block, reduced UNet with
   and without IP tokens, a 10-step DDIM trajectory, and the add_noise known-answer test values.
"""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def h(t):
    return t.half().float()


def ref_attention_goldens():
    sys.path.insert(0, REFERENCE)
    from src.modules.attention import BasicAttention   # reference-authored, imported only here
    cases = {
        # name: (kind, clips, frames, tokens, heads, head_dim, ctx_tokens, ctx_dim)
        "self_d8": ("self", 2, 4, 64, 8, 8, None, None),
        "cross_frame_d8": ("cross_frame", 2, 4, 64, 8, 8, None, None),
        "cross_frame_d40": ("cross_frame", 1, 3, 80, 2, 40, None, None),
        "cross_frame_d80": ("cross_frame", 1, 2, 64, 2, 80, None, None),
        "cross_frame_d160": ("cross_frame", 1, 2, 48, 2, 160, None, None),
        "text_d40": ("text", 2, 2, 64, 2, 40, 77, 96),
    }
    for name, (kind, clips, frames, tokens, heads, d, lctx, dctx) in cases.items():
        c = heads * d
        torch.manual_seed(sum(map(ord, name)))
        attn = BasicAttention(c, dctx if kind == "text" else c, head_dim=d, num_heads=heads)
        with torch.no_grad():
            for p in attn.parameters():
                p.copy_(h(p))
        g = torch.Generator().manual_seed(17)
        x = h(torch.randn(clips * frames, tokens, c, generator=g))
        if kind == "self":
            ctx = None
        elif kind == "cross_frame":
            ctx = x[0:clips * frames:frames].repeat_interleave(frames, dim=0)      # i2v:484-485
        else:
            ctx = h(torch.randn(clips * frames, lctx, dctx, generator=g))
        with torch.no_grad():
            y = attn(x, ctx)
        t = {"x": x, "y": y, "to_q": attn.to_q.weight.detach(), "to_k": attn.to_k.weight.detach(),
             "to_v": attn.to_v.weight.detach(), "to_out_w": attn.to_out[0].weight.detach(),
             "to_out_b": attn.to_out[0].bias.detach()}
        if kind == "text":
            t["ctx"] = ctx
        meta = dict(kind=kind, clips=str(clips), frames=str(frames), heads=str(heads), head_dim=str(d))
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, f"ref_attention_{name}.safetensors"),
                  metadata=meta)
        print("wrote ref_attention_" + name, tuple(y.shape))


def ref_block_goldens():
    """More reference-authored code that imports without diffusers (round 2):
    * `BasicTransformerBlock` (src/modules/attention.py:64-77): x = attn1(LN1(x)) + x; x = attn2(LN2(x), ctx) + x --
      pins the LayerNorm -> attention -> residual COMPOSITION of the spatial block (i2v:444-445, 468-473, 501,
      510-533) for both the oracle and the HIP block (their GEGLU feed-forward is switched off by zeroing ff.net.2);
    * `positional_emb` (src/modules/util.py:4-8): sinusoid frequencies 1 / 10000^(2i / C), [sin | cos] order (the hot
      path's Timesteps(flip_sin_to_cos=True) is [cos | sin], unet:763);
    * `ResBlock` (src/modules/resnet.py:19-72): bias-free conv3x3 -> GroupNorm(8) (output captured before the GELU)
      and the 1x1 `res_conv`: conv + GroupNorm kernels against a reference-authored composition."""
    sys.path.insert(0, REFERENCE)
    from src.modules.attention import BasicTransformerBlock
    from src.modules.resnet import ResBlock
    from src.modules.util import positional_emb
    for name, (c, dctx, d, heads, batch, tokens, lctx) in {
            "d40": (320, 768, 40, 8, 2, 96, 77), "d8": (64, 96, 8, 8, 3, 40, 7)}.items():
        torch.manual_seed(1000 + c)
        blk = BasicTransformerBlock(c, dctx, head_dim=d, num_heads=heads).eval()
        with torch.no_grad():
            for n, p in blk.named_parameters():
                if "norm" in n:       # default LayerNorm affine is (1, 0): make it visible
                    p.add_(0.2 * torch.randn(p.shape))
                p.copy_(h(p))
        g = torch.Generator().manual_seed(23)
        x, ctx = h(torch.randn(batch, tokens, c, generator=g)), h(torch.randn(batch, lctx, dctx, generator=g))
        with torch.no_grad():
            y = blk(x, ctx)
        t = {"x": x, "ctx": ctx, "y": y}
        t.update({k: v.detach().clone() for k, v in blk.state_dict().items()})
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, f"ref_transformer_block_{name}.safetensors"),
                  metadata=dict(heads=str(heads), head_dim=str(d)))
        print("wrote ref_transformer_block_" + name, tuple(y.shape))
    tt = torch.tensor([[0.0], [1.0], [2.0], [16.0], [40.0], [481.0], [999.0]])
    save_file({"t": tt, "emb320": positional_emb(tt, 320).contiguous(), "emb32": positional_emb(tt, 32).contiguous()},
              os.path.join(OUT, "ref_positional_emb.safetensors"))
    print("wrote ref_positional_emb")
    torch.manual_seed(77)
    rb = ResBlock(64, 128, 32, group_nums=8).eval()
    with torch.no_grad():
        for n, p in rb.named_parameters():
            if n in ("conv1.1.weight", "conv1.1.bias"):
                p.add_(0.2 * torch.randn(p.shape))
            p.copy_(h(p))
    cap = {}
    rb.conv1[0].register_forward_hook(lambda m, a, o: cap.__setitem__("conv", o.detach()))
    rb.conv1[1].register_forward_hook(lambda m, a, o: cap.__setitem__("gn", o.detach()))
    g = torch.Generator().manual_seed(78)
    x = h(torch.randn(3, 64, 12, 12, generator=g))
    with torch.no_grad():
        rb(x, h(torch.randn(3, 32, generator=g)))
        res = rb.res_conv(x)
    save_file({"x": x, "conv_w": rb.conv1[0].weight.detach().contiguous(), "gn_w": rb.conv1[1].weight.detach().contiguous(),
               "gn_b": rb.conv1[1].bias.detach().contiguous(), "y_conv": cap["conv"].contiguous(),
               "y_conv_gn": cap["gn"].contiguous(), "res_w": rb.res_conv.weight.detach().contiguous(),
               "res_b": rb.res_conv.bias.detach().contiguous(), "y_res": res.contiguous()},
              os.path.join(OUT, "ref_resblock_conv_gn.safetensors"), metadata=dict(groups="8", eps="1e-05"))
    print("wrote ref_resblock_conv_gn")


def ref_temporal_and_resblock_goldens():
    """Round 3: two more compositions executed by reference-authored code (imports without diffusers).

    * `VideoTransformer.forward` (src/modules/attention.py:79-131): the reference's frames-as-sequence path --
      `(b t) s c -> (b s) t c`, a `BasicTransformerBlock` over the frame axis (LN -> self-attention -> +x, twice: with
      no context its attn2 is self-attention too, :52), `(b s) t c -> (b t) s c` -- is exactly the geometry of the motion
      module's double self-attention that the HIP path runs in (b, pixel, frame) row order.  The forward is
      run WHOLE; the parts that are not under test are neutralised through their parameters (spatial attn1 / attn2
      `to_out` zeroed => x_spatial = x; `frame_pos_embed` last Linear zeroed => emb_out = 0; time_mixer mix_factor =
      -100 => alpha = sigmoid(-100) = 0 => output = x_temporal), so y - x is the temporal block's output, every
      rearrangement done by the reference.  8 heads x 64 (the class's fixed defaults), 16 frames.
    * `ResBlock.forward` (src/modules/resnet.py:63-72), whole: conv3x3 -> GroupNorm -> GELU, + emb_layer(t) broadcast
      over pixels (Linear -> SiLU -> Linear), conv3x3 -> GroupNorm -> GELU, + 1x1 res_conv(x)."""
    sys.path.insert(0, REFERENCE)
    from src.modules.attention import VideoTransformer
    from src.modules.resnet import ResBlock
    torch.manual_seed(2024)
    c, frames, hh, ww = 512, 16, 4, 4
    vt = VideoTransformer(c).eval()
    with torch.no_grad():
        for a in (vt.attn1, vt.attn2):
            a.to_out[0].weight.zero_()
            a.to_out[0].bias.zero_()
        vt.frame_pos_embed[2].weight.zero_()
        vt.frame_pos_embed[2].bias.zero_()
        vt.time_mixer.mix_factor.fill_(-100.0)
        for n, p in vt.video_attn.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn(p.shape))
            p.copy_(h(p))
    g = torch.Generator().manual_seed(2025)
    x = h(torch.randn(frames, c, hh, ww, generator=g))                     # one clip: (b t) c h w with b = 1
    with torch.no_grad():
        y = vt(x, None, frames, torch.zeros(1, frames)) - x
    t = {"x": x, "y": y}
    t.update({k: v.detach().half() for k, v in vt.video_attn.state_dict().items()})   # fp16-representable: stored as fp16
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, "ref_video_transformer_temporal.safetensors"),
              metadata=dict(heads="8", head_dim="64", frames=str(frames)))
    print("wrote ref_video_transformer_temporal", tuple(y.shape), float(y.abs().max()))

    torch.manual_seed(88)
    rb = ResBlock(64, 128, 32, group_nums=8).eval()
    with torch.no_grad():
        for n, p in rb.named_parameters():
            if n in ("conv1.1.weight", "conv1.1.bias", "conv2.1.weight", "conv2.1.bias"):
                p.add_(0.2 * torch.randn(p.shape))
            p.copy_(h(p))
    g = torch.Generator().manual_seed(89)
    x = h(torch.randn(3, 64, 12, 12, generator=g))
    ts = h(torch.randn(3, 32, generator=g))
    with torch.no_grad():
        y = rb(x, ts)
    t = {"x": x, "timesteps": ts, "y": y}
    t.update({k: v.detach().clone() for k, v in rb.state_dict().items()})
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, "ref_resblock_forward.safetensors"),
              metadata=dict(groups="8", eps="1e-05"))
    print("wrote ref_resblock_forward", tuple(y.shape), float(y.abs().max()))


def oracle_goldens():
    from oracle.blocks import DDPMScheduler
    from oracle.i2v_adapter import I2VAdapterTransformerBlock
    from oracle.pipeline_i2v_adapter import I2VAdapterPipeline
    from oracle.unet_motion_cross_frame_attn import CrossFrameAttnDownBlockMotion
    from tests.parity import (oracle_small_unet, round_fp16_, small_ip_state_dict, small_unet_inputs)
    out = {}
    with torch.no_grad():
        # transformer block, reference test shape (test/test_i2v_adapter.py:73-110) with 2 clips x 4 frames
        torch.manual_seed(101)
        blk = round_fp16_(I2VAdapterTransformerBlock(256, 8, 32, dropout=0.0, cross_attention_dim=512,
                                                     activation_fn="gelu")).eval()
        g = torch.Generator().manual_seed(102)
        x, ctx = h(torch.randn(8, 64, 256, generator=g)), h(torch.randn(8, 77, 512, generator=g))
        out["block_y_cross_frame"] = blk(x, enable_cross_frame_attn=True, num_frames=4, encoder_hidden_states=ctx)
        out["block_y_plain"] = blk(x, enable_cross_frame_attn=False, encoder_hidden_states=ctx)
        # down block (test/test_unet_motion_cross_frame_attn.py:18-57) with 1 clip x 4 frames, 8 x 8
        torch.manual_seed(103)
        db = round_fp16_(CrossFrameAttnDownBlockMotion(in_channels=64, out_channels=128, temb_channels=512,
                                                       cross_attention_dim=768, num_layers=2,
                                                       num_attention_heads=8)).eval()
        g = torch.Generator().manual_seed(104)
        xs = h(torch.randn(4, 64, 8, 8, generator=g))
        temb = h(torch.randn(4, 512, generator=g))
        ctx2 = h(torch.randn(4, 77, 768, generator=g))
        y, states = db(hidden_states=xs, temb=temb, enable_cross_frame_attn=True, encoder_hidden_states=ctx2,
                       num_frames=4)
        out["down_block_y"] = y
        out["down_block_state0"] = states[0]
        # reduced UNet, without and with IP tokens
        inp = small_unet_inputs()
        ou = oracle_small_unet()
        out["unet_y"] = ou(inp["sample"], inp["timestep"], True, inp["ctx"]).sample
        out["unet_y_no_cross_frame"] = ou(inp["sample"], inp["timestep"], False, inp["ctx"]).sample
        oi = oracle_small_unet(ip=True)
        out["unet_y_ip"] = oi(inp["sample"], inp["timestep"], True, inp["ctx"],
                              added_cond_kwargs={"image_embeds": inp["image_embeds"]}).sample
        # DDIM trajectory (config-1 plumbing on the reduced UNet)
        g = torch.Generator().manual_seed(31)
        pe, ne = h(torch.randn(1, 7, 64, generator=g)), h(torch.randn(1, 7, 64, generator=g))
        cond = torch.randn(1, 4, 16, 16, generator=g)
        out["ddim_latents"] = I2VAdapterPipeline(ou)(
            pe, ne, cond, num_frames=4, num_inference_steps=10, guidance_scale=7.5, frame_similarity_sample_ratio=0.9,
            generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
            prior_noise_generator=torch.Generator().manual_seed(7), blur_sigma=1.0).frames
        # add_noise KAT values (test/test_first_frame_pertubation.py): sqrt(alphas_cumprod) of DDPMScheduler(1000)
        sch = DDPMScheduler(1000)
        out["ddpm_sqrt_alphas_cumprod"] = sch.alphas_cumprod ** 0.5
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, "oracle_outputs.safetensors"))
    for k, v in out.items():
        print("oracle", k, tuple(v.shape))


def ref_gradient_goldens():
    """Gradients of reference-authored code (round 3, pins the training step's backward -- SURVEY 8 f4 -- to something other
    than this repo's own oracle): torch autograd through the reference's `BasicTransformerBlock`
    (src/modules/attention.py:64-77: x = attn1(LN1(x)) + x; x = attn2(LN2(x), ctx) + x) on the weights and inputs of the
    committed ref_transformer_block_* fixtures (re-created from the same seeds and checked against their `y`), for a given
    output gradient dy: d / d x (LayerNorm backward, both attention backwards, the Linear dgrads, the residual paths) and the
    parameter gradients of attn1.to_out (a weight gradient dY^T X and a bias column sum, at the position where the I2V
    adapter's to_out adds, i2v:494).  Only dy and the gradients are stored."""
    from safetensors.torch import load_file
    sys.path.insert(0, REFERENCE)
    from src.modules.attention import BasicTransformerBlock
    for name, (c, dctx, d, heads, batch, tokens, lctx) in {
            "d40": (320, 768, 40, 8, 2, 96, 77), "d8": (64, 96, 8, 8, 3, 40, 7)}.items():
        torch.manual_seed(1000 + c)
        blk = BasicTransformerBlock(c, dctx, head_dim=d, num_heads=heads).eval()
        with torch.no_grad():
            for n, p in blk.named_parameters():
                if "norm" in n:
                    p.add_(0.2 * torch.randn(p.shape))
                p.copy_(h(p))
        g = torch.Generator().manual_seed(23)
        x, ctx = h(torch.randn(batch, tokens, c, generator=g)), h(torch.randn(batch, lctx, dctx, generator=g))
        x.requires_grad_()
        dy = h(torch.randn(batch, tokens, c, generator=torch.Generator().manual_seed(31)))
        y = blk(x, ctx)
        old = load_file(os.path.join(OUT, f"ref_transformer_block_{name}.safetensors"))
        assert torch.equal(y.detach(), old["y"]) and torch.equal(x.detach(), old["x"]), "not the committed fixture's block"
        y.backward(dy)
        t = {"dy": dy, "dx": x.grad.detach(), "d_attn1_to_out_weight": blk.attn1.to_out[0].weight.grad.detach(),
             "d_attn1_to_out_bias": blk.attn1.to_out[0].bias.grad.detach()}
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(OUT, f"ref_grads_transformer_block_{name}.safetensors"),
                  metadata=dict(heads=str(heads), head_dim=str(d)))
        print("wrote ref_grads_transformer_block_" + name, tuple(x.grad.shape))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if os.path.isdir(REFERENCE):
        if "--only-new" not in sys.argv:
            ref_attention_goldens()
            ref_block_goldens()
            ref_temporal_and_resblock_goldens()
        ref_gradient_goldens()
    else:
        print("reference not present: keeping the committed ref_attention_* fixtures")
    if "--only-new" not in sys.argv:
        oracle_goldens()
