"""GPU parity of the model that bench.py times: SD-1.5-width UNetMotionCrossFrameAttnModel (channels 320 / 640 / 1280 /
1280, head_dim 40 / 80 / 160) against the CPU oracle on the same fp16-representable weights.

* one CFG forward (B = 2) at BASELINE config 1's shape (8 f x 256^2 => sample (2, 8, 4, 32, 32)), with and without
  cross-frame attention and with the IP-Adapter branch: the composed 8-wave LDS-DMA GEMM / conv kernels (N % 320 == 0),
  split-K, head_dim 40 / 80 / 160 flash attention over the fused [q | k | q_adapter] strided views, the dual-source
  out-projection, the LDS temporal attention and the ROWPERM store, as unet:1289-1451 composes them;
* module-level cases at the shapes of config 2's levels (C = 320 / 640 at 32 x 32, C = 1280 at 8 x 8);
* full-size PROPERTY tests for configs 2, 3 and 5 (the oracle cannot run those sizes in seconds): finite, frame 0 ==
  condition latents exactly (pipe:699-700), eager == hipGraph bit for bit, run-to-run identical.

Tolerances (measured on MI355X in round 2, profiles/r2_parity_errors.jsonl; asserted bounds <= 3x the measured error:
UNet forward 1.5e-3 .. 2.1e-3 max-abs at max|ref| 1.3, modules 3.8e-4 .. 6.4e-4 of max|ref|):
the binding gate is the fixed tolerance against the fp32 oracle; the forward cases also run the oracle in its
fp16-emulating mode (oracle/fp16_emulation.py: every aten op's result rounded to fp16) and LOG that distance next to the
HIP error as information.
"""
import pytest
import torch

from tests.parity import (SD15, compare, full_width_pair, hip_model_random, host_threads, log_error,
                          oracle_from_hip)

pytestmark = pytest.mark.gpu

# asserted bounds: max-abs error of one full-width CFG forward against the fp32 oracle (max|ref| ~ 1.3), and the factor
# by which the HIP error may exceed the fp16-emulated reference's own error
# (r4: <= 2x the measured errors -- forwards 1.9e-3 .. 2.14e-3, modules <= 6.5e-4 of max; profiles/r4_parity_errors.jsonl)
FWD_ABS_TOL = 4.2e-3
MODULE_REL_TOL = 1.4e-3


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def h(t):
    return t.half().float()


@pytest.fixture(scope="module")
def pair(dev):
    host_threads()
    return full_width_pair(dev, seed=1234, ip=False)


@pytest.fixture(scope="module")
def pair_ip(dev):
    host_threads()
    return full_width_pair(dev, seed=4321, ip=True)


def _inputs(frames=8, hw=32, seed=3, clip_dim=1024):
    g = torch.Generator().manual_seed(seed)
    return dict(sample=h(torch.randn(2, frames, 4, hw, hw, generator=g)), ctx=h(torch.randn(2, 77, 768, generator=g)),
                image_embeds=h(torch.randn(2, clip_dim, generator=g)), t=torch.tensor([481, 481]))


def _forward_three_ways(ou, hu, dev, inp, cross_frame, ip):
    from oracle.fp16_emulation import emulate_reference_fp16
    added = {"image_embeds": inp["image_embeds"]} if ip else None
    added_d = {"image_embeds": inp["image_embeds"].to(dev)} if ip else None
    with torch.no_grad():
        ref = ou(inp["sample"], inp["t"], cross_frame, inp["ctx"], added_cond_kwargs=added).sample
        with emulate_reference_fp16():
            emu = ou(inp["sample"], inp["t"], cross_frame, inp["ctx"], added_cond_kwargs=added).sample
        got = hu(inp["sample"].to(dev), inp["t"].to(dev), cross_frame, inp["ctx"].to(dev),
                 added_cond_kwargs=added_d).sample
    return ref, emu, got


@pytest.mark.parametrize("cross_frame", [True, False])
def test_full_width_unet_forward(dev, pair, cross_frame):
    ou, hu = pair
    inp = _inputs()
    ref, emu, got = _forward_three_ways(ou, hu, dev, inp, cross_frame, ip=False)
    assert got.shape == ref.shape == (2, 8, 4, 32, 32)
    err_emu = (emu - ref).abs().max().item()
    err, scale = compare(got, ref, abs_tol=FWD_ABS_TOL, name=f"full-width UNet forward cross_frame={cross_frame}")
    log_error(f"full-width fp16-emulated reference cross_frame={cross_frame}", err_emu, scale, None)
    print(f"full-width UNet cross_frame={cross_frame}: HIP err {err:.3e}, fp16-emulated reference err {err_emu:.3e}, "
          f"max|ref| {scale:.3e}")
    if cross_frame:
        # the adapter branch must contribute at full width too (K1 with head_dim 40 / 80 / 160)
        with torch.no_grad():
            off = hu(inp["sample"].to(dev), inp["t"].to(dev), False, inp["ctx"].to(dev)).sample
        assert (got - off).abs().max().item() > 10 * FWD_ABS_TOL


def test_full_width_unet_forward_upsample_size(dev, pair):
    """SD-1.5 width at 36 x 28 latents (288 x 224 pixels: divisible by 8 as pipe:213-214 demands, not by 64): the levels are 36 x 28,
    18 x 14, 9 x 7, 5 x 4, so two of the three up-samplers interpolate to 2 n - 1 (unet:1304-1311, 1414-1415 forward_upsample_size),
    the fused 320-channel kernels see 1008 rows per image (not a multiple of their 128-row tiles: the un-fused forms run) and every
    GEMM / conv / attention runs on ragged row counts -- against the oracle, same tolerance as the aligned sizes."""
    ou, hu = pair
    g = torch.Generator().manual_seed(21)
    sample, ctx = h(torch.randn(2, 4, 4, 36, 28, generator=g)), h(torch.randn(2, 77, 768, generator=g))
    t = torch.tensor([481, 481])
    with torch.no_grad():
        ref = ou(sample, t, True, ctx).sample
        got = hu(sample.to(dev), t.to(dev), True, ctx.to(dev)).sample
    assert got.shape == ref.shape == (2, 4, 4, 36, 28)
    err, scale = compare(got, ref, abs_tol=FWD_ABS_TOL, name="full-width UNet forward at 36 x 28 latents (forward_upsample_size)")
    print(f"full-width UNet at 36 x 28 latents: HIP err {err:.3e} (rms {(got.float().cpu() - ref).pow(2).mean().sqrt().item():.3e}), max|ref| {scale:.3e}")


def test_full_width_unet_forward_ip(dev, pair_ip):
    ou, hu = pair_ip
    inp = _inputs(seed=5)
    ref, emu, got = _forward_three_ways(ou, hu, dev, inp, True, ip=True)
    err_emu = (emu - ref).abs().max().item()
    err, scale = compare(got, ref, abs_tol=FWD_ABS_TOL, name="full-width UNet forward + IP-Adapter")
    log_error("full-width fp16-emulated reference + IP-Adapter", err_emu, scale, None)
    print(f"full-width UNet + IP: HIP err {err:.3e}, fp16-emulated reference err {err_emu:.3e}, max|ref| {scale:.3e}")
    with pytest.raises(ValueError, match="image_embeds"):
        hu(inp["sample"].to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev))
    # IP tokens change the prediction; the descriptor API can switch the branch off and on again (unet:1118-1161)
    procs = hu.attn_processors
    assert sum(p.num_tokens == 4 for p in procs.values()) == 16 and len(procs) == 16 * 3 + 21 * 2
    with torch.no_grad():
        hu.set_attn_processor({k: type(p)(0, 1.0) for k, p in procs.items()})
        off = hu(inp["sample"].to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev),
                 added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
        hu.set_attn_processor(procs)
        on = hu(inp["sample"].to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev),
                added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
    assert torch.equal(on, got) and (off - got).abs().max().item() > 10 * FWD_ABS_TOL


# ------------------------------------------------------------------------------------------------ module level
def _module_pair(hip_cls, oracle_cls, kwargs, dev, seed):
    hm = hip_model_random(kwargs, dev, seed=seed, cls=hip_cls)
    return oracle_from_hip(hm, oracle_cls, kwargs), hm


@pytest.mark.parametrize("c,hw,frames", [(320, 32, 16), (640, 32, 16), (1280, 8, 16), (1280, 16, 8)])
def test_full_width_transformer_2d(dev, c, hw, frames):
    """I2VAdapterTransformer2DModel at the widths / head dims of the SD-1.5 levels, cross-frame on (i2v:184-354)."""
    from oracle.i2v_adapter import I2VAdapterTransformer2DModel as O
    host_threads()
    kw = dict(num_attention_heads=8, attention_head_dim=c // 8, in_channels=c, num_layers=1, cross_attention_dim=768,
              norm_num_groups=32)
    o, m = _module_pair(pkg().I2VAdapterTransformer2DModel, O, kw, dev, seed=c + hw)
    g = torch.Generator().manual_seed(c)
    x = h(torch.randn(2 * frames, c, hw, hw, generator=g))
    ctx = h(torch.randn(2 * frames, 77, 768, generator=g))
    with torch.no_grad():
        ref = o(x, enable_cross_frame_attn=True, num_frames=frames, encoder_hidden_states=ctx, return_dict=False)[0]
        got = m(x.half().to(dev), enable_cross_frame_attn=True, num_frames=frames,
                encoder_hidden_states=ctx.half().to(dev), return_dict=False)[0]
    compare(got, ref, rel=MODULE_REL_TOL, name=f"full-width T2D C={c} {hw}x{hw} F={frames}")
    from i2v_adapter_unofficial_amd import i2v_adapter as mod, kernels as K
    fused_shape = K.cross_attn_fused_supported(2 * frames * hw * hw, c, 8, c // 8, 77, hw * hw)
    assert fused_shape == (c == 320)       # the 64^2-level width of SD-1.5 runs i2v_cross_attn_fused_f16 for norm2 -> attn2
    if fused_shape:      # the same module on the un-fused kernels (I2V_TEXT_FUSED=0): both against the oracle, and close
        assert mod.FUSED_TEXT_ATTN
        mod.FUSED_TEXT_ATTN = False
        try:
            with torch.no_grad():
                plain = m(x.half().to(dev), enable_cross_frame_attn=True, num_frames=frames,
                          encoder_hidden_states=ctx.half().to(dev), return_dict=False)[0]
        finally:
            mod.FUSED_TEXT_ATTN = True
        compare(plain, ref, rel=MODULE_REL_TOL, name=f"full-width T2D, un-fused text attention C={c}")
        compare(got, plain, rel=MODULE_REL_TOL, name=f"T2D fused vs un-fused text attention C={c}")
        assert not torch.equal(got, plain)
        # ... and with to_out + residual as a GEMM launch behind the fused attention (I2V_ATTN_OUTP=0)
        assert mod.FUSED_ATTN_OUT
        mod.FUSED_ATTN_OUT = False
        try:
            with torch.no_grad():
                pair = m(x.half().to(dev), enable_cross_frame_attn=True, num_frames=frames,
                         encoder_hidden_states=ctx.half().to(dev), return_dict=False)[0]
        finally:
            mod.FUSED_ATTN_OUT = True
        compare(pair, ref, rel=MODULE_REL_TOL, name=f"full-width T2D, to_out as a GEMM C={c}")
        compare(got, pair, rel=MODULE_REL_TOL, name=f"T2D to_out inside vs behind the fused text attention C={c}")


@pytest.mark.parametrize("c,hw,frames", [(320, 32, 16), (640, 16, 16), (1280, 8, 16), (320, 16, 32), (320, 32, 8)])
def test_full_width_motion_module(dev, c, hw, frames):
    from oracle.blocks import TransformerTemporalModel as O
    host_threads()
    kw = dict(num_attention_heads=8, attention_head_dim=c // 8, in_channels=c, norm_num_groups=32,
              attention_bias=False, activation_fn="geglu", positional_embeddings="sinusoidal",
              num_positional_embeddings=32)
    o, m = _module_pair(pkg().TransformerTemporalModel, O, kw, dev, seed=c + hw + 1)
    x = h(torch.randn(2 * frames, c, hw, hw, generator=torch.Generator().manual_seed(c + 1)))
    from i2v_adapter_unofficial_amd import blocks, kernels as K
    fused_shape = K.motion_attn_supported(2 * frames * hw * hw, c, 8, c // 8, frames)
    # the 64^2-level width of SD-1.5 runs i2v_motion_attn_f16 -- with 16 frames, with the 8 of configs[0] and (r5) the 32 of
    # configs[4] (32 f x 768^2: BOTH routes of that configuration are checked against the oracle here)
    assert fused_shape == (c == 320)
    with torch.no_grad():
        ref = o(x, num_frames=frames)[0]
        got = m(x.half().to(dev), num_frames=frames)[0]
        compare(got, ref, rel=MODULE_REL_TOL, name=f"full-width motion module C={c} {hw}x{hw} F={frames}")
        if fused_shape:      # the same module on the un-fused kernels (I2V_MOTION_FUSED=0): both against the oracle, and close
            assert blocks.FUSED_MOTION_ATTN
            blocks.FUSED_MOTION_ATTN = False
            try:
                plain = m(x.half().to(dev), num_frames=frames)[0]
            finally:
                blocks.FUSED_MOTION_ATTN = True
            compare(plain, ref, rel=MODULE_REL_TOL, name=f"full-width motion module, un-fused attention sub-block C={c}")
            compare(got, plain, rel=MODULE_REL_TOL, name=f"motion module fused vs un-fused C={c}")
            assert not torch.equal(got, plain)               # (different roundings: the two paths really differ)
            # to_out + residual as a GEMM launch behind the fused attention (I2V_ATTN_OUTP=0)
            assert blocks.FUSED_ATTN_OUT
            blocks.FUSED_ATTN_OUT = False
            try:
                pair = m(x.half().to(dev), num_frames=frames)[0]
            finally:
                blocks.FUSED_ATTN_OUT = True
            compare(pair, ref, rel=MODULE_REL_TOL, name=f"full-width motion module, to_out as a GEMM C={c}")
            compare(got, pair, rel=MODULE_REL_TOL, name=f"motion module to_out inside vs behind the fused attention C={c}")
            # the one-launch feed-forward (i2v_ff_fused_f16) against the LayerNorm-folded GEGLU GEMM + output GEMM (I2V_FF_FUSED=0)
            assert blocks.FUSED_FF and m.transformer_blocks[0].ff.fused_supported(torch.empty(2 * frames * hw * hw, c))
            blocks.FUSED_FF = False
            try:
                plain_ff = m(x.half().to(dev), num_frames=frames)[0]
            finally:
                blocks.FUSED_FF = True
            compare(plain_ff, ref, rel=MODULE_REL_TOL, name=f"full-width motion module, un-fused feed-forward C={c}")
            compare(got, plain_ff, rel=MODULE_REL_TOL, name=f"motion module fused vs un-fused feed-forward C={c}")
            assert not torch.equal(got, plain_ff)


@pytest.mark.parametrize("kind,c,hw,frames,gain", [("t2d", 320, 32, 16, 4.0), ("t2d", 1280, 8, 16, 3.0), ("motion", 320, 32, 16, 4.0),
                                                     ("motion", 640, 16, 16, 3.0)])
def test_full_width_sharp_attention(dev, kind, c, hw, frames, gain):
    """the transformer and the motion module with every to_q / to_k weight multiplied by `gain`: logits grow by gain^2 (to
    ~30 - 60, where trained checkpoints live; torch's default init keeps them under 3), the softmax rows become nearly one-hot
    and the running-max / rescale path of every attention kernel decides the result.  (A row maximum taken from a quarter of
    the keys passed every unit-scale test of rounds 1-2.)"""
    host_threads()
    if kind == "t2d":
        from oracle.i2v_adapter import I2VAdapterTransformer2DModel as O
        kw = dict(num_attention_heads=8, attention_head_dim=c // 8, in_channels=c, num_layers=1, cross_attention_dim=768,
                  norm_num_groups=32)
        hip_cls = pkg().I2VAdapterTransformer2DModel
    else:
        from oracle.blocks import TransformerTemporalModel as O
        kw = dict(num_attention_heads=8, attention_head_dim=c // 8, in_channels=c, norm_num_groups=32, attention_bias=False,
                  activation_fn="geglu", positional_embeddings="sinusoidal", num_positional_embeddings=32)
        hip_cls = pkg().TransformerTemporalModel
    m = hip_model_random(kw, dev, seed=c + hw + 7, cls=hip_cls)
    with torch.no_grad():
        n_scaled = 0
        for name, prm in m.named_parameters():
            if name.endswith(("to_q.weight", "to_k.weight")):
                prm.mul_(gain)
                n_scaled += 1
    assert n_scaled >= 4
    o = oracle_from_hip(m, O, kw)
    g = torch.Generator().manual_seed(c + 2)
    x = h(torch.randn(2 * frames, c, hw, hw, generator=g))
    with torch.no_grad():
        if kind == "t2d":
            ctx = h(torch.randn(2 * frames, 77, 768, generator=g))
            ref = o(x, enable_cross_frame_attn=True, num_frames=frames, encoder_hidden_states=ctx, return_dict=False)[0]
            got = m(x.half().to(dev), enable_cross_frame_attn=True, num_frames=frames, encoder_hidden_states=ctx.half().to(dev),
                    return_dict=False)[0]
        else:
            ref, got = o(x, num_frames=frames)[0], m(x.half().to(dev), num_frames=frames)[0]
    compare(got, ref, rel=3.5e-4 * gain, name=f"sharp attention {kind} C={c} gain={gain}")


@pytest.mark.parametrize("cin,cout,hw", [(320, 320, 64), (960, 320, 32), (2560, 1280, 8), (1280, 1280, 16)])
def test_full_width_resnet(dev, cin, cout, hw):
    from oracle.blocks import ResnetBlock2D as O
    host_threads()
    kw = dict(in_channels=cin, out_channels=cout, temb_channels=1280, eps=1e-5, groups=32)
    o, m = _module_pair(pkg().ResnetBlock2D, O, kw, dev, seed=cin + cout)
    g = torch.Generator().manual_seed(cin)
    n = 32 if hw <= 16 else 8
    x = h(torch.randn(n, cin, hw, hw, generator=g))
    temb = h(torch.randn(n, 1280, generator=g))
    from i2v_adapter_unofficial_amd import blocks
    with torch.no_grad():
        ref = o(x, temb)
        got = m(x.half().to(dev), temb.half().to(dev))
        compare(got, ref, rel=MODULE_REL_TOL, name=f"full-width ResnetBlock2D {cin}->{cout} {hw}x{hw}")
        # norm2's statistics from conv1's epilogue (default where the form exists) against the norm's own statistics pass
        assert blocks.GN_FROM_CONV
        blocks.GN_FROM_CONV = False
        try:
            own = m(x.half().to(dev), temb.half().to(dev))
        finally:
            blocks.GN_FROM_CONV = True
        compare(own, ref, rel=MODULE_REL_TOL, name=f"full-width ResnetBlock2D, norm2 with its own statistics pass {cin}->{cout}")
        compare(got, own, rel=MODULE_REL_TOL, name=f"ResnetBlock2D statistics from the epilogue vs own pass {cin}->{cout}")


# ------------------------------------------------------------------------------------------------ full-size properties
def _pipeline_run(hu, dev, frames, size, ip, use_graph, steps=2, seed=0):
    p = pkg()
    g = torch.Generator().manual_seed(100 + seed)
    lat = size // 8
    pe, ne = h(torch.randn(1, 77, 768, generator=g)), h(torch.randn(1, 77, 768, generator=g))
    cond = torch.randn(1, 4, lat, lat, generator=g)
    kw = {}
    if ip:
        kw["image_embeds"] = h(torch.randn(1, 1024, generator=g))
    pipe = p.I2VAdapterPipeline(unet=hu)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond, num_frames=frames,
               num_inference_steps=25, guidance_scale=7.5, frame_similarity_sample_ratio=steps / 25 + 1e-3,
               generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
               prior_noise_generator=torch.Generator().manual_seed(7), blur_sigma=1.0, use_graph=use_graph, **kw)
    return out.frames, cond


@pytest.mark.parametrize("name,frames,size,ip", [("config2", 16, 512, False), ("config3", 16, 512, True),
                                                 ("config5", 32, 768, False)])
def test_full_size_properties(dev, pair, pair_ip, name, frames, size, ip):
    """BASELINE configs 2 / 3 / 5 at their full sizes (two DDIM steps of the truncated 25-step schedule)."""
    hu = (pair_ip if ip else pair)[1]
    eager, cond = _pipeline_run(hu, dev, frames, size, ip, use_graph=False)
    lat = size // 8
    assert eager.shape == (1, frames, 4, lat, lat) and eager.dtype == torch.float32
    assert torch.isfinite(eager).all(), f"{name}: non-finite latents"
    assert torch.equal(eager[:, 0].cpu(), cond), f"{name}: frame 0 must equal the condition latents (pipe:699-700)"
    assert eager[:, 1:].std().item() > 0.1, f"{name}: degenerate latents"
    graph, _ = _pipeline_run(hu, dev, frames, size, ip, use_graph=True)
    assert torch.equal(eager, graph), f"{name}: hipGraph replay differs from eager launches"
    again, _ = _pipeline_run(hu, dev, frames, size, ip, use_graph=True)
    assert torch.equal(graph, again), f"{name}: same seeds must reproduce the trajectory bit for bit"


@pytest.mark.parametrize("ip", [False, True])
def test_cfg_shared_prefix(dev, pair, pair_ip, ip, monkeypatch):
    """The step computes the prompt-independent prefix of the UNet (conv_in, the first resnet, the self- / cross-frame
    attention stage of the first transformer) ONCE for the two CFG halves (pipeline CFG_SHARED; pipe:672-673 duplicates the
    latents and the reference computes both halves).  The arithmetic per element is the same, but the half batch can be
    dispatched to other tile forms (another summation order in a GroupNorm partial or a K loop), and a one-ulp difference
    early in an fp16 network grows to the size of the fp16 path's own error at its output -- so the binding check is the
    same as for the un-shared path: the fp32 oracle on the CFG-shaped batch, same tolerance.  Eager == replay bit for bit."""
    import importlib
    ou, hu = pair_ip if ip else pair
    inp = _inputs(seed=9)
    lat = inp["sample"][:1]
    sample = torch.cat([lat, lat])                                   # pipe:672 `torch.cat([latents] * 2)`
    added = {"image_embeds": inp["image_embeds"]} if ip else None
    added_d = {"image_embeds": inp["image_embeds"].to(dev)} if ip else None
    with torch.no_grad():
        ref = ou(sample, inp["t"], True, inp["ctx"], added_cond_kwargs=added).sample
        twice = hu(sample.to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev), added_cond_kwargs=added_d).sample
        once = hu(sample.to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev), added_cond_kwargs=added_d,
                  cross_attention_kwargs={"cfg_shared_prefix": True}).sample
    e2, scale = compare(twice, ref, abs_tol=FWD_ABS_TOL, name=f"CFG batch, prefix computed twice (ip={ip})")
    e1, _ = compare(once, ref, abs_tol=FWD_ABS_TOL, name=f"CFG batch, prefix computed once (ip={ip})")
    print(f"CFG-shaped batch vs oracle: prefix twice {e2:.3e}, once {e1:.3e}, once vs twice "
          f"{(once - twice).abs().max().item():.3e}, max|ref| {scale:.3e}")
    assert (once[0] - once[1]).abs().max().item() > 10 * FWD_ABS_TOL     # the halves differ once the prompt has entered
    # the pipeline's step: on / off, eager / replayed
    pl = importlib.import_module(pkg().I2VAdapterPipeline.__module__)
    monkeypatch.setattr(pl, "CFG_SHARED", False)
    p_twice, _ = _pipeline_run(hu, dev, 8, 256, ip, use_graph=False)
    monkeypatch.setattr(pl, "CFG_SHARED", True)
    p_once, _ = _pipeline_run(hu, dev, 8, 256, ip, use_graph=False)
    d = (p_once - p_twice).abs().max().item()
    print(f"two steps at 8 f x 256^2, prefix once vs twice: max |diff| {d:.3e} at max {p_twice.abs().max().item():.3e}")
    assert d <= 3e-3 * p_twice.abs().max().item()
    p_graph, _ = _pipeline_run(hu, dev, 8, 256, ip, use_graph=True)
    assert torch.equal(p_graph, p_once)


def test_config2_full_size_forward_vs_oracle(dev, pair):
    """BASELINE configs[1] at its FULL size: one CFG forward (2, 16, 4, 64, 64) of the SD-1.5-width model against the
    fp32 CPU oracle (~50 s of host time on the GPU box; attention through F.scaled_dot_product_attention, the op the
    reference's AttnProcessor2_0 calls, because the explicit-softmax form would materialise 2 x 17 GB of scores).
    Measured in round 2: 1.9e-3 max-abs at max|ref| 2.0 (bench.py reports the same comparison as `parity`)."""
    import torch.nn.functional as F
    from oracle import blocks as oblocks
    ou, hu = pair
    inp = _inputs(frames=16, hw=64, seed=11)
    orig = oblocks.Attention._sdpa
    oblocks.Attention._sdpa = lambda self, q, k, v: F.scaled_dot_product_attention(q, k, v, scale=self.scale)
    torch.set_num_threads(min(32, torch.get_num_threads() * 2))
    try:
        with torch.no_grad():
            ref = ou(inp["sample"], inp["t"], True, inp["ctx"]).sample
    finally:
        oblocks.Attention._sdpa = orig
        host_threads()
    with torch.no_grad():
        got = hu(inp["sample"].to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev)).sample
    assert got.shape == (2, 16, 4, 64, 64)
    err, scale = compare(got, ref, abs_tol=4.0e-3, name="config 2 full-size UNet forward (16f x 512^2)")
    print(f"config 2 full size: HIP err {err:.3e} at max|ref| {scale:.3e}")


# ------------------------------------------------------------------------------------------------ training step, SD-1.5 width
def test_training_step_full_width_vs_autograd(dev, pair):
    """SURVEY 8 f4 at the width `bench.py --train` times: `UNetAdapterTrainer` on the SD-1.5-width UNet
    (channels 320 / 640 / 1280 / 1280, head_dim 40 / 80 / 160, 16 spatial transformers with the adapter) at a small clip
    (2 f x 256 x 256: latents 32 x 32) against torch autograd over the fp32 oracle: the loss of
    train_image_to_video.py:848-856 (MSE without the first frame) and the gradient of all 16 x 3 trainable adapter tensors
    (unet:979-1026) AND, as under `--update_motion_modules` (train_image_to_video.py:452, 669; unet:984-999), of all 21 x 26
    motion-module tensors.  Tolerances as on the reduced UNet (tests/test_training_gpu.py): loss 5e-4 rel (fp32 head: measured 6e-6), gradients 2e-2 of
    their largest entry (measured: <= 4.2e-3, the worst on a 1e-5-sized gradient; asserted 8e-3)."""
    from i2v_adapter_unofficial_amd.training import UNetAdapterTrainer
    ou, hu = pair
    frames, lat = 2, 32
    g = torch.Generator().manual_seed(606)
    sample = h(torch.randn(1, frames, 4, lat, lat, generator=g))
    ctx = h(torch.randn(1, 77, 768, generator=g))
    target = h(torch.randn(1, frames, 4, lat, lat, generator=g))
    t = torch.tensor([481])
    for prm in ou.parameters():
        prm.requires_grad_(False)
    train = {n: prm for n, prm in ou.named_parameters()
             if ".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n or ".motion_modules." in n}
    assert len(train) == 16 * 3 + 21 * 26
    for prm in train.values():
        prm.requires_grad_(True)
        prm.grad = None
    try:
        pred = ou(sample, t, True, ctx).sample
        mask = torch.ones_like(pred)
        mask[:, 0] = 0
        loss = ((pred.float() - target) ** 2 * mask).sum() / mask.sum()
        loss.backward()
        tr = UNetAdapterTrainer(hu, update_motion_modules=True)
        y = tr.forward(sample.half().to(dev), t.to(dev), ctx.half().to(dev))
        got_pred = y[..., :4].float().cpu().permute(0, 3, 1, 2).reshape(pred.shape)
        compare(got_pred, pred.detach(), abs_tol=FWD_ABS_TOL, name="training forward, SD-1.5 width")
        got_loss, grads = tr.backward(target.to(dev), loss_scale=2.0 ** 12)
        assert abs(got_loss.item() - loss.item()) <= 5e-4 * abs(loss.item()), (got_loss.item(), loss.item())   # measured <= 7e-6
        assert set(grads) == set(train)
        worst = {".i2v_adapter.": 0.0, ".motion_modules.": 0.0}
        for name, prm in train.items():
            part = ".i2v_adapter." if ".i2v_adapter." in name else ".motion_modules."
            # (motion modules: measured <= 9.6e-3, the worst on the 5e-6-sized to_q / to_k gradients of the two-frame temporal
            # attention at 8 x 8 -- fp16 activation gradients under the 2^12 loss scale; everything else <= 5e-3)
            err, scale = compare(grads[name], prm.grad, rel=8e-3 if part == ".i2v_adapter." else 1.5e-2,
                                 name=f"SD-1.5-width training step: d loss / d {name}")
            worst[part] = max(worst[part], err / scale)
        print(f"full-width training step: loss {got_loss.item():.6f} vs {loss.item():.6f}, worst gradient error of max: {worst}")
    finally:
        for prm in train.values():
            prm.requires_grad_(False)
            prm.grad = None


# ------------------------------------------------------------------------------------------------ config 4: B samples per call
def _batch_inputs(nb, frames, lat, seed=40):
    g = torch.Generator().manual_seed(seed)
    return dict(pe=h(torch.randn(nb, 77, 768, generator=g)), ne=h(torch.randn(nb, 77, 768, generator=g)),
                ie=h(torch.randn(nb, 1024, generator=g)), cond=torch.randn(nb, 4, lat, lat, generator=g))


def _pipe_call(pipe, inp, idx, frames, use_graph=True, steps=2):
    """samples `idx` of `inp` in ONE pipeline call; every random draw of sample i comes from generators seeded by i alone
    (lists of generators), so a sample's noise does not depend on which other samples share its call"""
    sel = lambda t: t[idx]
    gens = lambda base: [torch.Generator().manual_seed(base + 10 * i) for i in idx]
    return pipe(prompt_embeds=sel(inp["pe"]), negative_prompt_embeds=sel(inp["ne"]), image_embeds=sel(inp["ie"]),
                condition_image_latents=sel(inp["cond"]), num_frames=frames, num_inference_steps=25, guidance_scale=7.5,
                frame_similarity_sample_ratio=steps / 25 + 1e-3, generator=gens(1), prior_mask_generator=gens(2),
                prior_noise_generator=gens(3), blur_sigma=1.0, use_graph=use_graph).frames


@pytest.mark.parametrize("nb", [2, 4])
def test_config4_samples_per_replay_match_single_sample_runs(dev, pair_ip, nb):
    """BASELINE configs[3] runs several (image, prompt) pairs per graph replay (`bench.py --pairs 64 --batch B`): CFG batch
    2 B with the context ordered [neg_0 .. neg_B-1, pos_0 .. pos_B-1] (pipe:613-622), per-sample frame-0 K / V through
    kv_group (i2v:484-485), per-sample time / context rows (unet:1344, 1355).  SD-1.5 width, IP-Adapter on, hipGraph on:
    every sample of a B-sample call must equal the same sample run alone (batch semantics pipe:582-587: no cross-sample
    op).  Tile heights, split-K and the GroupNorm slab form are chosen by the batch, so the summation order may differ:
    the bound is fp16 rounding noise, not bit equality (measured in round 3: see profiles/r3_parity_errors.jsonl)."""
    hu = pair_ip[1]
    frames, lat = 8, 16
    inp = _batch_inputs(nb, frames, lat)
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    idx = list(range(nb))
    both = _pipe_call(pipe, inp, idx, frames)
    assert both.shape == (nb, frames, 4, lat, lat) and torch.isfinite(both).all()
    assert torch.equal(both[:, 0].cpu(), inp["cond"])
    again = _pipe_call(pipe, inp, idx, frames)
    assert torch.equal(both, again), "a cached graph replay of the same samples must be bit-identical"
    for i in idx:
        alone = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp, [i], frames)
        # measured 8.3e-4 .. 9.8e-4 of max (two trajectories of independent fp16 rounding noise), bound 2x
        compare(both[i: i + 1], alone, rel=2e-3, name=f"config 4: sample {i} of {nb} per call vs alone")
        others = [j for j in idx if j != i]
        assert (both[others[0]] - both[i]).abs().max().item() > 0.1       # the samples really differ
    # eager launches of the same call: bit-identical to the graph
    eager = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp, idx, frames, use_graph=False)
    assert torch.equal(eager, both)


def test_config4_full_size_two_samples_per_replay(dev, pair_ip):
    """BASELINE configs[3] at its FULL clip size in the driver-run suite: two (image, prompt) pairs per
    graph replay at 16 f x 512 x 512 with the IP-Adapter on (CFG batch 4 x 16 frames, 262144-row GEMMs: other tile forms,
    split-K plans and GroupNorm forms than the single-sample step).  Each sample of the two-sample call must equal the
    same sample run alone up to fp16 rounding noise (pipe:582-587, 613-622: no cross-sample op), frame 0 must be the
    condition latents exactly (pipe:699-700), the replay must be reproducible and equal to eager launches."""
    hu = pair_ip[1]
    frames, lat = 16, 64
    inp = _batch_inputs(2, frames, lat, seed=44)
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    both = _pipe_call(pipe, inp, [0, 1], frames)
    assert both.shape == (2, frames, 4, lat, lat) and torch.isfinite(both).all()
    assert torch.equal(both[:, 0].cpu(), inp["cond"])
    assert torch.equal(both, _pipe_call(pipe, inp, [0, 1], frames)), "a cached replay of the same samples must be bit-identical"
    for i in (0, 1):
        alone = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp, [i], frames)
        err, scale = compare(both[i: i + 1], alone, rel=2e-3, name=f"config 4 at 16f x 512^2: sample {i} of 2 per replay vs alone")
        print(f"config 4 full size: sample {i} in a 2-sample replay vs alone: {err:.3e} at max {scale:.3e}")
    assert (both[0] - both[1]).abs().max().item() > 0.1                    # the samples really differ
    del pipe
    eager = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp, [0, 1], frames, use_graph=False)
    assert torch.equal(eager, both)


def test_graph_reuse_with_a_new_prompt_and_image_refreshes_the_packed_context(dev, pair_ip):
    """the same with the IP-Adapter installed: the image tokens' packed K / V are a second buffer the graph reads"""
    _graph_reuse_case(pair_ip[1])


def test_graph_reuse_with_a_new_prompt_refreshes_the_packed_context(dev, pair):
    """A captured step is replayed for the next sample of the same shape with its inputs copied into the graph's buffers
    (pipeline `_run_steps`); the fused text cross-attention reads the prompt's K / V as packed MFMA fragments, which must be
    rewritten IN PLACE with the new prompt's (no Python runs between replays).  Sample B through a pipeline that captured its
    graph on sample A must equal sample B through a fresh pipeline bit for bit, and differ from sample A.  16 frames at
    128 x 128: all three fused sub-block kernels are on the path (C = 320 at the top level, rows a multiple of 128)."""
    _graph_reuse_case(pair[1])


def _graph_reuse_case(hu):
    frames, lat = 16, 16
    inp = _batch_inputs(2, frames, lat, seed=46)
    assert pkg().kernels.cross_attn_fused_supported(2 * frames * lat * lat, 320, 8, 40, 77, frames * lat * lat)
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    a = _pipe_call(pipe, inp, [0], frames)
    b_reused = _pipe_call(pipe, inp, [1], frames)                   # cache hit: same shapes, new prompt / image / noise
    b_fresh = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp, [1], frames)
    assert torch.equal(b_reused, b_fresh), (b_reused - b_fresh).abs().max().item()
    assert (a - b_fresh).abs().max().item() > 0.1
    # and the same prompt changed in isolation (same latents, same noise): only the context differs
    inp2 = dict(inp)
    inp2["pe"] = inp["pe"].flip(0).contiguous()
    inp2["ie"] = inp["ie"].flip(0).contiguous()
    c_reused = _pipe_call(pipe, inp2, [1], frames)
    c_fresh = _pipe_call(pkg().I2VAdapterPipeline(unet=hu), inp2, [1], frames)
    assert torch.equal(c_reused, c_fresh) and not torch.equal(c_fresh, b_fresh)


def test_config4_two_sample_cfg_forward_vs_oracle(dev, pair_ip):
    """one CFG forward of TWO samples (batch 4 = [neg_0, neg_1, pos_0, pos_1]) with IP tokens against the oracle: the
    per-sample structure (frame-0 K / V per clip, context and time rows per sample) at SD-1.5 width."""
    ou, hu = pair_ip
    g = torch.Generator().manual_seed(404)
    frames, lat = 8, 16
    x2 = h(torch.randn(2, frames, 4, lat, lat, generator=g))
    sample = torch.cat([x2, x2])                                             # pipe:672
    ctx = h(torch.randn(4, 77, 768, generator=g))
    ie = torch.cat([torch.zeros(2, 1024), h(torch.randn(2, 1024, generator=g))])    # pipe:343, 621-622
    t = torch.tensor([601, 601, 601, 601])
    with torch.no_grad():
        ref = ou(sample, t, True, ctx, added_cond_kwargs={"image_embeds": ie}).sample
        got = hu(sample.to(dev), t.to(dev), True, ctx.to(dev), added_cond_kwargs={"image_embeds": ie.to(dev)}).sample
    err, scale = compare(got, ref, abs_tol=FWD_ABS_TOL, name="config 4: two-sample CFG forward + IP vs oracle")
    print(f"two-sample CFG forward: err {err:.3e} at max|ref| {scale:.3e}")
    # the two samples must not leak into each other: sample 0's rows equal a forward of sample 0 alone
    sel = [0, 2]
    with torch.no_grad():
        one = hu(sample[sel].to(dev), t[sel].to(dev), True, ctx[sel].to(dev),
                 added_cond_kwargs={"image_embeds": ie[sel].to(dev)}).sample
    # two fp16 paths with different tile heights / split-K / GroupNorm forms: independent rounding noise of the size of
    # the error against the oracle (measured 2.0e-3 at max|ref| 1.33)
    compare(got[sel], one, abs_tol=FWD_ABS_TOL, name="config 4: sample 0 inside a two-sample forward vs alone")


def test_rccl_world_size_1_weight_broadcast_full_model():
    """the RCCL branch of bench.py on ONE GPU: `init_process_group("nccl", world_size=1, device_id=cuda:0)`, the flat
    2.76 GB state-dict broadcast of the SD-1.5-width model on the device, the MAX all-reduce of the elapsed time and the
    barrier.  (The 8-GPU run is the driver's; this exercises RCCL initialisation and the device flatten path.)  In a child
    process: a process group must not leak into the test session."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, socket, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from tests.parity import SD15, hip_model_random
from i2v_adapter_unofficial_amd.sharding import broadcast_model_weights
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = hip_model_random(SD15, dev, seed=7)
before = {k: v.clone() for k, v in list(m.state_dict().items())[::97]}
nbytes = broadcast_model_weights(m, src=0)
assert 2.5e9 < nbytes < 3.1e9, nbytes
for k, v in before.items():
    assert torch.equal(m.state_dict()[k], v), k
tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
assert float(tt.item()) == 1.25 and dist.get_world_size() == 1
dist.destroy_process_group()
print("RCCL_OK", nbytes)
""" % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ------------------------------------------------------------------------------------------------ full-size oracle FIXTURES (r6)
# tests/golden/oracle_full_*.safetensors (oracle/make_full_fixtures.py, made in the build container): the fp32 CPU oracle's outputs
# at BASELINE's full sizes on weights re-created from a seed by the CPU generator, so that every configuration is compared with the
# oracle here in seconds (config 5's oracle forward alone is 15 minutes of 8 cores) -- VERDICT r5 items 2 / weak #3.
# Tolerances (measured r6, both residual-stream modes, profiles/r6_parity_fixtures.jsonl): forward max-abs <= 4e-3 (the bench gate;
# measured 1.9 - 2.2e-3 default, 1.6 - 2.0e-3 precise at max|ref| 1.5 - 1.8) and rms error <= 1.8e-3 of the output's rms (measured
# 1.17 - 1.20e-3 default, 1.01 - 1.03e-3 precise: this weight draw -- jittered norm affines, rms 0.32 -- sits above the bench's
# 0.89e-3); latents <= 3.3e-3 of max|latent| at every stored step (REL_TOL_TRAJECTORY).
FWD_RMS_REL_TOL = 1.8e-3


def _golden(name):
    import os
    from safetensors.torch import load_file
    from tests.parity import GOLDEN_DIR
    path = os.path.join(GOLDEN_DIR, name)
    if not os.path.exists(path):
        pytest.fail(f"{path} is missing: python -m oracle.make_full_fixtures (build container)")
    return load_file(path)


def _seeded_hip_model(dev, ip):
    """the HIP model on the fixture's weights (re-created on the CPU from the seed, pinned by the fixture's checksum)"""
    from tests.parity import (FULL_SEED, FULL_SEED_IP, hip_unet_from_oracle, oracle_full_width_cpu_seeded, sd15_ip_state_dict,
                              weights_checksum)
    host_threads()
    ou = oracle_full_width_cpu_seeded(FULL_SEED_IP if ip else FULL_SEED, ip=ip)
    chk = weights_checksum(ou)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=sd15_ip_state_dict(ou) if ip else None)
    del ou
    return hu, chk


@pytest.fixture(scope="module")
def seeded(dev):
    return _seeded_hip_model(dev, ip=False)


@pytest.fixture(scope="module")
def seeded_ip(dev):
    return _seeded_hip_model(dev, ip=True)


def _both_modes(fn):
    """fn() under the default and under the precise residual stream (blocks.set_precise_stream): {'default': ..., 'precise': ...}"""
    from i2v_adapter_unofficial_amd import blocks
    out = {}
    for mode in (False, True):
        prev = blocks.set_precise_stream(mode)
        try:
            out["precise" if mode else "default"] = fn()
        finally:
            blocks.set_precise_stream(prev)
    return out


@pytest.mark.parametrize("name,frames,h_lat,ip", [("config2", 16, 64, False), ("config3", 16, 64, True), ("config5", 32, 96, False)])
def test_full_size_forward_vs_oracle_fixture(dev, request, name, frames, h_lat, ip):
    """BASELINE configs[1] / [2] / [4] at FULL size: the CFG forward (unet:1289-1451; IP-Adapter unet:1230-1287, 1346-1355; 32 frames =
    the positional table's limit, unet:725) against the committed fp32 oracle output, in both residual-stream modes."""
    from tests.parity import full_forward_inputs
    hu, chk = request.getfixturevalue("seeded_ip" if ip else "seeded")
    fx = _golden(f"oracle_full_{name}.safetensors")
    assert torch.allclose(fx["weights_checksum"], chk, rtol=1e-12, atol=0), \
        "the weights re-created from the seed differ from the ones the fixture was made with (another torch build?)"
    ref = fx["noise_pred"]
    inp = full_forward_inputs(frames, h_lat, ip)
    added = {"image_embeds": inp["image_embeds"].to(dev)} if ip else None
    assert ref.shape == (2, frames, 4, h_lat, h_lat)

    def fwd():
        with torch.no_grad():
            return hu(inp["sample"].to(dev), inp["t"].to(dev), True, inp["ctx"].to(dev), added_cond_kwargs=added).sample.float().cpu()
    got = _both_modes(fwd)
    rms_ref = ref.pow(2).mean().sqrt().item()
    res = {}
    for mode, y in got.items():
        err, scale = compare(y, ref, abs_tol=4.0e-3, name=f"{name} full-size forward vs oracle fixture ({mode} stream)")
        rms = (y - ref).pow(2).mean().sqrt().item()
        log_error(f"{name} full-size forward vs oracle fixture ({mode} stream), rms", rms, rms_ref, FWD_RMS_REL_TOL * rms_ref)
        res[mode] = (err, rms)
        assert rms <= FWD_RMS_REL_TOL * rms_ref, f"{name} ({mode}): rms error {rms:.3e} at rms {rms_ref:.3e}"
    print(f"{name} full size vs oracle fixture (max|ref| {ref.abs().max().item():.3f}, rms {rms_ref:.3f}): "
          + "; ".join(f"{m} max {e:.3e} rms {r:.3e}" for m, (e, r) in res.items()))
    assert res["precise"][1] < res["default"][1]


def test_config2_latent_trajectory_vs_oracle_fixture(dev, seeded):
    """north_star words its tolerance on LATENTS: the 25-step CFG DDIM trajectory of config 2 (pipe:629-700; hipGraph-captured step)
    against the fp32 oracle pipeline's latents after steps 1, 2, 3, 5, 10, 15, 20, 25 -- the error's growth along the trajectory,
    printed and bounded per step, in both residual-stream modes.  (With random-init weights the CFG-amplified latents grow along the
    trajectory -- max|latent| is printed -- so the bound is relative to max|latent| of the step.)"""
    from tests.parity import REL_TOL_TRAJECTORY, trajectory_inputs
    hu, chk = seeded
    fx = _golden("oracle_latents_config2_25steps.safetensors")
    assert torch.allclose(fx["weights_checksum"], chk, rtol=1e-12, atol=0)
    steps = sorted(int(k[len("latents_step"):]) for k in fx if k.startswith("latents_step"))
    kw, gens = trajectory_inputs(16, 64)
    pipe = pkg().I2VAdapterPipeline(unet=hu)

    def run():
        snaps = {}

        def cb(i, t, latents):
            if i + 1 in steps:
                snaps[i + 1] = latents.detach().float().cpu().clone()
        final = pipe(**kw, **gens(), callback=cb).frames.float().cpu()
        return snaps, final
    got = _both_modes(run)
    worst = {}
    for mode, (snaps, final) in got.items():
        line = []
        for s in steps:
            ref = fx[f"latents_step{s:02d}"]
            d = (snaps[s] - ref)
            rel = d.abs().max().item() / ref.abs().max().item()
            line.append(f"{s}: {d.abs().max().item():.2e}/{ref.abs().max().item():.1f}={rel:.2e}")
            log_error(f"config 2 latents after step {s} vs oracle fixture ({mode} stream)", d.abs().max().item(), ref.abs().max().item(),
                      REL_TOL_TRAJECTORY * ref.abs().max().item(), rms_err=d.pow(2).mean().sqrt().item(), rms_ref=ref.pow(2).mean().sqrt().item())
            worst[mode] = max(worst.get(mode, 0.0), rel)
            assert rel <= REL_TOL_TRAJECTORY, f"latents after step {s} ({mode}): {rel:.3e} of max|latent|"
        print(f"config 2 latent trajectory vs oracle ({mode} stream), max-abs err / max|latent| per step: " + "  ".join(line))
        assert torch.equal(final[:, 0], kw["condition_image_latents"]), "frame 0 must equal the condition latents (pipe:699-700)"
        compare(final, fx["final"], rel=REL_TOL_TRAJECTORY, name=f"config 2 final latents vs oracle fixture ({mode} stream)")
    # the hipGraph route (no callback) reproduces the callback route's final latents bit for bit
    from i2v_adapter_unofficial_amd import blocks
    graph_final = pipe(**kw, **gens()).frames.float().cpu()
    assert torch.equal(graph_final, got["precise" if blocks.precise_stream() else "default"][1])
