"""GPU parity of the HIP module / model / pipeline API against the CPU oracle.

The module-level cases use the shapes of the reference's own tests (test/test_i2v_adapter.py,
test/test_unet_motion_cross_frame_attn.py), which only assert output shapes; here the same calls are also
compared numerically with the oracle (tolerance: tests/parity.py).  The model-level cases use the reduced UNet
(SD-1.5 topology, narrow channels) of SURVEY 8c.
"""
import pytest
import torch

from tests.parity import (REL_TOL_MODULE, REL_TOL_TRAJECTORY, REL_TOL_UNET, SMALL_UNET, compare, hip_unet_from_oracle,
                          log_error, oracle_small_unet, randomize_adapter_out_, round_fp16_, small_ip_state_dict,
                          small_unet_inputs)

pytestmark = pytest.mark.gpu


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def h(t):
    return t.half().float()


def _pair(oracle_cls, hip_cls, dev, seed, *args, **kwargs):
    torch.manual_seed(seed)
    o = round_fp16_(oracle_cls(*args, **kwargs)).eval()
    m = hip_cls(*args, **kwargs)
    m.load_state_dict(o.state_dict())
    return o, m.to(device=dev, dtype=torch.float16).eval()


@pytest.mark.parametrize("cross_frame", [True, False])
def test_transformer_block(dev, cross_frame):
    """reference test/test_i2v_adapter.py:73-124 (hidden 256, 8 heads, ctx 77 x 512, activation 'gelu')."""
    from oracle.i2v_adapter import I2VAdapterTransformerBlock as O
    o, m = _pair(O, pkg().I2VAdapterTransformerBlock, dev, 1, 256, 8, 32, dropout=0.0, cross_attention_dim=512,
                 activation_fn="gelu")
    g = torch.Generator().manual_seed(2)
    frames, batch = 8, 4
    bf = frames * batch if cross_frame else batch
    x = h(torch.randn(bf, 64, 256, generator=g))
    ctx = h(torch.randn(bf, 77, 512, generator=g))
    kw = dict(enable_cross_frame_attn=cross_frame, num_frames=frames if cross_frame else None)
    with torch.no_grad():
        ref = o(x, encoder_hidden_states=ctx, **kw)
        got = m(x.half().to(dev), encoder_hidden_states=ctx.half().to(dev), attention_mask=None,
                encoder_attention_mask=None, **kw)
    assert got.shape == (bf, 64, 256)
    compare(got, ref, name="I2VAdapterTransformerBlock")


def test_transformer_block_errors(dev):
    m = pkg().I2VAdapterTransformerBlock(64, 8, 8, cross_attention_dim=32).to(dev).half()
    x = torch.zeros(6, 16, 64, dtype=torch.float16, device=dev)
    ctx = torch.zeros(6, 7, 32, dtype=torch.float16, device=dev)
    with pytest.raises(ValueError, match="num_frames"):
        m(x, enable_cross_frame_attn=True, encoder_hidden_states=ctx)
    with pytest.raises(ValueError, match="divisible"):
        m(x, enable_cross_frame_attn=True, num_frames=4, encoder_hidden_states=ctx)
    with pytest.raises(pkg().HipLibraryError):
        m(x.cpu(), encoder_hidden_states=ctx.cpu())


@pytest.mark.parametrize("cross_frame", [True, False])
def test_transformer_2d_model(dev, cross_frame):
    """reference test/test_i2v_adapter.py:11-71 (8 heads, C = 512, ctx dim 1024, 16 x 16, F = 8)."""
    from oracle.i2v_adapter import I2VAdapterTransformer2DModel as O
    o, m = _pair(O, pkg().I2VAdapterTransformer2DModel, dev, 3, 8, 64, in_channels=512, out_channels=512,
                 num_layers=1, cross_attention_dim=1024, norm_num_groups=32)
    g = torch.Generator().manual_seed(4)
    frames = 8
    bf = 2 * frames if cross_frame else 6
    x = h(torch.randn(bf, 512, 16, 16, generator=g))
    ctx = h(torch.randn(bf, 77, 1024, generator=g))
    kw = dict(enable_cross_frame_attn=cross_frame, num_frames=frames if cross_frame else None, return_dict=False)
    with torch.no_grad():
        ref = o(x, encoder_hidden_states=ctx, **kw)[0]
        got = m(x.half().to(dev), encoder_hidden_states=ctx.half().to(dev), attention_mask=None,
                encoder_attention_mask=None, **kw)[0]
    assert got.shape == (bf, 512, 16, 16)
    compare(got, ref, name="I2VAdapterTransformer2DModel")


@pytest.mark.parametrize("cross_frame", [True, False])
def test_down_block(dev, cross_frame):
    """reference test/test_unet_motion_cross_frame_attn.py:18-92 (64 -> 128 ch, temb 512, ctx 768, 16 x 16, F = 8)."""
    from oracle.unet_motion_cross_frame_attn import CrossFrameAttnDownBlockMotion as O
    o, m = _pair(O, pkg().CrossFrameAttnDownBlockMotion, dev, 5, in_channels=64, out_channels=128,
                 temb_channels=512, cross_attention_dim=768, num_layers=2, num_attention_heads=8)
    g = torch.Generator().manual_seed(6)
    frames, bf = 8, 16
    x = h(torch.randn(bf, 64, 16, 16, generator=g))
    temb = h(torch.randn(bf, 512, generator=g))
    ctx = h(torch.randn(bf, 77, 768, generator=g))
    with torch.no_grad():
        ref, ref_states = o(hidden_states=x, temb=temb, enable_cross_frame_attn=cross_frame,
                            encoder_hidden_states=ctx, num_frames=frames)
        got, got_states = m(hidden_states=x.half().to(dev), temb=temb.half().to(dev),
                            enable_cross_frame_attn=cross_frame, encoder_hidden_states=ctx.half().to(dev),
                            num_frames=frames)
    assert got.shape == (bf, 128, 8, 8) and len(got_states) == 3
    compare(got, ref, name="CrossFrameAttnDownBlockMotion")
    for i, (a, b) in enumerate(zip(got_states, ref_states)):
        compare(a, b, name=f"output_states[{i}]")


def test_motion_module_and_resnet(dev):
    from oracle.blocks import ResnetBlock2D as OR, TransformerTemporalModel as OT
    g = torch.Generator().manual_seed(8)
    o, m = _pair(OT, pkg().TransformerTemporalModel, dev, 7, num_attention_heads=8, in_channels=320,
                 norm_num_groups=32, attention_bias=False, activation_fn="geglu", positional_embeddings="sinusoidal",
                 num_positional_embeddings=32, attention_head_dim=40)
    x = h(torch.randn(2 * 16, 320, 8, 8, generator=g))
    with torch.no_grad():
        compare(m(x.half().to(dev), num_frames=16)[0], o(x, num_frames=16)[0], name="TransformerTemporalModel F=16")
        compare(m(x[:24].half().to(dev), num_frames=12)[0], o(x[:24], num_frames=12)[0], name="TransformerTemporalModel F=12")
    o, m = _pair(OR, pkg().ResnetBlock2D, dev, 9, 64, 96, temb_channels=128, eps=1e-5, groups=32)
    x = h(torch.randn(4, 64, 12, 12, generator=g))
    temb = h(torch.randn(4, 128, generator=g))
    with torch.no_grad():
        compare(m(x.half().to(dev), temb.half().to(dev)), o(x, temb), name="ResnetBlock2D")


@pytest.mark.parametrize("cross_frame,ip", [(True, False), (False, False), (True, True)])
def test_small_unet_forward(dev, cross_frame, ip):
    ou = oracle_small_unet(ip=False)
    ipsd = small_ip_state_dict(ou) if ip else None
    if ip:
        ou._load_ip_adapter_weights(ipsd)
        round_fp16_(ou)
        ipsd = {k: {kk: vv.half().float() for kk, vv in v.items()} for k, v in ipsd.items()}
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=ipsd)
    inp = small_unet_inputs()
    added = {"image_embeds": inp["image_embeds"]} if ip else None
    added_d = {"image_embeds": inp["image_embeds"].to(dev)} if ip else None
    with torch.no_grad():
        ref = ou(inp["sample"], inp["timestep"], cross_frame, inp["ctx"], added_cond_kwargs=added).sample
        got = hu(inp["sample"].to(dev), inp["timestep"].to(dev), cross_frame, inp["ctx"].to(dev),
                 added_cond_kwargs=added_d).sample
    assert got.shape == ref.shape == (2, 4, 4, 16, 16) and got.dtype == torch.float32
    err, scale = compare(got, ref, rel=REL_TOL_UNET, name="UNetMotionCrossFrameAttnModel")
    print(f"small UNet cross_frame={cross_frame} ip={ip}: max abs err {err:.3e} (max|ref| {scale:.3e})")
    if ip:
        with pytest.raises(ValueError, match="image_embeds"):
            hu(inp["sample"].to(dev), inp["timestep"].to(dev), cross_frame, inp["ctx"].to(dev))


@pytest.mark.parametrize("hh,ww", [(12, 20), (15, 10), (9, 13), (4, 4)])
def test_small_unet_forward_upsample_size(dev, hh, ww):
    """latent sizes that are not multiples of 8 (unet:1304-1311, 1414-1415 `forward_upsample_size`): the stride-2 down-samplers
    round up (12 -> 6 -> 3 -> 2), and every up-sampler but the last block's interpolates to the size of the skip tensor it meets --
    2x or 2x - 1 -- before its convolution (diffusers Upsample2D `F.interpolate(size=output_size, mode="nearest")`).  The oracle
    takes the same branch; ragged token counts (225 pixels per image) go through the generic kernel forms."""
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    g = torch.Generator().manual_seed(hh * 31 + ww)
    hlf = lambda t: t.half().float()
    sample, ctx = hlf(torch.randn(2, 4, 4, hh, ww, generator=g)), hlf(torch.randn(2, 7, 64, generator=g))
    t = torch.tensor([10, 500])
    with torch.no_grad():
        ref = ou(sample, t, True, ctx).sample
        got = hu(sample.to(dev), t.to(dev), True, ctx.to(dev)).sample
    assert got.shape == ref.shape == (2, 4, 4, hh, ww)
    err, scale = compare(got, ref, rel=REL_TOL_UNET, name=f"UNet forward at {hh} x {ww} latents (forward_upsample_size)")
    print(f"small UNet at {hh} x {ww}: max abs err {err:.3e} (max|ref| {scale:.3e})")


@pytest.mark.parametrize("gain", [2.0, 3.0, 4.0])
def test_small_unet_forward_sharp_attention(dev, gain):
    """the whole UNet with every attention layer's to_q / to_k (spatial, cross-frame adapter, text, IP, temporal) multiplied by
    `gain`: logits x gain^2, nearly one-hot attention rows everywhere -- the regime of trained checkpoints, which torch's default
    init never reaches (the running-max defect of rounds 1-2 was invisible at unit scale)."""
    ou = oracle_small_unet(ip=False)
    ipsd = small_ip_state_dict(ou)
    ou._load_ip_adapter_weights(ipsd)
    n_scaled = 0
    with torch.no_grad():
        for name, prm in ou.named_parameters():
            if name.endswith(("to_q.weight", "to_k.weight")):
                prm.mul_(gain)
                n_scaled += 1
    assert n_scaled > 40
    round_fp16_(ou)
    ipsd = {k: {kk: vv.half().float() for kk, vv in v.items()} for k, v in ipsd.items()}
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=ipsd)
    inp = small_unet_inputs()
    from oracle.fp16_emulation import emulate_reference_fp16
    added = {"image_embeds": inp["image_embeds"]}
    with torch.no_grad():
        ref = ou(inp["sample"], inp["timestep"], True, inp["ctx"], added_cond_kwargs=added).sample
        with emulate_reference_fp16():
            emu = ou(inp["sample"], inp["timestep"], True, inp["ctx"], added_cond_kwargs=added).sample
        got = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev),
                 added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
    # nearly one-hot rows make the result sensitive to WHICH of two close keys wins: the yardstick is what fp16 rounding of the
    # same op graph costs (oracle/fp16_emulation.py), not the unit-scale tolerance
    scale = ref.abs().max().item()
    err = (got.float().cpu() - ref).abs().max().item()
    err_emu = (emu - ref).abs().max().item()
    print(f"small UNet, attention weights x{gain:g}: HIP err {err:.3e}, fp16-emulated graph err {err_emu:.3e} (max|ref| {scale:.3e})")
    log_error(f"small UNet, attention weights x{gain:g}", err, scale, None)
    log_error(f"small UNet, attention weights x{gain:g}, fp16-emulated graph", err_emu, scale, None)
    assert torch.isfinite(got).all()
    # measured on MI355X: 1.0 - 1.6 x the emulated graph's error (the pre-scaled Q adds |logit| 2^-12 to each logit, which the
    # fused SDPA of the emulation does not).  At x4 the rows are one-hot and the MAXIMUM over the outputs is decided by which of
    # two near-tied keys wins somewhere in 37 attention layers -- chaotic in the rounding pattern: 2.2e-2 in round 3, 2.7e-2 in
    # round 4 (polynomial GELU, GroupNorm partials merged per group) against 1.3e-2 for the emulated graph both times.  That
    # case is bounded at 3x the emulated error, and by its RMS (not dominated by single flips) at 2x.
    factor = 3.0 if gain >= 4.0 else 2.0
    assert err <= max(factor * err_emu, REL_TOL_UNET * scale), (err, err_emu, scale)
    rms = (got.float().cpu() - ref).pow(2).mean().sqrt().item()
    rms_emu = (emu - ref).pow(2).mean().sqrt().item()
    log_error(f"small UNet, attention weights x{gain:g}, rms", rms, scale, None)
    log_error(f"small UNet, attention weights x{gain:g}, rms of the fp16-emulated graph", rms_emu, scale, None)
    assert rms <= 2.0 * rms_emu + 1e-4 * scale, (rms, rms_emu, scale)


def test_projected_context_is_bit_exact(dev):
    """the per-sample K / V^T cache (project_context) gives the bits of the per-call projection, with and without the
    IP branch, and refreshing it in place for the next sample rewrites the same buffers."""
    ou = oracle_small_unet(ip=False)
    ipsd = small_ip_state_dict(ou)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=ipsd)
    inp = small_unet_inputs()
    ctx = inp["ctx"].to(dev).half()
    ctx_ip = hu._project_image_embeds({"image_embeds": inp["image_embeds"].to(dev)})
    with torch.no_grad():
        direct = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, ctx,
                    added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
        pc = hu.project_context(ctx, ctx_ip)
        assert len(pc.kv) == 16 and all(v[2] is not None for v in pc.kv.values())
        cached = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, pc,
                    added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
        assert torch.equal(direct, cached)
        ptrs = {a: tuple(t.data_ptr() for t in v) for a, v in pc.kv.items()}
        ctx2 = (ctx.float() * 0.5).half()
        hu.project_context(ctx2, ctx_ip, out=pc)
        assert ptrs == {a: tuple(t.data_ptr() for t in v) for a, v in pc.kv.items()}
        direct2 = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, ctx2,
                     added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
        cached2 = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, pc,
                     added_cond_kwargs={"image_embeds": inp["image_embeds"].to(dev)}).sample
        assert torch.equal(direct2, cached2) and not torch.equal(direct, direct2)


def test_adapter_contributes(dev):
    """with a non-zero adapter to_out the cross-frame branch must change the output (K1 is exercised)."""
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    inp = small_unet_inputs()
    with torch.no_grad():
        a = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
        b = hu(inp["sample"].to(dev), inp["timestep"].to(dev), False, inp["ctx"].to(dev)).sample
    assert (a - b).abs().max().item() > 1e-3


def test_from_unet2d_and_adapter_roundtrip(dev):
    from oracle.unet_motion_cross_frame_attn import (UNet2DConditionModel as OU2, UNetMotionCrossFrameAttnModel as OM)
    from oracle.blocks import MotionAdapter as OMA
    from oracle.i2v_adapter import I2VAdapterModule as OIA
    p = pkg()
    torch.manual_seed(21)
    kw = dict(block_out_channels=(32, 64, 128, 128), attention_head_dim=4, norm_num_groups=8, cross_attention_dim=64)
    ou2 = round_fp16_(OU2(**kw))
    oma = round_fp16_(OMA(block_out_channels=(32, 64, 128, 128), motion_num_attention_heads=4, motion_norm_num_groups=8))
    oia = round_fp16_(OIA(2, (32, 64, 128, 128), 4))
    om = OM.from_unet2d(ou2, oma, oia).eval()
    hu2 = p.UNet2DConditionModel(**kw)
    hu2.load_state_dict(ou2.state_dict())
    hma = p.MotionAdapter(block_out_channels=(32, 64, 128, 128), motion_num_attention_heads=4, motion_norm_num_groups=8)
    hma.load_state_dict(oma.state_dict())
    hia = p.I2VAdapterModule(2, (32, 64, 128, 128), 4)
    hia.load_state_dict(oia.state_dict())
    hm = p.UNetMotionCrossFrameAttnModel.from_unet2d(hu2.to(dev).half(), hma, hia).eval()
    assert next(hm.parameters()).dtype == torch.float16 and next(hm.parameters()).is_cuda
    sd_o, sd_h = om.state_dict(), hm.state_dict()
    assert set(sd_o) == set(sd_h)
    for k in sd_o:
        assert torch.equal(sd_o[k].half(), sd_h[k].cpu().half()), k
    ad = hm.obtain_i2v_adapter_modules()
    assert set(ad.state_dict()) == set(oia.state_dict())
    inp = small_unet_inputs()
    with torch.no_grad():
        ref = om(inp["sample"], inp["timestep"], True, inp["ctx"]).sample
        got = hm(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
    compare(got, ref, rel=REL_TOL_UNET, name="from_unet2d model")


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipeline_trajectory(dev, use_graph):
    """config-1 style plumbing: DDIM + CFG loop with frame-0 re-injection on the reduced UNet vs the oracle loop."""
    from oracle.pipeline_i2v_adapter import I2VAdapterPipeline as OP
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    g = torch.Generator().manual_seed(31)
    pe, ne = h(torch.randn(1, 7, 64, generator=g)), h(torch.randn(1, 7, 64, generator=g))
    cond = torch.randn(1, 4, 16, 16, generator=g)
    kw = dict(num_frames=4, num_inference_steps=10, guidance_scale=7.5, frame_similarity_sample_ratio=0.9)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    ref = OP(ou)(pe, ne, cond, **kw, **gens()).frames
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    got = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond, use_graph=use_graph,
               **kw, **gens()).frames
    assert got.shape == (1, 4, 4, 16, 16)
    assert torch.equal(got[:, 0].cpu(), cond), "frame 0 must equal the condition latents exactly (pipe:699-700)"
    err, scale = compare(got, ref, rel=REL_TOL_TRAJECTORY, name="DDIM trajectory (9 steps)")
    print(f"pipeline use_graph={use_graph}: max abs latent err {err:.3e} (max|ref| {scale:.3e})")
    if not use_graph:
        # the second tolerance of SURVEY section 7: the same trajectory through the fp16-EMULATING oracle (the rounding
        # pattern of the reference's own fp16 GPU path, latents in fp16): the HIP path (fp32 latents, fp32 DDIM update)
        # must be at least as close to exact arithmetic as that, and the two fp16 trajectories must agree
        from oracle.fp16_emulation import emulate_reference_fp16
        with emulate_reference_fp16():
            emu = OP(ou)(pe, ne, cond, **kw, **gens()).frames
        err_emu = (emu - ref).abs().max().item()
        log_error("DDIM trajectory: fp16-emulated reference vs fp32 oracle", err_emu, scale, None)
        log_error("DDIM trajectory: HIP vs fp16-emulated reference", (got.cpu() - emu).abs().max().item(), scale, None)
        print(f"  fp16-emulated reference trajectory: err vs fp32 {err_emu:.3e}; HIP vs emulated "
              f"{(got.cpu() - emu).abs().max().item():.3e}")
    again = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond, use_graph=use_graph,
                 **kw, **gens()).frames
    assert torch.equal(got, again), "same seeds must reproduce the trajectory bit for bit"
    # pipe:693-697: a per-step callback under the DEFAULT mode (use_graph=True) runs the steps as eager launches instead
    # of raising; it sees every step (callback_steps = 1) and the trajectory is the same bit for bit
    seen = []
    cb = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond,
              callback=lambda i, t, lat: seen.append((i, int(t), tuple(lat.shape))), **kw, **gens()).frames
    assert [s[0] for s in seen] == list(range(9)) and seen[0][2] == (1, 4, 4, 16, 16)
    assert torch.equal(cb, got)
    assert again.data_ptr() != got.data_ptr()
    if use_graph:
        # the captured step is reused by later calls of the same shape: a DIFFERENT sample through the cached graph must
        # equal that sample through a fresh pipeline (inputs really are copied into the graph's static buffers), and the
        # first call's result must not have been overwritten
        g2 = torch.Generator().manual_seed(77)
        pe2, ne2 = h(torch.randn(1, 7, 64, generator=g2)), h(torch.randn(1, 7, 64, generator=g2))
        cond2 = torch.randn(1, 4, 16, 16, generator=g2)
        gens2 = lambda: dict(generator=torch.Generator().manual_seed(15), prior_mask_generator=torch.Generator().manual_seed(16),
                             prior_noise_generator=torch.Generator().manual_seed(17))
        keep = got.clone()
        assert len(pipe._graph_cache) == 1
        other = pipe(prompt_embeds=pe2, negative_prompt_embeds=ne2, condition_image_latents=cond2, use_graph=True,
                     **kw, **gens2()).frames
        assert len(pipe._graph_cache) == 1 and torch.equal(got, keep)
        fresh = pkg().I2VAdapterPipeline(unet=hu)(prompt_embeds=pe2, negative_prompt_embeds=ne2,
                                                  condition_image_latents=cond2, use_graph=True, **kw, **gens2()).frames
        assert torch.equal(other, fresh) and not torch.equal(other, got)


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipeline_trajectory_two_samples_with_ip(dev, use_graph):
    """the loop with TWO samples per call and the IP-Adapter branch on (config 4's path on the reduced UNet) against the
    oracle loop: CFG batch 4 ordered [neg_0, neg_1, pos_0, pos_1] (pipe:613-614), zero negative image embeds
    (pipe:343, 621-622), per-sample frame-0 re-injection (pipe:669, 699-700)."""
    from oracle.pipeline_i2v_adapter import I2VAdapterPipeline as OP
    from tests.parity import small_ip_state_dict
    ou = oracle_small_unet(ip=True)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou))
    g = torch.Generator().manual_seed(41)
    pe, ne = h(torch.randn(2, 7, 64, generator=g)), h(torch.randn(2, 7, 64, generator=g))
    ie = h(torch.randn(2, 48, generator=g))
    cond = torch.randn(2, 4, 16, 16, generator=g)
    kw = dict(num_frames=4, num_inference_steps=10, guidance_scale=7.5, frame_similarity_sample_ratio=0.5,
              image_embeds=ie, blur_sigma=0.8)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    ref = OP(ou)(pe, ne, cond, **kw, **gens()).frames
    got = pkg().I2VAdapterPipeline(unet=hu)(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond,
                                            use_graph=use_graph, **kw, **gens()).frames
    assert got.shape == (2, 4, 4, 16, 16) and torch.equal(got[:, 0].cpu(), cond)
    compare(got, ref, rel=REL_TOL_TRAJECTORY, name="DDIM trajectory, 2 samples per call + IP (5 steps)")
    assert (got[0] - got[1]).abs().max().item() > 0.1


def test_pipeline_trajectory_stochastic_ddim(dev):
    """eta > 0 (pipe:550, 659-660 `prepare_extra_step_kwargs` -> diffusers DDIMScheduler.step): sigma_t leaves the direction
    coefficient and comes back as one fresh Gaussian draw of the latents' shape per step from `generator` -- the same draws, in
    the same order, as the oracle loop makes; eta = 0 stays the deterministic path bit for bit."""
    from oracle.pipeline_i2v_adapter import I2VAdapterPipeline as OP
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    g = torch.Generator().manual_seed(33)
    pe, ne = h(torch.randn(1, 7, 64, generator=g)), h(torch.randn(1, 7, 64, generator=g))
    cond = torch.randn(1, 4, 16, 16, generator=g)
    kw = dict(num_frames=4, num_inference_steps=10, guidance_scale=7.5, frame_similarity_sample_ratio=0.9)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    run = lambda eta: pipe(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image_latents=cond, eta=eta, **kw, **gens()).frames
    det = run(0.0)
    for eta in (0.5, 1.0):
        ref = OP(ou)(pe, ne, cond, eta=eta, **kw, **gens()).frames
        got = run(eta)
        assert torch.equal(got[:, 0].cpu(), cond)
        compare(got, ref, rel=REL_TOL_TRAJECTORY, name=f"stochastic DDIM trajectory, eta = {eta}")
        assert (got - det).abs().max().item() > 0.1, "eta > 0 must change the trajectory"
        assert torch.equal(got, run(eta)), "same seeds, same trajectory"


def test_stream_forks_opt_in_child_process(dev):
    """I2V_STREAMS=1 (read at import) forks the independent chains of the 8 x 8 level onto a side stream (streams.py):
    the module, UNet and pipeline tests (oracle parity, eager == hipGraph bit for bit, cached-graph replays) re-run that way
    in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_modules_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "not opt_in_child"], cwd=root,
                       env=dict(os.environ, I2V_STREAMS="1", I2V_STREAMS_MAX_ROWS="1000000"), capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
