"""GPU parity of the PRECISE residual stream (ABI 9: i2v_gemm_params.residual_lo / c_lo, i2v_ff_fused_params.res2_lo / out_lo).

The stream between the UNet's modules travels as an fp16 pair hi + lo (hi = fp16(x), lo = fp16(x - hi)): every kernel that adds a
residual and writes a module's output reads both halves and writes both.  Reference semantics: pipe:666-697 / unet:1289-1451 run in
fp32 on the reference's CPU path, where the residual adds (diffusers ResnetBlock2D `input_tensor + hidden_states`, i2v:314,
TransformerTemporalModel `hidden_states + residual`) do not round.

Checked here, through the C ABI:
  * every epilogue form that carries the pair (8-wave 256- / 128-row tiles, row-permuted store, deep-pipeline tiles, split-K reduce,
    the generic kernel, the 3x3 convolution, the fused feed-forward's tail) against an fp64 reference of the same fp16 operands:
    hi + lo within 2e-5 of max|ref| (the fp32 accumulation's own noise; an fp16 result alone is 2.4e-4), hi == fp16(hi + lo);
  * with no low half coming in, hi is bit-identical to the default epilogue's result;
  * the SD-1.5-width UNet forward in precise mode against the fp32 oracle: closer than the default mode (measured r6: rms 3.7e-4 ->
    see profiles/r6_parity_modes.jsonl).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PAIR_TOL = 2e-5        # |hi + lo - fp64 reference| / max|ref|


def K():
    import i2v_adapter_unofficial_amd as pkg
    return pkg.kernels


def h(t):
    return t.half().float()


def pair(t32, dev):
    """fp32 tensor -> (hi on the device with its low half attached, the value hi + lo in fp64)"""
    hi = t32.half()
    lo = (t32 - hi.float()).half()
    d = hi.to(dev)
    d._i2v_lo = lo.to(dev)
    return d, hi.double() + lo.double()


def check_pair(out, ref64, name):
    k = K()
    lo = k.lo_of(out)
    assert lo is not None and lo.shape == out.shape and lo.dtype == torch.float16, f"{name}: no low half attached"
    hi64, lo64 = out.double().cpu(), lo.double().cpu()
    scale = ref64.abs().max().item()
    err = (hi64 + lo64 - ref64).abs().max().item()
    assert err <= PAIR_TOL * scale, f"{name}: |hi + lo - ref| = {err:.3e} at max|ref| {scale:.3e}"
    # hi is the fp16 rounding of the pair's value (what every consumer reads as an operand): lo is at most half an ulp of it
    assert bool((lo64.abs() <= 2.0 ** -11 * hi64.abs() * 1.002 + 6.2e-8).all()), f"{name}: |lo| exceeds half an ulp of hi"
    err_hi = (hi64 - ref64).abs().max().item()
    assert err_hi > 4 * err, f"{name}: the low half carries nothing (hi alone {err_hi:.3e}, pair {err:.3e})"
    return err / scale


@pytest.mark.parametrize("M,N,K_,kind", [
    (49152, 320, 320, "plain"),            # 8-wave kernel, 256-row tiles
    (16500, 320, 192, "plain"),            # 128-row tiles, ragged M
    (16384, 320, 64, "rowperm"),           # the motion module's exit
    (2048, 1280, 1280, "plain"),           # deep-pipeline form (8 x 8 level)
    (2048, 1280, 1280, "rowperm"),
    (2048, 1280, 5120, "plain"),           # split-K + reduce kernel
    (300, 200, 136, "plain"),              # generic kernel, scalar tails
    (1000, 320, 320, "plain"),
    (49152, 320, 128, "dual")])            # conv_shortcut over the skip concat (two K ranges), no residual
def test_gemm_precise_pair(dev, M, N, K_, kind):
    k = K()
    g = torch.Generator().manual_seed(M + 5 * N + K_)
    a = h(torch.randn(M, K_, generator=g))
    w = h(torch.randn(N, K_, generator=g) / math.sqrt(K_))
    b = h(torch.randn(N, generator=g))
    ad, wd, bd = a.half().to(dev), w.half().to(dev), b.half().to(dev)
    r32 = torch.randn(M, N, generator=g) * 3
    rd, r64 = pair(r32, dev)
    if kind == "dual":
        a2 = h(torch.randn(M, 64, generator=g))
        w2 = h(torch.randn(N, K_ + 64, generator=g) / 12)
        out = k.gemm(ad, w2.half().to(dev), bd, a2=a2.half().to(dev), precise=True)
        check_pair(out, torch.cat([a, a2], 1).double() @ w2.double().T + b.double(), "dual-source gemm")
        assert torch.equal(out, k.gemm(ad, w2.half().to(dev), bd, a2=a2.half().to(dev)))
        return
    store, y = {}, a.double() @ w.double().T + b.double()
    if kind == "rowperm":
        B_, F_ = 2, 16
        HW = M // (B_ * F_)
        store = dict(store=k.I2V_STORE_ROWPERM, frames=F_, hw=HW)
        y = y.reshape(B_, HW, F_, N).permute(0, 2, 1, 3).reshape(M, N)
    out = k.gemm(ad, wd, bd, residual=rd, out_scale=0.5, precise=True, **store)
    rel = check_pair(out, (y + r64) * 0.5, f"gemm {kind} {M}x{N}x{K_}")
    # without a low half coming in: hi bit-identical to the default epilogue, the pair still exact
    plain = r32.half().to(dev)
    out0 = k.gemm(ad, wd, bd, residual=plain, precise=True, **store)
    assert torch.equal(out0, k.gemm(ad, wd, bd, residual=plain, **store)), "hi differs from the default epilogue's result"
    check_pair(out0, y + r32.half().double(), f"gemm {kind} without residual_lo")
    print(f"gemm {kind} {M}x{N}x{K_}: pair error {rel:.2e} of max")


def _pack_conv(w):
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
    return pack_conv3x3(w)


@pytest.mark.parametrize("n,hh,cin,cout,stride,res", [(8, 32, 320, 320, 1, True), (4, 32, 640, 320, 1, True), (32, 8, 1280, 1280, 1, True),
                                                      (4, 32, 320, 320, 2, False), (2, 16, 8, 320, 1, False)])
def test_conv3x3_precise_pair(dev, n, hh, cin, cout, stride, res):
    """ResnetBlock2D conv2 + residual (unet:203-214), Downsample2D (unet:250-259) and conv_in (unet:757-759, 8 padded channels:
    the generic kernel) as producers of the precise stream."""
    k = K()
    g = torch.Generator().manual_seed(n + hh + cin + cout)
    x = h(torch.randn(n, cin, hh, hh, generator=g))
    wc = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    bc = h(torch.randn(cout, generator=g))
    ref = F.conv2d(x.double(), wc.double(), bc.double(), padding=1, stride=stride)
    xt = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    kw = {}
    if res:
        rd, r64 = pair(torch.randn(n, hh, hh, cout, generator=g) * 2, dev)
        kw["residual"] = rd
        ref = ref + r64.permute(0, 3, 1, 2)
    out = k.conv3x3(xt, _pack_conv(wc).to(dev), bc.half().to(dev), stride=stride, precise=True, **kw)
    check_pair(out, ref.permute(0, 2, 3, 1).contiguous(), f"conv3x3 {n}x{hh}^2 {cin}->{cout} s{stride}")
    if res:
        plain = kw["residual"].clone()                        # (a clone carries no low half)
        assert torch.equal(k.conv3x3(xt, _pack_conv(wc).to(dev), bc.half().to(dev), residual=plain, precise=True),
                           k.conv3x3(xt, _pack_conv(wc).to(dev), bc.half().to(dev), residual=plain))


@pytest.mark.parametrize("batch,frames,hw", [(3, 0, 0), (1, 16, 8), (2, 16, 200), (2, 32, 36)])
def test_ff_fused_tail_precise_pair(dev, batch, frames, hw):
    """the fused feed-forward's tail (proj_out + the module's residual, i2v:298-314 / the motion module's exit) with the residual
    as a pair: hi + lo against fp64 of the launch's own fp16 intermediate, hi bit-identical to the default tail when lo is absent."""
    k = K()
    c, inner, eps = 320, 1280, 1e-5
    rows = batch * frames * hw if frames else 128 * 5 * batch
    g = torch.Generator().manual_seed(rows + frames)
    x = h(torch.randn(rows, c, generator=g) * 1.2 + 0.2)
    gamma, beta = h(1 + 0.2 * torch.randn(c, generator=g)), h(0.1 * torch.randn(c, generator=g))
    w1, b1 = h(torch.randn(2 * inner, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(2 * inner, generator=g))
    w2, b2 = h(torch.randn(c, inner, generator=g) * inner ** -0.5), h(0.1 * torch.randn(c, generator=g))
    w3, b3 = h(torch.randn(c, c, generator=g) * c ** -0.5), h(0.1 * torch.randn(c, generator=g))
    D = lambda t: t.half().to(dev)
    packed = k.pack_ff_fused(D(w1), D(b1), D(w2), D(b2))
    ptail = k.pack_ff_tail(D(w3), D(b3))
    xd = D(x)
    rd, r64 = pair(torch.randn(rows, c, generator=g) * 2, dev)
    yd = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps)          # the block's fp16 output the tail projects
    z = yd.double().cpu() @ w3.double().T + b3.double()
    if frames:
        z = z.view(batch, hw, frames, c).permute(0, 2, 1, 3).reshape(rows, c)
    out = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=(ptail, rd, frames, hw), precise=True)
    check_pair(out, z + r64, f"ff_fused tail frames={frames}")
    plain = rd.clone()
    out0 = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=(ptail, plain, frames, hw), precise=True)
    assert k.lo_of(out0) is None            # a residual without a low half: the default tail
    zero = rd.clone()
    zero._i2v_lo = torch.zeros_like(zero)
    out1 = k.ff_fused(xd, D(gamma).float(), D(beta).float(), packed, eps=eps, tail=(ptail, zero, frames, hw), precise=True)
    assert torch.equal(out1, out0), "hi differs from the default tail's result"


def test_precise_stream_rejects_what_it_cannot_carry(dev):
    k = K()
    a, w = torch.randn(256, 64, device=dev).half(), torch.randn(128, 64, device=dev).half()
    with pytest.raises(ValueError, match="precise"):
        k.gemm(a, w, epilogue=k.I2V_EPI_GEGLU, precise=True)
    import ctypes as C
    from i2v_adapter_unofficial_amd import _lib
    lib = _lib.load()
    p = _lib.GemmParams()
    out, lo = torch.empty(256, 128, device=dev).half(), torch.empty(256, 128, device=dev).half()
    p.a, p.lda, p.w, p.ldw, p.c, p.ldc = a.data_ptr(), 64, w.data_ptr(), 64, out.data_ptr(), 128
    p.M, p.N, p.K, p.out_scale = 256, 128, 64, 1.0
    p.residual_lo = lo.data_ptr()                      # without `residual`
    assert lib.i2v_gemm_f16(C.byref(p), None) == -1 and b"residual_lo" in lib.i2v_last_error()
    p.residual_lo, p.c_lo, p.epilogue = None, lo.data_ptr(), k.I2V_EPI_GELU
    assert lib.i2v_gemm_f16(C.byref(p), None) == -1 and b"precise" in lib.i2v_last_error()


@pytest.fixture(scope="module")
def small_pair(dev):
    from tests.parity import hip_unet_from_oracle, oracle_small_unet
    ou = oracle_small_unet()
    return ou, hip_unet_from_oracle(ou, dev)


def test_small_unet_precise_mode(dev, small_pair):
    """the reduced UNet (every module kind, the generic / deep kernels) in both modes against the oracle; the mode is restored."""
    from i2v_adapter_unofficial_amd import blocks
    from tests.parity import REL_TOL_UNET, compare, small_unet_inputs
    ou, hu = small_pair
    inp = small_unet_inputs()
    entry = blocks.set_precise_stream(False)          # (whatever I2V_STREAM_PRECISE set for the process: this test switches itself)
    try:
        with torch.no_grad():
            ref = ou(inp["sample"], inp["timestep"], True, inp["ctx"]).sample
            base = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
            blocks.set_precise_stream(True)
            got = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
            blocks.set_precise_stream(False)
            again = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
    finally:
        blocks.set_precise_stream(entry)
    assert torch.equal(again, base), "switching the precise stream off must restore the default path bit for bit"
    e1, scale = compare(got, ref, rel=REL_TOL_UNET, name="small UNet, precise stream")
    e0, _ = compare(base, ref, rel=REL_TOL_UNET, name="small UNet, default stream")
    rms = lambda t: (t.float().cpu() - ref).pow(2).mean().sqrt().item()
    print(f"small UNet vs oracle: default max {e0:.3e} rms {rms(base):.3e}; precise max {e1:.3e} rms {rms(got):.3e} (max|ref| {scale:.3e})")
    assert rms(got) < rms(base), "the precise stream must bring the forward closer to the fp32 oracle"


def test_full_width_unet_precise_mode(dev):
    """SD-1.5 width, (2, 8, 4, 32, 32): the precise stream through the 8-wave kernels, the fused feed-forward tails (64^2-level
    shapes appear at 32^2 here as the 320-channel level), conv_shortcut / samplers, vs the fp32 oracle; must beat the default mode's RMS."""
    from i2v_adapter_unofficial_amd import blocks
    from tests.parity import compare, full_width_pair, host_threads
    host_threads()
    ou, hu = full_width_pair(dev, seed=1234, ip=False)
    g = torch.Generator().manual_seed(3)
    sample, ctx = h(torch.randn(2, 8, 4, 32, 32, generator=g)), h(torch.randn(2, 77, 768, generator=g))
    t = torch.tensor([481, 481])
    entry = blocks.set_precise_stream(False)
    try:
        with torch.no_grad():
            ref = ou(sample, t, True, ctx).sample
            base = hu(sample.to(dev), t.to(dev), True, ctx.to(dev)).sample
            blocks.set_precise_stream(True)
            got = hu(sample.to(dev), t.to(dev), True, ctx.to(dev)).sample
    finally:
        blocks.set_precise_stream(entry)
    e1, scale = compare(got, ref, abs_tol=4.2e-3, name="full-width UNet forward, precise stream")
    e0, _ = compare(base, ref, abs_tol=4.2e-3, name="full-width UNet forward, default stream (same weights)")
    rms = lambda t_: (t_.float().cpu() - ref).pow(2).mean().sqrt().item()
    print(f"full-width UNet vs oracle: default max {e0:.3e} rms {rms(base):.3e}; precise max {e1:.3e} rms {rms(got):.3e} "
          f"(max|ref| {scale:.3e})")
    assert rms(got) < 0.95 * rms(base)


def test_pipeline_graph_is_recaptured_when_the_mode_changes(dev, small_pair):
    """the pipeline keeps its captured step across calls (pipe:666-697 as one hipGraph): the residual-stream mode is part of what a
    captured step has baked in, so switching it must re-capture -- the same pipeline object gives the other mode's latents, equal to
    what a fresh pipeline in that mode computes, and switching back reproduces the first result bit for bit."""
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd import blocks
    _, hu = small_pair
    g = torch.Generator().manual_seed(4)
    kw = dict(prompt_embeds=h(torch.randn(1, 7, 64, generator=g)), negative_prompt_embeds=h(torch.randn(1, 7, 64, generator=g)),
              condition_image_latents=torch.randn(1, 4, 16, 16, generator=g), num_frames=4, num_inference_steps=4, guidance_scale=7.5,
              blur_sigma=1.0)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    pipe = pkg.I2VAdapterPipeline(unet=hu)
    entry = blocks.set_precise_stream(False)
    try:
        a = pipe(**kw, **gens()).frames.clone()
        blocks.set_precise_stream(True)
        b = pipe(**kw, **gens()).frames.clone()
        fresh = pkg.I2VAdapterPipeline(unet=hu)(**kw, **gens()).frames
        blocks.set_precise_stream(False)
        c = pipe(**kw, **gens()).frames
    finally:
        blocks.set_precise_stream(entry)
    assert not torch.equal(a, b) and torch.equal(b, fresh) and torch.equal(a, c)
    # ... and per call: `precise_stream=` of `__call__` (restores the process setting)
    d = pipe(**kw, **gens(), precise_stream=True).frames
    assert torch.equal(d, b) and blocks.precise_stream() == entry
    assert (a - b).abs().max().item() < 2e-2 * a.abs().max().item()
