"""Shared builders for the parity tests: matched (oracle, HIP) module pairs with identical fp16-representable
weights, seeded synthetic inputs, the comparison helper with the stated fp16 tolerance, and
`smoke_check()` used by __graft_entry__.smoke().

Tolerance of the fp16 path (stated once, used everywhere): the HIP path stores activations in fp16 (eps = 9.8e-4)
and accumulates / normalises in fp32; against the fp32 CPU oracle run on the SAME fp16-rounded weights and inputs
an output must satisfy   max|hip - oracle| <= REL_TOL * max|oracle|.  The bounds were <= 3x the errors MEASURED on
MI355X in round 2 (profiles/r2_parity_errors.jsonl: every compare() of the GPU suite) and are <= ~2x since round 4
(profiles/r4_parity_errors.jsonl; the results are deterministic for a given binary):
  single modules (transformer block / T2D / motion module / resnet / down block): measured 3.8e-4 .. 1.02e-3
      -> REL_TOL_MODULE = 2e-3   (the lower end is the rounding of the fp16 output alone: eps / 2 = 4.9e-4)
  whole UNet forward (reduced and SD-1.5 width): measured 1.0e-3 .. 1.3e-3 of max -> REL_TOL_UNET = 3e-3
  9-step CFG DDIM trajectory: measured 1.6e-3 of max|latent| -> REL_TOL_TRAJECTORY = 3.3e-3
  single kernels: 3e-3 (tests/test_kernels_gpu.py).
The reference's own fp16 GPU path, emulated by oracle/fp16_emulation.py, measures 1.6e-3 .. 1.9e-3 on the same UNet
forwards: the HIP path must not exceed it by more than 25 % (tests/test_full_width_gpu.py).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (r4: every bound <= ~2x the error measured on MI355X, profiles/r4_parity_errors.jsonl)
REL_TOL_MODULE = 2e-3
REL_TOL_UNET = 3e-3
REL_TOL_TRAJECTORY = 3.3e-3
# SD-1.5 + AnimateDiff motion adapter + I2V-Adapter topology (unet:703-726 defaults with cross_attention_dim = 768):
# the model bench.py times (BASELINE configs[1..4])
SD15 = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
            layers_per_block=2, cross_attention_dim=768, num_attention_heads=8, norm_num_groups=32,
            motion_num_attention_heads=8, motion_max_seq_length=32)
SMALL_UNET = dict(block_out_channels=(32, 64, 128, 128), num_attention_heads=4, norm_num_groups=8,
                  cross_attention_dim=64, motion_num_attention_heads=4, motion_max_seq_length=32)


def round_fp16_(module):
    """make every parameter / buffer exactly representable in fp16 (the HIP path runs fp16 weights)."""
    with torch.no_grad():
        for p in list(module.parameters()) + list(module.buffers()):
            if p.dtype.is_floating_point:
                p.copy_(p.half().float())
    return module


def randomize_adapter_out_(module, std=0.02, seed=99):
    """a freshly assembled model has zero adapter to_out (i2v:181-182) => K1 would not contribute."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if ".i2v_adapter.to_out." in name or name.startswith("i2v_adapter.to_out."):
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return module


def log_error(name, err, scale, bound, **extra):
    """append one measured error to $I2V_PARITY_LOG (JSON lines): the source of the asserted bounds."""
    path = os.environ.get("I2V_PARITY_LOG")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(dict(name=name, err=err, max_ref=scale, rel=err / max(scale, 1e-30), bound=bound,
                                    **extra)) + "\n")


def compare(got, ref, rel=REL_TOL_MODULE, name="", abs_tol=None):
    """max|got - ref| <= rel * max|ref|  (or <= abs_tol when given).  Returns (err, max|ref|)."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    bound = abs_tol if abs_tol is not None else rel * scale
    log_error(name, err, scale, bound)
    assert err <= bound, f"{name}: max abs err {err:.4e} > bound {bound:.4e} (max|ref| {scale:.4e})"
    return err, scale


def oracle_small_unet(seed=1234, ip=False):
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel
    torch.manual_seed(seed)
    m = UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    randomize_adapter_out_(m)
    if ip:
        m._load_ip_adapter_weights(small_ip_state_dict(m))
    return round_fp16_(m).eval()


def small_ip_state_dict(oracle_unet, clip_dim=48, seed=7):
    """ip-adapter_sd15.bin layout for the small UNet: key ids 1, 3, ... in attn_processors order."""
    g = torch.Generator().manual_seed(seed)
    cross = oracle_unet.config.cross_attention_dim
    sd = {"image_proj": {"proj.weight": torch.randn(4 * cross, clip_dim, generator=g) * 0.1,
                         "proj.bias": torch.randn(4 * cross, generator=g) * 0.1,
                         "norm.weight": 1 + 0.1 * torch.randn(cross, generator=g),
                         "norm.bias": 0.1 * torch.randn(cross, generator=g)},
          "ip_adapter": {}}
    mods = dict(oracle_unet.named_modules())
    names = [n for n in oracle_unet.attn_processor_names() if n.endswith("attn2.processor") and "motion_modules" not in n]
    for i, n in enumerate(names):
        a = mods[n[: -len(".processor")]]
        sd["ip_adapter"][f"{2 * i + 1}.to_k_ip.weight"] = torch.randn(a.inner_dim, cross, generator=g) * 0.1
        sd["ip_adapter"][f"{2 * i + 1}.to_v_ip.weight"] = torch.randn(a.inner_dim, cross, generator=g) * 0.1
    return sd


def hip_unet_from_oracle(oracle_unet, device, ip_state_dict=None, dtype=torch.float16):
    import i2v_adapter_unofficial_amd as pkg
    cfg = {k: v for k, v in dict(oracle_unet.config).items()}
    cfg["encoder_hid_dim_type"] = None
    m = pkg.UNetMotionCrossFrameAttnModel.from_config(cfg)
    sd = {k: v for k, v in oracle_unet.state_dict().items() if "_ip." not in k and not k.startswith("encoder_hid_proj")}
    m.load_state_dict(sd)
    m = m.to(device=device, dtype=dtype)
    if ip_state_dict is not None:
        m._load_ip_adapter_weights(ip_state_dict)
    return m.eval()


def small_unet_inputs(b=2, f=4, hw=16, lt=7, clip_dim=48, seed=11):
    g = torch.Generator().manual_seed(seed)
    h = lambda t: t.half().float()
    return dict(sample=h(torch.randn(b, f, 4, hw, hw, generator=g)),
                timestep=torch.tensor([10, 500][:b] if b <= 2 else list(range(10, 10 + b))),
                ctx=h(torch.randn(b, lt, SMALL_UNET["cross_attention_dim"], generator=g)),
                image_embeds=h(torch.randn(b, clip_dim, generator=g)))


def host_threads():
    """oracle thread count: hosts that expose hundreds of hardware threads run the unfused fp32 graph several times
    slower with all of them than with 16 (measured on the GPU box in round 1)."""
    n = min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    return n


def sd15_ip_state_dict(unet, clip_dim=1024, seed=7):
    """ip-adapter_sd15.bin layout for any UNet of this family: key ids 1, 3, ... in attn_processors
    order (unet:1276-1279); fp16-representable values."""
    g = torch.Generator().manual_seed(seed)
    cross = unet.config["cross_attention_dim"]
    r = lambda *shape, s=0.05: (torch.randn(*shape, generator=g) * s).half().float()
    sd = {"image_proj": {"proj.weight": r(4 * cross, clip_dim, s=clip_dim ** -0.5), "proj.bias": r(4 * cross, s=0.1),
                         "norm.weight": (1 + 0.1 * torch.randn(cross, generator=g)).half().float(),
                         "norm.bias": r(cross, s=0.1)},
          "ip_adapter": {}}
    mods = dict(unet.named_modules())
    names = [n for n in unet.attn_processor_names() if n.endswith("attn2.processor") and "motion_modules" not in n]
    for i, n in enumerate(names):
        a = mods[n[: -len(".processor")]]
        sd["ip_adapter"][f"{2 * i + 1}.to_k_ip.weight"] = r(a.inner_dim, cross, s=cross ** -0.5)
        sd["ip_adapter"][f"{2 * i + 1}.to_v_ip.weight"] = r(a.inner_dim, cross, s=cross ** -0.5)
    return sd


def hip_model_random(cls_kwargs, dev, seed=1234, norm_jitter=0.1, cls=None):
    """HIP module materialised on the GPU (meta -> to_empty -> fp16) with synthetic weights drawn ON the device:
    torch's default Linear / Conv law, jittered norm affines, adapter to_out ~ N(0, 0.02^2)."""
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
    cls = cls or pkg.UNetMotionCrossFrameAttnModel
    with torch.device("meta"):
        m = cls(**cls_kwargs)
    m = m.to_empty(device=dev).half()
    init_random_weights_(m, seed=seed, norm_jitter=norm_jitter)
    return m.eval()


def oracle_from_hip(hip, oracle_cls, cls_kwargs):
    """CPU oracle with exactly the HIP module's (fp16-representable) weights."""
    with torch.device("meta"):
        o = oracle_cls(**cls_kwargs)
    o = o.to_empty(device="cpu").float()
    sd = {k: v.detach().float().cpu() for k, v in hip.state_dict().items()
          if "_ip." not in k and not k.startswith("encoder_hid_proj")}
    o.load_state_dict(sd)
    return o.eval()


def full_width_pair(dev, seed=1234, ip=False):
    """(oracle, HIP) SD-1.5-width UNetMotionCrossFrameAttnModel pair: the model bench.py times."""
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel as OracleUNet
    hip = hip_model_random(SD15, dev, seed=seed)
    ou = oracle_from_hip(hip, OracleUNet, SD15)
    if ip:
        ipsd = sd15_ip_state_dict(ou)
        ou._load_ip_adapter_weights(ipsd)
        hip._load_ip_adapter_weights(ipsd)
    return ou, hip


def smoke_check():
    """One small invocation of the hot path on cuda:0 (reduced UNet: same topology as SD-1.5, narrow channels)
    checked against the CPU oracle."""
    dev = torch.device("cuda:0")
    ou = oracle_small_unet()
    hu = hip_unet_from_oracle(ou, dev)
    inp = small_unet_inputs()
    with torch.no_grad():
        ref = ou(inp["sample"], inp["timestep"], True, inp["ctx"]).sample
        got = hu(inp["sample"].to(dev), inp["timestep"].to(dev), True, inp["ctx"].to(dev)).sample
    err, scale = compare(got, ref, rel=REL_TOL_UNET, name="smoke: small UNet forward")
    print(f"smoke ok: small UNet forward max abs err {err:.3e} (max|ref| {scale:.3e})")


# ---------------------------------------------------------------------------------------------- full-size oracle fixtures (r6)
# tests/golden/oracle_full_*.safetensors (oracle/make_full_fixtures.py): fp32 outputs of the CPU oracle at BASELINE's FULL sizes on
# weights that are re-created from a seed with the CPU generator -- the same numbers in the build container and on the GPU box --
# so that the driver's `-m gpu` run checks every configuration against the oracle in seconds (the oracle forward of config 5 alone
# takes minutes of host time).  `weights_checksum` pins the re-created weights to the ones the fixture was made with.
FULL_SEED, FULL_SEED_IP = 1234, 4321


def oracle_full_width_cpu_seeded(seed=FULL_SEED, ip=False, norm_jitter=0.1):
    """SD-1.5-width ORACLE UNet with weights drawn by torch's default initialisers under `torch.manual_seed(seed)` on the CPU
    (+ jittered norm affines, adapter to_out ~ N(0, 0.02^2), every value rounded to fp16); ip: + the synthetic IP-Adapter."""
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel as OracleUNet
    torch.manual_seed(seed)
    ou = OracleUNet(**SD15)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in ou.modules():
            if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)) and m.weight is not None:
                m.weight.add_(torch.randn(m.weight.shape, generator=g) * norm_jitter)
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * norm_jitter)
    randomize_adapter_out_(ou)
    round_fp16_(ou)
    if ip:
        ou._load_ip_adapter_weights(sd15_ip_state_dict(ou))
    return ou.eval()


def weights_checksum(module):
    """a few numbers that change when any weight does: fp64 sums of |w| over (every 7th parameter, all of them weighted by index)."""
    with torch.no_grad():
        ps = list(module.parameters())
        a = sum(p.double().abs().sum() for p in ps[::7])
        b = sum((i % 13 + 1) * p.double().sum() for i, p in enumerate(ps))
    return torch.tensor([float(a), float(b), float(len(ps))], dtype=torch.float64)


def full_forward_inputs(frames, h_lat, ip, seed=11):
    """the CFG-shaped batch bench.py's `parity` uses: the SAME latents twice against two prompts, one timestep (pipe:672-673)."""
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, frames, 4, h_lat, h_lat, generator=g).half().float()
    d = dict(sample=torch.cat([lat, lat]), ctx=torch.randn(2, 77, 768, generator=g).half().float(), t=torch.tensor([481, 481]))
    if ip:
        d["image_embeds"] = torch.randn(2, 1024, generator=g).half().float()
    return d


def trajectory_inputs(frames, h_lat, seed=21):
    """(pipeline kwargs, generator factory) of the 25-step DDIM trajectory fixture (pipe:629-700)."""
    g = torch.Generator().manual_seed(seed)
    h16 = lambda t: t.half().float()
    kw = dict(prompt_embeds=h16(torch.randn(1, 77, 768, generator=g)), negative_prompt_embeds=h16(torch.randn(1, 77, 768, generator=g)),
              condition_image_latents=torch.randn(1, 4, h_lat, h_lat, generator=g), num_frames=frames, num_inference_steps=25,
              guidance_scale=7.5, blur_sigma=1.0, frame_similarity_sample_ratio=1.0)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    return kw, gens


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
