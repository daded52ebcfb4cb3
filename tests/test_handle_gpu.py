"""GPU tests of the model handle's whole-model forward (include/i2v_hip.h "Model handle", ABI 9): unet:1289-1451 --
`UNetMotionCrossFrameAttnModel.forward`, the call of pipe:676-683 -- issued by `i2v_unet_forward` in C from a recorded launch plan.

  * through ctypes: ONE call per forward, nothing of blocks.py / kernels.py on the call path (their launch wrappers are made to
    raise while it runs); the result equals the module API's forward BIT FOR BIT on the reduced UNet and at SD-1.5 width (+ IP-Adapter),
    eagerly and as the handle's captured + replayed step; new inputs in the same buffers are read; the registry decides what is read
    (a weight registered under a plan key changes the result exactly as the module API's does with that weight);
  * through a C host (tests/c_host/unet_forward_host.c, built by __graft_entry__.build() with gcc): plan, weights and inputs from
    files, no Python in the process.
"""
import os
import struct
import subprocess

import pytest
import torch

from tests.parity import SD15, hip_model_random, hip_unet_from_oracle, oracle_small_unet, sd15_ip_state_dict, small_ip_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def _inputs(dev, b, f, hw, ctx_dim, clip_dim=None, lt=77, seed=5, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    d = dict(sample=torch.randn(b, f, 4, hw, hw, generator=g).to(dtype).to(dev),
             t=torch.tensor([481.0, 37.0][:b] if b <= 2 else [float(10 + i) for i in range(b)], dtype=torch.float32, device=dev),
             ctx=torch.randn(b, lt, ctx_dim, generator=g).half().to(dev), ie=None)
    if clip_dim:
        d["ie"] = torch.randn(b, clip_dim, generator=g).half().to(dev)
    return d


def _module_forward(hu, inp, cross_frame=True):
    added = {"image_embeds": inp["ie"]} if inp["ie"] is not None else None
    with torch.no_grad():
        return hu(inp["sample"], inp["t"], cross_frame, inp["ctx"], added_cond_kwargs=added).sample


def _handle_for(hu, inp, cross_frame=True):
    H = pkg().handle
    blob, weights = H.record_forward_plan(hu, inp["sample"], inp["t"], inp["ctx"], image_embeds=inp["ie"],
                                          enable_cross_frame_attn=cross_frame)
    hd = pkg().UNetHandle(hu, ip_num_tokens=4 if inp["ie"] is not None else 0)
    b, f, _, hh, ww = inp["sample"].shape
    hd.plan(b, f, hh, ww, ctx_len=inp["ctx"].shape[1], has_ip=inp["ie"] is not None)
    launches, keys = hd.set_plan(blob)
    assert launches > 100 and sorted(keys) == sorted(weights)
    hd.set_weights(weights)
    arena = torch.empty(hd.activation_bytes, dtype=torch.uint8, device=inp["sample"].device)
    hd.set_workspace(arena)
    return hd, blob, weights, arena


def _no_python_launches(monkeypatch):
    """while the handle's forward runs, every launch wrapper of the host mirror raises: the launches must come from C"""
    K = pkg().kernels

    def boom(*a, **k):
        raise AssertionError("a kernels.py wrapper ran during i2v_unet_forward")
    for name in ("gemm", "conv3x3", "attention", "groupnorm", "layernorm", "ff_fused", "motion_attn", "cross_attn_fused", "ln_qkv",
                 "temporal_attention", "nchw_to_tokens", "tokens_to_nchw", "timestep_embedding", "silu", "copy3d"):
        monkeypatch.setattr(K, name, boom)


@pytest.mark.parametrize("ip,cross_frame,dtype", [(False, True, torch.float16), (True, True, torch.float32), (False, False, torch.float16)])
def test_small_unet_forward_through_the_c_abi(dev, monkeypatch, ip, cross_frame, dtype):
    ou = oracle_small_unet(ip=ip)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou) if ip else None)
    inp = _inputs(dev, 2, 4, 16, 64, clip_dim=48 if ip else None, lt=7, dtype=dtype)
    ref = _module_forward(hu, inp, cross_frame)
    hd, blob, weights, arena = _handle_for(hu, inp, cross_frame)
    out = torch.full_like(ref, float("nan"))
    with monkeypatch.context() as m:
        _no_python_launches(m)
        hd.forward(inp["sample"], inp["t"], inp["ctx"], inp["ie"], out)
        torch.cuda.synchronize()
    assert torch.equal(out, ref), f"C-ABI forward differs from the module API: max |d| {(out.float() - ref.float()).abs().max().item():.3e}"
    # other inputs in the same buffers: the plan reads the arguments, not recorded values
    inp2 = _inputs(dev, 2, 4, 16, 64, clip_dim=48 if ip else None, lt=7, seed=6, dtype=dtype)
    ref2 = _module_forward(hu, inp2, cross_frame)
    out2 = torch.empty_like(ref)
    hd.forward(inp2["sample"], inp2["t"], inp2["ctx"], inp2["ie"], out2)
    torch.cuda.synchronize()
    assert torch.equal(out2, ref2) and not torch.equal(out2, ref)
    # the handle's step: captured once, replayed; reads what the buffers hold at replay time
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    buf = {k: (v.clone() if v is not None else None) for k, v in inp.items()}
    out3 = torch.zeros_like(ref)
    hd.capture(s, lambda: hd.forward(buf["sample"], buf["t"], buf["ctx"], buf["ie"], out3, stream=s))
    assert hd.has_step
    hd.replay(s)
    s.synchronize()
    assert torch.equal(out3, ref)
    for k in ("sample", "t", "ctx", "ie"):
        if buf[k] is not None:
            buf[k].copy_(inp2[k])
    torch.cuda.synchronize()
    hd.replay(s)
    s.synchronize()
    assert torch.equal(out3, ref2)
    # a missing weight is named, and the registry decides what the forward reads
    hd2 = pkg().UNetHandle(hu, ip_num_tokens=4 if ip else 0)
    hd2.plan(2, 4, 16, 16, ctx_len=7, has_ip=ip)
    hd2.set_plan(blob)
    hd2.set_workspace(arena)
    first = sorted(weights)[0]
    hd2.set_weights({k: v for k, v in weights.items() if k != first})
    with pytest.raises(Exception, match="is not registered"):
        hd2.forward(inp["sample"], inp["t"], inp["ctx"], inp["ie"], out)
    key = next(k for k in sorted(weights) if k.endswith("conv_in.bias") or k.endswith("#b_in"))
    changed = dict(weights)
    changed[key] = pkg().handle.base_tensor(weights[key]).clone() + 0.25
    hd2.set_weights(changed)
    out4 = torch.empty_like(ref)
    hd2.forward(inp["sample"], inp["t"], inp["ctx"], inp["ie"], out4)
    torch.cuda.synchronize()
    assert not torch.equal(out4, ref) and torch.isfinite(out4).all()
    hd.close()
    hd2.close()


@pytest.mark.parametrize("ip", [False, True])
def test_full_width_forward_through_the_c_abi(dev, monkeypatch, ip):
    """SD-1.5 width, CFG-shaped batch (2, 8, 4, 32, 32): every kernel family of the timed step -- the 8-wave GEMM / conv forms, split-K,
    the fused 320-channel sub-block kernels (ln_qkv, cross_attn_fused with the prompt's fragments packed by i2v_pack_ctx_fragments_f16,
    motion_attn, ff_fused + tail), flash attention at d = 40 / 80 / 160 -- from the launch plan, bit for bit."""
    hu = hip_model_random(SD15, dev, seed=77)
    if ip:
        hu._load_ip_adapter_weights(sd15_ip_state_dict(hu))
    inp = _inputs(dev, 2, 8, 32, 768, clip_dim=1024 if ip else None)
    ref = _module_forward(hu, inp)
    hd, blob, weights, arena = _handle_for(hu, inp)
    out = torch.full_like(ref, float("nan"))
    with monkeypatch.context() as m:
        _no_python_launches(m)
        hd.forward(inp["sample"], inp["t"], inp["ctx"], inp["ie"], out)
        torch.cuda.synchronize()
    assert torch.equal(out, ref)
    sd_keys = set(hu.state_dict())
    n_sd = sum(k in sd_keys for k in weights)
    assert n_sd > 300 and all(("#" in k) != (k in sd_keys) for k in weights)      # checkpoint tensors go by their state-dict key
    print(f"SD-1.5 width, ip={ip}: {hd.set_plan(blob)[0]} launches, {len(weights)} weight keys ({n_sd} of them state-dict keys: tensors "
          f"read as the checkpoint holds them; {len(weights) - n_sd} re-laid-out packs) "
          f"({sum(pkg().handle.base_tensor(w).numel() * w.element_size() for w in weights.values()) / 1e9:.2f} GB), arena "
          f"{hd.activation_bytes / 1e9:.2f} GB, plan {len(blob) / 1e6:.2f} MB")
    hd.close()


def test_c_host_runs_the_forward_without_python(dev, tmp_path):
    """tests/c_host/unet_forward_host.c: plan.bin + weights.bin + inputs.bin -> out.bin in a process that links libi2v_hip.so and the
    HIP runtime only; its result equals the module API's forward bit for bit (it also checks eager == captured replay itself)."""
    exe = os.path.join(ROOT, "tests", "c_host", "unet_forward_host")
    if not os.path.exists(exe):
        pytest.fail("tests/c_host/unet_forward_host is not built (python __graft_entry__.py)")
    ou = oracle_small_unet(ip=True)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou))
    inp = _inputs(dev, 2, 4, 16, 64, clip_dim=48, lt=7)
    ref = _module_forward(hu, inp)
    H = pkg().handle
    blob, weights = H.record_forward_plan(hu, inp["sample"], inp["t"], inp["ctx"], image_embeds=inp["ie"])
    H.save_plan(blob, tmp_path / "plan.bin")
    H.save_weights(weights, tmp_path / "weights.bin")
    cfg = hu.config
    ints = [cfg.in_channels, cfg.out_channels, *cfg.block_out_channels, cfg.layers_per_block, cfg.num_attention_heads, cfg.cross_attention_dim,
            cfg.norm_num_groups, cfg.motion_max_seq_length, cfg.motion_num_attention_heads, 1, 4,
            2, 4, 16, 16, 7, 48, 0, 0]
    with open(tmp_path / "inputs.bin", "wb") as f:
        f.write(b"I2VI" + struct.pack("<22i", *ints))
        for t in (inp["sample"], inp["t"], inp["ctx"], inp["ie"]):
            f.write(t.cpu().contiguous().numpy().tobytes())
    r = subprocess.run([exe, str(tmp_path / "plan.bin"), str(tmp_path / "weights.bin"), str(tmp_path / "inputs.bin"),
                        str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout.strip())
    import numpy as np
    got = torch.from_numpy(np.fromfile(tmp_path / "out.bin", dtype=np.float16).copy()).view(ref.shape)
    assert torch.equal(got, ref.cpu())


# ------------------------------------------------------------------------------------------------ the whole denoising loop from C
def _prepared_state(pipe, hu, dev, B, F, h_lat, ctx_len, ctx_dim, clip_dim, seed, T=25):
    """a pipeline state as `I2VAdapterPipeline.__call__` / bench.py build it: static buffers + the per-sample buffers (pipe:629-660)"""
    g = torch.Generator().manual_seed(seed)
    sch = pipe.scheduler
    sch.set_timesteps(T)
    ts = sch.timesteps
    ie = torch.randn(2 * B, clip_dim, generator=g).half().to(dev) if clip_dim else None
    st = dict(latents=torch.randn(B, F, 4, h_lat, h_lat, generator=g).to(dev), cond=torch.randn(B, 4, h_lat, h_lat, generator=g).to(dev),
              copies=2, num_frames=F, guidance=7.5, t_table=ts.float().to(dev), coef=sch.step_coefficients(ts).to(dev),
              step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
              ctx_text=torch.randn(2 * B, ctx_len, ctx_dim, generator=g).half().to(dev),
              ctx_ip=hu._project_image_embeds({"image_embeds": ie}) if ie is not None else None)
    with torch.no_grad():
        st["ctx_proj"] = hu.project_context(st["ctx_text"], st["ctx_ip"])
        st["temb_table"] = hu.project_time_table(st["t_table"])
    return st, ie


def _python_loop(pipe, hu, st, ie, latents0, n_steps):
    """the reference loop through the host mirror: per-sample preparation, then n_steps x `_step` (pipe:663-700)"""
    with torch.no_grad():
        st["latents"].copy_(latents0)
        st["step_idx"].zero_()
        ip = hu._project_image_embeds({"image_embeds": ie}) if ie is not None else None
        hu.project_context(st["ctx_text"], ip, out=st["ctx_proj"])
        hu.project_time_table(st["t_table"], out=st["temb_table"])
        for _ in range(n_steps):
            pipe._step(st)
        torch.cuda.synchronize()
    return st["latents"].clone()


def _scramble(st, hu):
    H = pkg().handle
    for t in H.sample_buffers(hu, st).values():
        t.fill_(float("nan"))


@pytest.mark.parametrize("width", ["small", "sd15"])
def test_denoising_loop_through_the_c_abi(dev, monkeypatch, width):
    """pipe:663-700 without the host mirror on the call path: `record_prepare_plan` + `record_step_plan`, then per sample ONE
    `i2v_unet_run` (context K / V^T, ImageProjection, time-embedding table) and per step ONE graph replay of ONE `i2v_unet_run`
    (i2v_ddim_prep -> the UNet as the pipeline routes it -> i2v_ddim_cfg_step).  For a NEW prompt / image (other contents in the same
    buffers than at recording time) the latents after 4 steps equal the Python loop's bit for bit."""
    H = pkg().handle
    if width == "small":
        ou = oracle_small_unet(ip=True)
        hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou))
        dims = dict(B=1, F=4, h_lat=16, ctx_len=7, ctx_dim=64, clip_dim=48)
    else:
        hu = hip_model_random(SD15, dev, seed=78)
        hu._load_ip_adapter_weights(sd15_ip_state_dict(hu))
        dims = dict(B=1, F=8, h_lat=32, ctx_len=77, ctx_dim=768, clip_dim=1024)
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    st, ie = _prepared_state(pipe, hu, dev, seed=31, **dims)
    with torch.no_grad():
        step_blob, w_step = H.record_step_plan(pipe, st)            # (its warm-up run also makes the fused kernels' context fragments)
        prep_blob, w_prep = H.record_prepare_plan(pipe, st, image_embeds=ie)
    weights = {**w_step, **w_prep}
    assert any(k.startswith("sample#") for k in w_step) and any(k.startswith("sample#") for k in w_prep)
    problem = H._step_problem(st)
    handles = []
    for blob in (prep_blob, step_blob):
        hd = pkg().UNetHandle(hu, ip_num_tokens=4)
        hd.plan(*problem[:4], ctx_len=problem[4], has_ip=bool(problem[5]))
        hd.set_plan(blob)
        hd.set_weights(weights)
        handles.append(hd)
    hp, hs = handles
    arena = torch.empty(max(hp.activation_bytes, hs.activation_bytes), dtype=torch.uint8, device=dev)
    hp.set_workspace(arena)
    hs.set_workspace(arena)
    # a new sample in the same buffers
    g = torch.Generator().manual_seed(32)
    st["ctx_text"].copy_(torch.randn(st["ctx_text"].shape, generator=g).half())
    ie.copy_(torch.randn(ie.shape, generator=g).half())
    st["cond"].copy_(torch.randn(st["cond"].shape, generator=g))
    latents0 = torch.randn(st["latents"].shape, generator=g).to(dev)
    n_steps = 4
    ref = _python_loop(pipe, hu, st, ie, latents0, n_steps)
    _scramble(st, hu)
    st["latents"].copy_(latents0)
    st["step_idx"].zero_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with monkeypatch.context() as m:
        _no_python_launches(m)
        with torch.cuda.stream(s):
            hp.run({H.PREP_CONTEXT: st["ctx_text"], H.PREP_TIMESTEPS: st["t_table"], H.PREP_IMAGE_EMBEDS: ie}, stream=s)
        io = {H.STEP_LATENTS: st["latents"], H.STEP_COND: st["cond"], H.STEP_INDEX: st["step_idx"], H.STEP_COEF: st["coef"]}
        hs.capture(s, lambda: hs.run(io, stream=s))
        for _ in range(n_steps):
            hs.replay(s)
        s.synchronize()
    assert torch.equal(st["latents"], ref), f"max |d| {(st['latents'] - ref).abs().max().item():.3e}"
    assert int(st["step_idx"].item()) == n_steps and torch.equal(st["latents"][:, 0] * 0, ref[:, 0] * 0)
    print(f"{width}: preparation {hp.set_plan(prep_blob)[0]} launches, step {hs.set_plan(step_blob)[0]} launches, "
          f"{sum(k.startswith('sample#') for k in weights)} per-sample buffers")
    for hd in handles:
        hd.close()


def test_c_host_runs_the_denoising_loop_without_python(dev, tmp_path):
    """`handle.export_denoiser` (prepare.plan + step.plan + weights.bin + manifest.json, recorded on zeros) and
    `write_denoise_inputs` -> tests/c_host/denoise_host.c: the final latents of a process that links libi2v_hip.so and the HIP
    runtime only equal the Python loop's bit for bit."""
    import json
    exe = os.path.join(ROOT, "tests", "c_host", "denoise_host")
    if not os.path.exists(exe):
        pytest.fail("tests/c_host/denoise_host is not built (python __graft_entry__.py)")
    H = pkg().handle
    ou = oracle_small_unet(ip=True)
    hu = hip_unet_from_oracle(ou, dev, ip_state_dict=small_ip_state_dict(ou))
    pipe = pkg().I2VAdapterPipeline(unet=hu)
    dims = dict(B=1, F=4, h_lat=16, ctx_len=7, ctx_dim=64, clip_dim=48)
    manifest = H.export_denoiser(pipe, str(tmp_path), num_frames=4, latent_height=16, latent_width=16, batch=1, ctx_len=7, clip_dim=48,
                                 num_inference_steps=25, guidance_scale=7.5)
    assert json.load(open(tmp_path / "manifest.json")) == manifest and manifest["weights"]["per_sample_buffers"] > 0
    assert manifest["problem"] == dict(batch=2, frames=4, height=16, width=16, ctx_len=7, has_ip=1)
    # the sample: the Python loop on a state of its own
    st, ie = _prepared_state(pipe, hu, dev, seed=41, **dims)
    latents0 = st["latents"].clone()
    n_steps = 6
    ref = _python_loop(pipe, hu, st, ie, latents0, n_steps)
    H.write_denoise_inputs(tmp_path / "inputs.bin", hu.config, manifest, latents0, st["cond"], st["ctx_text"], ie)
    r = subprocess.run([exe, str(tmp_path / "prepare.plan"), str(tmp_path / "step.plan"), str(tmp_path / "weights.bin"),
                        str(tmp_path / "inputs.bin"), str(tmp_path / "out.bin"), str(n_steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(r.stdout.strip())
    import numpy as np
    got = torch.from_numpy(np.fromfile(tmp_path / "out.bin", dtype=np.float32).copy()).view(ref.shape)
    assert torch.equal(got, ref.cpu())
