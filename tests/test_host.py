"""Host-side logic of the HIP package that needs no GPU: weight repacking, scheduler tables, sharding arithmetic,
state-dict compatibility with the oracle, loud failure on CPU tensors."""
import pytest
import torch

from tests.parity import SMALL_UNET, oracle_small_unet


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def test_pack_conv3x3_orders():
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3, pack_geglu
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = pack_conv3x3(w, cin_pad=8)
    assert p.shape == (2, 72) and p.dtype == torch.float16
    for co in range(2):
        for ky in range(3):
            for kx in range(3):
                for ci in range(8):
                    exp = w[co, ci, ky, kx] if ci < 3 else 0
                    assert p[co, (ky * 3 + kx) * 8 + ci] == exp
    # channel counts that are multiples of 64 are laid out channel-block-major: k = ((ci // 64) * 9 + tap) * 64 + ci % 64
    w2 = torch.randn(3, 128, 3, 3)
    p2 = pack_conv3x3(w2)
    assert p2.shape == (3, 9 * 128)
    for co, ci, ky, kx in ((0, 0, 0, 0), (1, 63, 2, 1), (2, 64, 1, 2), (0, 127, 2, 2), (1, 70, 0, 1)):
        assert p2[co, ((ci // 64) * 9 + ky * 3 + kx) * 64 + ci % 64] == w2[co, ci, ky, kx].half()
    wg, bg = pack_geglu(torch.arange(8.0).reshape(4, 2), torch.arange(4.0))
    assert wg[:, 0].tolist() == [0, 4, 2, 6] and bg.tolist() == [0, 2, 1, 3]       # (value_i, gate_i) interleaved


def test_scheduler_tables_match_oracle():
    from oracle.blocks import DDIMScheduler as O
    P = pkg().DDIMScheduler
    o, p = O(), P()
    o.set_timesteps(25)
    p.set_timesteps(25)
    assert torch.equal(o.timesteps, p.timesteps) and torch.equal(o.alphas_cumprod, p.alphas_cumprod)
    ts = p.timesteps[3:]
    coef = p.step_coefficients(ts)
    x, eps = torch.randn(5), torch.randn(5)
    for i, t in enumerate(ts.tolist()):
        sa, sb, pa, pb = coef[i]
        assert torch.allclose(pa * (x - sb * eps) / sa + pb * eps, o.step(eps, t, x), atol=1e-6)
    n = torch.randn(2, 3, 4)
    assert torch.equal(o.add_noise(x.new_ones(2, 3, 4), n, torch.tensor([5, 900])),
                       p.add_noise(x.new_ones(2, 3, 4), n, torch.tensor([5, 900])))


def test_state_dict_compatible_with_oracle_and_reference_layout():
    ou = oracle_small_unet()
    m = pkg().UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    assert set(m.state_dict()) == set(ou.state_dict())
    m.load_state_dict(ou.state_dict())
    ad = m.obtain_i2v_adapter_modules()
    assert all("i2v_adapter" in k for k in ad.state_dict()) and len(ad.state_dict()) == 80
    assert len(m.obtain_motion_modules().state_dict()) > 0
    assert m.attn_processor_names() == ou.attn_processor_names()
    m.freeze_unet_params()
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert trainable and all(".i2v_adapter.to_q." in n or ".i2v_adapter.to_out." in n for n in trainable)


def test_cpu_tensors_fail_loudly():
    p = pkg()
    m = p.I2VAdapterTransformerBlock(64, 8, 8, cross_attention_dim=32)
    with pytest.raises(p.HipLibraryError, match="no CPU fallback"):
        m(torch.zeros(2, 16, 64), encoder_hidden_states=torch.zeros(2, 7, 32))
    u = p.UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    with pytest.raises(p.HipLibraryError, match="no CPU fallback"):
        u(torch.zeros(1, 2, 4, 16, 16), 10, True, torch.zeros(1, 7, 64))
    with pytest.raises(p.HipLibraryError):
        p.kernels.layernorm(torch.zeros(4, 8, dtype=torch.float16), torch.ones(8, dtype=torch.float16),
                            torch.zeros(8, dtype=torch.float16), 1e-5)


def test_pipeline_call_defaults_follow_reference():
    """pipe:556 (`output_type="pil"`) and pipe:693-697 (per-step callback): resolved on the host, no GPU needed."""
    import inspect
    p = pkg()
    u = p.UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    sig = inspect.signature(p.I2VAdapterPipeline.__call__)
    assert sig.parameters["output_type"].default is None and sig.parameters["use_graph"].default is True
    no_vae = p.I2VAdapterPipeline(unet=u)
    assert no_vae._resolve_call_defaults(None, None, True) == ("latent", True)
    assert no_vae._resolve_call_defaults("pt", None, True) == ("pt", True)
    # a callback under the default mode falls back to eager launches instead of raising
    assert no_vae._resolve_call_defaults(None, lambda i, t, x: None, True) == ("latent", False)
    from i2v_adapter_unofficial_amd.vae import AutoencoderKL
    vae = AutoencoderKL(block_out_channels=(32, 32, 64, 64), norm_num_groups=8, layers_per_block=1)
    with_vae = p.I2VAdapterPipeline(vae=vae, unet=u)
    assert with_vae._resolve_call_defaults(None, None, True) == ("pil", True)
    assert with_vae._resolve_call_defaults("latent", None, False) == ("latent", False)


def test_constructor_validation_matches_reference():
    p = pkg()
    with pytest.raises(ValueError, match="same number of `down_block_types`"):
        p.UNetMotionCrossFrameAttnModel(down_block_types=("DownBlockMotion",), up_block_types=("UpBlockMotion",) * 2,
                                        block_out_channels=(32,))
    with pytest.raises(ValueError, match="block_out_channels"):
        p.UNetMotionCrossFrameAttnModel(block_out_channels=(32, 64))
    with pytest.raises(ValueError, match="cross_attention_dim must be specified"):
        from i2v_adapter_unofficial_amd.unet_motion_cross_frame_attn import get_down_block
        get_down_block("CrossFrameAttnDownBlockMotion", 1, 32, 32, 128, True, 1e-5, "silu", 4, resnet_groups=8)


def test_shard_range_partitions_everything():
    from i2v_adapter_unofficial_amd.sharding import shard_items, shard_range
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    assert shard_items(list(range(64)), 3, 8) == list(range(24, 32))          # config 4: 64 pairs, 8 per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_checkpoint_roundtrip_reference_file_layout(tmp_path):
    """save_pretrained / from_pretrained over diffusers' files (config.json + diffusion_pytorch_model.safetensors):
    the containers the reference loads at pipe:733-746 and writes at unet:1080-1116."""
    import json
    import os
    p = pkg()
    torch.manual_seed(3)
    m = p.UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
    init_random_weights_(m, seed=5, norm_jitter=0.1)
    # adapter: reference layout = <dir>/i2v_adapter/{config.json, diffusion_pytorch_model.safetensors}
    m.save_i2v_adapter_modules(str(tmp_path / "i2v_adapter"))
    m.save_motion_modules(str(tmp_path / "motion_modules"))
    cfg = json.load(open(tmp_path / "i2v_adapter" / "config.json"))
    assert cfg["_class_name"] == "I2VAdapterModule" and cfg["block_depth"] == 2 and cfg["num_attention_heads"] == 4
    assert cfg["block_out_channels"] == [32, 64, 128, 128]
    assert os.path.isfile(tmp_path / "i2v_adapter" / "diffusion_pytorch_model.safetensors")
    ad = p.I2VAdapterModule.from_pretrained(str(tmp_path / "i2v_adapter"))
    mo = p.MotionAdapter.from_pretrained(str(tmp_path), subfolder="motion_modules", torch_dtype=torch.float16)
    assert next(mo.parameters()).dtype == torch.float16
    sd = m.state_dict()
    for k, v in ad.state_dict().items():
        assert torch.equal(v, sd[k]), k
    for k, v in mo.state_dict().items():
        assert torch.equal(v.float(), sd[k].half().float()), k
    # a second model picks the saved parts up through the reference's loaders (unet:1028-1041)
    m2 = p.UNetMotionCrossFrameAttnModel(**SMALL_UNET)
    m2.load_i2v_adapter(ad)
    m2.load_motion_modules(p.MotionAdapter.from_pretrained(str(tmp_path / "motion_modules")))
    for k, v in m2.state_dict().items():
        if "i2v_adapter" in k or "motion_modules" in k:
            assert torch.equal(v, sd[k]), k
    # whole model + the pickle variant + variant file names
    m.save_pretrained(str(tmp_path / "unet"), safe_serialization=False, variant="fp16")
    assert os.path.isfile(tmp_path / "unet" / "diffusion_pytorch_model.fp16.bin")
    m3 = p.UNetMotionCrossFrameAttnModel.from_pretrained(str(tmp_path / "unet"), variant="fp16")
    assert all(torch.equal(v, sd[k]) for k, v in m3.state_dict().items())
    with pytest.raises(EnvironmentError):
        p.I2VAdapterModule.from_pretrained(str(tmp_path / "nowhere"))
    # IP-Adapter file (ip-adapter_sd15.bin layout, pipe:783) in both serialisations
    from i2v_adapter_unofficial_amd.checkpoint import load_ip_adapter_file
    from tests.parity import sd15_ip_state_dict
    ipsd = sd15_ip_state_dict(m, clip_dim=48)
    torch.save(ipsd, tmp_path / "ip-adapter_sd15.bin")
    back = load_ip_adapter_file(str(tmp_path), weight_name="ip-adapter_sd15.bin")
    assert set(back) == {"image_proj", "ip_adapter"} and torch.equal(back["ip_adapter"]["31.to_k_ip.weight"],
                                                                      ipsd["ip_adapter"]["31.to_k_ip.weight"])
    from safetensors.torch import save_file
    flat = {f"{a}.{k}": v for a, d in ipsd.items() for k, v in d.items()}
    save_file(flat, str(tmp_path / "ip-adapter_sd15.safetensors"))
    back = load_ip_adapter_file(str(tmp_path / "ip-adapter_sd15.safetensors"))
    assert torch.equal(back["image_proj"]["proj.weight"], ipsd["image_proj"]["proj.weight"])
    m._load_ip_adapter_weights(back)
    procs = m.attn_processors
    assert sum(v.num_tokens == 4 for v in procs.values()) == 16
    m.set_attn_processor({k: type(v)(0) for k, v in procs.items()})
    assert all(v.num_tokens == 0 for v in m.attn_processors.values())
    with pytest.raises(ValueError, match="number of processors"):
        m.set_attn_processor({"x": procs[next(iter(procs))]})


def test_vae_container_layout_and_image_plumbing(tmp_path):
    """AutoencoderKL mirrors diffusers' SD-1.5 VAE state-dict layout (248 tensors) and round-trips through the
    reference's checkpoint files; VaeImageProcessor pre / post-processing and GIF export are host-side plumbing."""
    import numpy as np
    import PIL.Image
    p = pkg()
    with torch.device("meta"):
        v = p.AutoencoderKL()
    sd = {k: tuple(t.shape) for k, t in v.state_dict().items()}
    assert len(sd) == 248
    assert sd["encoder.conv_in.weight"] == (128, 3, 3, 3) and sd["encoder.conv_out.weight"] == (8, 512, 3, 3)
    assert sd["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)
    assert sd["encoder.down_blocks.2.downsamplers.0.conv.weight"] == (512, 512, 3, 3)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sd
    assert sd["decoder.mid_block.attentions.0.to_q.weight"] == (512, 512) and sd["decoder.mid_block.attentions.0.to_q.bias"] == (512,)
    assert sd["decoder.mid_block.attentions.0.group_norm.weight"] == (512,)
    assert sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert sd["decoder.up_blocks.3.resnets.2.conv2.weight"] == (128, 128, 3, 3) and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert sd["decoder.conv_out.weight"] == (3, 128, 3, 3) and sd["quant_conv.weight"] == (8, 8, 1, 1)
    assert sd["post_quant_conv.weight"] == (4, 4, 1, 1)
    from oracle.vae import AutoencoderKL as O
    with torch.device("meta"):
        assert {k: tuple(t.shape) for k, t in O().state_dict().items()} == sd
    small = p.AutoencoderKL(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)
    small.save_pretrained(str(tmp_path / "vae"))
    back = p.AutoencoderKL.from_pretrained(str(tmp_path / "vae"))
    assert back.config["scaling_factor"] == 0.18215 and tuple(back.config["block_out_channels"]) == (32, 64, 64, 64)
    assert all(torch.equal(a, b) for a, b in zip(small.state_dict().values(), back.state_dict().values()))
    with pytest.raises(p.HipLibraryError, match="no CPU fallback"):
        small.decode(torch.zeros(1, 4, 4, 4))
    proc = p.VaeImageProcessor(vae_scale_factor=8)
    rgb = (np.random.RandomState(0).rand(37, 50, 3) * 255).astype("uint8")
    t = proc.preprocess(PIL.Image.fromarray(rgb), height=32, width=48)
    assert t.shape == (1, 3, 32, 48) and -1.0 <= t.min() and t.max() <= 1.0
    t2 = proc.preprocess(PIL.Image.fromarray(rgb[:32, :48]))
    assert torch.equal(t2, torch.from_numpy(rgb[:32, :48].astype("float32") / 255).permute(2, 0, 1)[None] * 2 - 1)
    pil = proc.postprocess(t2, "pil")
    assert np.array_equal(np.asarray(pil[0]), rgb[:32, :48])
    vid = p.tensor2vid(t2[None].repeat(1, 3, 1, 1, 1), proc, "pil")
    path = p.export_to_gif(vid[0], str(tmp_path / "x.gif"))
    assert PIL.Image.open(path).n_frames >= 1


def test_vae_from_pretrained_converts_deprecated_attention_names(tmp_path):
    """SD-1.5 `vae/` folders in the wild (runwayml/stable-diffusion-v1-5, sd-vae-ft-mse) store the mid-block attention
    as `attentions.0.{query,key,value,proj_attn}` (some as 1x1 convolutions); diffusers renames them on load
    (`_convert_deprecated_attention_blocks`, reached from pipe:754).  `from_pretrained` must accept both layouts."""
    import json
    from safetensors.torch import save_file
    p = pkg()
    from i2v_adapter_unofficial_amd.vae import AutoencoderKL
    from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
    kw = dict(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8, latent_channels=4)
    vae = init_random_weights_(AutoencoderKL(**kw), seed=9, norm_jitter=0.1)
    sd = vae.state_dict()
    ren = ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn."))
    old = {}
    for k, v in sd.items():
        nk = k
        if ".attentions." in k:
            for new_name, old_name in ren:
                nk = nk.replace(new_name, old_name)
            if nk != k and v.dim() == 2 and "encoder" in k:      # the conv-shaped variant of the same weights
                v = v[:, :, None, None]
        old[nk] = v.contiguous()
    assert any(".query." in k for k in old) and not any(".to_q." in k for k in old)
    d = tmp_path / "vae"
    d.mkdir()
    json.dump({"_class_name": "AutoencoderKL", **{k: list(v) if isinstance(v, tuple) else v for k, v in kw.items()}},
              open(d / "config.json", "w"))
    save_file(old, str(d / "diffusion_pytorch_model.safetensors"))
    back = AutoencoderKL.from_pretrained(str(d))
    for k, v in back.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # an unknown key is still an error, not silently dropped
    old["decoder.mid_block.attentions.0.bogus.weight"] = torch.zeros(1)
    save_file(old, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError, match="unexpected keys"):
        AutoencoderKL.from_pretrained(str(d))


def test_fresh_process_import_order():
    """every lazily exported class resolves in a fresh interpreter whatever is touched first (a submodule imported through
    the package's lazy `__getattr__` once re-entered it and died in a circular import on the GPU box only)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for first in ("UNetMotionCrossFrameAttnModel", "I2VAdapterPipeline", "TransformerTemporalModel", "AutoencoderKL",
                  "I2VAdapterModule"):
        code = (f"import sys; sys.path.insert(0, {root!r}); import i2v_adapter_unofficial_amd as p; p.{first}; "
                "from i2v_adapter_unofficial_amd import streams, blocks, sharding; p.I2VAdapterPipeline; print('ok')")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, first + "\n" + r.stderr[-1500:]


def test_lazy_pack_builds_fused_operands_on_first_use_only():
    """blocks.LazyPack: the fused kernels' operands are registered by `_pack` but built on first access -- `in` and
    `get` see them without building, a group of operands made together is built once."""
    from i2v_adapter_unofficial_amd.blocks import LazyPack
    calls = []
    p = LazyPack(a=1)
    p.lazy("b", lambda: calls.append("b") or 2)
    p.lazy_group(("c", "d"), lambda: calls.append("cd") or (3, 4))
    assert "b" in p and "c" in p and "zz" not in p and calls == []
    assert p["a"] == 1 and calls == []
    assert p["b"] == 2 and p["b"] == 2 and calls == ["b"]
    assert p.get("d") == 4 and p["c"] == 3 and calls == ["b", "cd"]
    assert p.get("zz", 7) == 7
    import pytest
    with pytest.raises(KeyError):
        p["zz"]
