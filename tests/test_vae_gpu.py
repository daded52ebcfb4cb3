"""GPU parity of the HIP AutoencoderKL (decode_latents pipe:300-320, condition-image encode pipe:626-627) and of the
kernels it adds (row softmax, asymmetric-pad stride-2 conv, Gaussian sample) against the CPU oracle / torch."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.parity import compare, hip_model_random, host_threads, log_error, oracle_from_hip

pytestmark = pytest.mark.gpu
SMALL_VAE = dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)
SD_VAE = dict(block_out_channels=(128, 256, 512, 512), norm_num_groups=32)


def pkg():
    import i2v_adapter_unofficial_amd as p
    return p


def K():
    return pkg().kernels


def h(t):
    return t.half().float()


def _pair(cfg, dev, seed):
    from oracle.vae import AutoencoderKL as O
    hv = hip_model_random(cfg, dev, seed=seed, cls=pkg().AutoencoderKL)
    return oracle_from_hip(hv, O, cfg), hv


@pytest.mark.parametrize("rows,cols,scale", [(5, 16, 1.0), (64, 4096, 0.044), (33, 1000, 2.5), (7, 36, 1.0)])
def test_softmax_rows(dev, rows, cols, scale):
    g = torch.Generator().manual_seed(rows + cols)
    ld = (cols + 7) // 8 * 8
    x = h(torch.randn(rows, ld, generator=g) * 3)
    xd = x.half().to(dev)
    out = K().softmax_rows(xd[:, :cols], scale)
    ref = F.softmax(x[:, :cols] * scale, dim=-1)
    assert (out.float().cpu() - ref).abs().max().item() <= 1e-3 * ref.max().item() + 1e-6
    K().softmax_rows(xd[:, :cols], scale, out=xd[:, :cols])               # in place, padded leading dimension
    assert torch.equal(xd[:, :cols], out) and torch.equal(xd[:, cols:].cpu(), x[:, cols:].half())


@pytest.mark.parametrize("n,hh,ww,cin,cout", [(2, 8, 8, 32, 48), (1, 9, 7, 64, 64), (4, 32, 32, 128, 128)])
def test_conv3x3_asym_pad(dev, n, hh, ww, cin, cout):
    """VAE encoder Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) + conv 3x3 stride 2 without padding."""
    from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
    g = torch.Generator().manual_seed(n + hh + cin)
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    b = h(torch.randn(cout, generator=g))
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    out = K().conv3x3(K().nchw_to_tokens(x.to(dev)), pack_conv3x3(w).to(dev), b.half().to(dev), stride=2, asym_pad=True)
    compare(K().tokens_to_nchw(out, dtype=torch.float32), ref, rel=3e-3, name="conv3x3 asym_pad stride 2")


def test_gaussian_sample(dev):
    g = torch.Generator().manual_seed(3)
    m = torch.randn(3, 8, 5, 7, generator=g) * 4
    m[0, 4:] = 50.0       # logvar above the clamp
    m[1, 4:] = -80.0      # below it
    eps = torch.randn(3, 4, 5, 7, generator=g)
    ref = m[:, :4] + torch.exp(0.5 * m[:, 4:].clamp(-30, 20)) * eps
    out = K().gaussian_sample(m.to(dev), eps.to(dev))
    assert torch.allclose(out.cpu(), ref, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("cfg,size,name", [(SMALL_VAE, 32, "reduced"), (SD_VAE, 64, "SD-1.5 width")])
def test_vae_encode_decode_vs_oracle(dev, cfg, size, name):
    host_threads()
    ov, hv = _pair(cfg, dev, seed=size)
    g = torch.Generator().manual_seed(size + 1)
    img = h(torch.rand(2, 3, size, size, generator=g) * 2 - 1)
    with torch.no_grad():
        od = ov.encode(img).latent_dist
        hd = hv.encode(img.to(dev)).latent_dist
        compare(hd.parameters, torch.cat([od.mean, od.logvar], dim=1), rel=4.5e-3,      # measured 1.3e-3 / 1.4e-3
                name=f"VAE encoder moments ({name})")
        z_o = od.sample(torch.Generator().manual_seed(9))
        z_h = hd.sample(torch.Generator().manual_seed(9))
        compare(z_h, z_o, rel=3e-3, name=f"VAE latent sample ({name})")
        assert z_h.shape == (2, 4, size // 8, size // 8) and z_h.dtype == torch.float32
        z = h(torch.randn(2, 4, size // 8, size // 8, generator=g))
        ref = ov.decode(z).sample
        got = hv.decode(z.to(dev)).sample
    assert got.shape == (2, 3, size, size) and got.dtype == torch.float32
    err, scale = compare(got, ref, rel=1e-2, name=f"VAE decoder ({name})")     # measured 2.8e-3 / 3.6e-3 in round 2
    print(f"VAE {name}: decoder max abs err {err:.3e} (max|ref| {scale:.3e})")


def test_vae_mid_attention_unaligned_token_count(dev):
    """6 x 6 = 36 tokens (not a multiple of 8): the padded key columns of the score / V^T buffers must contribute nothing."""
    from oracle.vae import VaeAttention as O
    kw = dict(channels=64, heads=1, norm_num_groups=8, eps=1e-6)
    hm = hip_model_random(kw, dev, seed=4, cls=pkg().vae.VaeAttention)
    om = oracle_from_hip(hm, O, kw)
    x = h(torch.randn(3, 64, 6, 6, generator=torch.Generator().manual_seed(5)))
    with torch.no_grad():
        ref = om(x)
        got = K().tokens_to_nchw(hm._fwd(K().nchw_to_tokens(x.to(dev))), dtype=torch.float32)
    compare(got, ref, rel=3e-3, name="VAE mid-block attention, 36 tokens")


def test_pipeline_with_vae_end_to_end(dev, tmp_path):
    """condition image (PIL) -> HIP VAE encode -> hipGraph DDIM loop -> HIP VAE decode -> PIL frames -> GIF
    (pipe:624-627, 663-711, 806-807) on the reduced models; the decode is checked against the oracle VAE on the
    pipeline's own final latents."""
    import numpy as np
    import PIL.Image
    from tests.parity import hip_unet_from_oracle, oracle_small_unet
    p = pkg()
    ov, hv = _pair(SMALL_VAE, dev, seed=77)
    hu = hip_unet_from_oracle(oracle_small_unet(), dev)
    pipe = p.I2VAdapterPipeline(vae=hv, unet=hu)
    assert pipe.vae_scale_factor == 8
    g = torch.Generator().manual_seed(3)
    rgb = (torch.rand(80, 72, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe, ne = h(torch.randn(1, 7, 64, generator=g)), h(torch.randn(1, 7, 64, generator=g))
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image=PIL.Image.fromarray(rgb), height=64, width=64,
              num_frames=4, num_inference_steps=10, guidance_scale=7.5, frame_similarity_sample_ratio=0.3)
    gens = lambda: dict(generator=torch.Generator().manual_seed(5), prior_mask_generator=torch.Generator().manual_seed(6),
                        prior_noise_generator=torch.Generator().manual_seed(7))
    lat = pipe(output_type="latent", **kw, **gens()).frames
    assert lat.shape == (1, 4, 4, 8, 8)
    vid = pipe(output_type="pt", **kw, **gens()).frames
    assert vid.shape == (1, 4, 3, 64, 64) and vid.dtype == torch.float32 and torch.isfinite(vid).all()
    with torch.no_grad():
        ref = ov.decode((lat[0].cpu() / hv.config["scaling_factor"])).sample
    compare(vid[0], ref, rel=1e-2, name="pipeline decode_latents vs oracle VAE")
    pipe.enable_vae_slicing()
    sliced = pipe(output_type="pt", **kw, **gens()).frames
    assert torch.equal(sliced, vid)
    frames = pipe(output_type="pil", **kw, **gens()).frames
    assert len(frames) == 1 and len(frames[0]) == 4 and frames[0][0].size == (64, 64)
    # the reference's default output is "pil" (pipe:556): with a VAE attached, no output_type means PIL frames
    dflt = pipe(**kw, **gens()).frames
    assert isinstance(dflt[0][0], PIL.Image.Image) and len(dflt[0]) == 4
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(dflt[0], frames[0]))
    path = p.export_to_gif(frames[0], os.path.join(tmp_path, "sample.gif"))
    gif = PIL.Image.open(path)
    assert gif.n_frames == 4 and gif.size == (64, 64)
    arr = pipe(output_type="np", **kw, **gens()).frames[0]
    assert arr.shape == (4, 64, 64, 3) and 0.0 <= arr.min() and arr.max() <= 1.0
    assert np.abs(arr - p.VaeImageProcessor.denormalize(vid[0].cpu()).permute(0, 2, 3, 1).numpy()).max() == 0
    # latents 4 x 4 (not a multiple of 8): three down-samplings 4 -> 2 -> 1 -> 1, the up path through forward_upsample_size (unet:1304-1311)
    small = pipe(output_type="latent", **{**kw, "height": 32, "width": 32}, **gens()).frames
    assert tuple(small.shape[-2:]) == (4, 4) and torch.isfinite(small).all()
    with pytest.raises(ValueError, match="vae"):
        p.I2VAdapterPipeline(unet=hu)(output_type="pt", **{k: v for k, v in kw.items() if k != "condition_image"},
                                      condition_image_latents=lat[:, 0], **gens())
