"""End-to-end time of one sample through I2VAdapterPipeline (25 DDIM steps, CFG, 16 f x 512 x 512, VAE encode of the
condition image + decode of the clip), random SD-1.5-width weights.  Prompt / image embeddings are inputs (the CLIP
encoders are out of scope)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
dev = torch.device("cuda:0")
unet = bench.build_hip_model(dev, seed=1234)
vae = pkg.AutoencoderKL()
init_random_weights_(vae, seed=3)
vae = vae.to(device=dev, dtype=torch.float16).eval()
pipe = pkg.I2VAdapterPipeline(unet=unet, vae=vae)
g = torch.Generator().manual_seed(1)
pe, ne = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
img = torch.rand(1, 3, 512, 512, generator=g)
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, condition_image=img, num_frames=16, height=512, width=512,
          num_inference_steps=25, guidance_scale=7.5, output_type="pt")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe(generator=torch.Generator().manual_seed(5), **kw).frames
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {i}: {dt * 1e3:.0f} ms for {tuple(out.shape)} (finite: {bool(torch.isfinite(out).all())})")
