"""AutoencoderKL decode / encode time at the pipeline's sizes (16 frames x 512 x 512), random SD-width weights."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.checkpoint import init_random_weights_
dev = torch.device("cuda:0")
vae = pkg.AutoencoderKL()
init_random_weights_(vae, seed=3)
vae = vae.to(device=dev, dtype=torch.float16).eval()
frames = int(os.environ.get("FRAMES", "16"))
lat = torch.randn(frames, 4, 64, 64, device=dev)
img = torch.randn(1, 3, 512, 512, device=dev)
with torch.no_grad():
    for name, fn in (("decode %d x 64x64 latents" % frames, lambda: vae.decode(lat).sample), ("encode 1 x 512x512", lambda: vae.encode(img).latent_dist.mean)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")

if os.environ.get("SHAPES"):
    from i2v_adapter_unofficial_amd.profiling import KernelProfile
    with torch.no_grad():
        torch.cuda._sleep(int(1e8))
        with KernelProfile() as prof:
            vae.decode(lat)
    rows = sorted(prof.by_shape().items(), key=lambda kv: -kv[1]["ms"])
    tot = sum(d["ms"] for _, d in rows)
    print(f"decode: {tot:.1f} ms in {sum(d['calls'] for _, d in rows)} launches")
    for name, d in rows[:22]:
        print(f"{name:52s} {d['calls']:3d} {d['ms']:8.2f} ms {d['tflops']:7.1f} TF {d['gbps']:6.0f} GB/s")
