"""Full-size oracle fixtures under tests/golden/ (TEST INFRASTRUCTURE; run in the build container, CPU only):

    python -m oracle.make_full_fixtures [config2 config3 config5 latents]

The CPU oracle's fp32 outputs at BASELINE.json's full sizes, on weights re-created from a seed by the CPU generator
(tests/parity.py `oracle_full_width_cpu_seeded`; weights are NOT stored, a checksum of them is), so that the GPU suite compares the
HIP path with the oracle on every configuration in seconds:

  oracle_full_config2.safetensors   noise prediction of the CFG forward (2, 16, 4, 64, 64)      unet:1289-1451, IP off
  oracle_full_config3.safetensors   the same with the IP-Adapter branch (pipe:787-796: the README's setting), unet:1230-1287, 1346-1355
  oracle_full_config5.safetensors   (2, 32, 4, 96, 96): 32 frames = the positional table's limit (unet:725), 768^2
  oracle_latents_config2_25steps.safetensors   latents after DDIM steps 1, 2, 3, 5, 10, 15, 20, 25 of the 25-step CFG trajectory
                                               (pipe:629-700) -- the quantity north_star's tolerance is worded on

fp32 (an fp16 file would round a |x| ~ 2 prediction by up to 4.9e-4: half of the tolerance under test).  ~2 minutes of 8 cores per
16 f x 512^2 forward; the trajectory takes 25 of them.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LATENT_STEPS = (1, 2, 3, 5, 10, 15, 20, 25)


def main():
    from oracle import blocks as ob
    from oracle.pipeline_i2v_adapter import I2VAdapterPipeline as OraclePipeline
    from tests import parity as P
    what = sys.argv[1:] or ["config2", "config3", "config5", "latents"]
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 1)))
    # the reference's attention op (AttnProcessor2_0 -> F.scaled_dot_product_attention); the explicit-softmax form would
    # materialise 2 x 17 GB of scores at the 64 x 64 level
    ob.Attention._sdpa = lambda self, q, k, v: F.scaled_dot_product_attention(q, k, v, scale=self.scale)
    models = {}

    def model(ip):
        if ip not in models:
            t0 = time.time()
            models[ip] = P.oracle_full_width_cpu_seeded(P.FULL_SEED_IP if ip else P.FULL_SEED, ip=ip)
            print(f"# oracle weights (ip={ip}) in {time.time() - t0:.0f} s, checksum {P.weights_checksum(models[ip]).tolist()}", flush=True)
        return models[ip]

    for name, frames, h_lat, ip in (("config2", 16, 64, False), ("config3", 16, 64, True), ("config5", 32, 96, False)):
        if name not in what:
            continue
        ou = model(ip)
        inp = P.full_forward_inputs(frames, h_lat, ip)
        added = {"image_embeds": inp["image_embeds"]} if ip else None
        t0 = time.time()
        with torch.no_grad():
            ref = ou(inp["sample"], inp["t"], True, inp["ctx"], added_cond_kwargs=added).sample
        print(f"# {name}: oracle forward {time.time() - t0:.0f} s, max|ref| {ref.abs().max():.4f}, rms {ref.pow(2).mean().sqrt():.4f}", flush=True)
        save_file({"noise_pred": ref.float().contiguous(), "weights_checksum": P.weights_checksum(ou)},
                  os.path.join(P.GOLDEN_DIR, f"oracle_full_{name}.safetensors"))
    if "latents" in what:
        ou = model(False)
        kw, gens = P.trajectory_inputs(16, 64)
        snaps = {}
        t0 = time.time()

        def cb(i, t, latents):
            if i + 1 in LATENT_STEPS:
                snaps[f"latents_step{i + 1:02d}"] = latents.detach().float().clone().contiguous()
            print(f"#   step {i + 1} (t = {int(t)}) at {time.time() - t0:.0f} s, max|latent| {latents.abs().max():.3f}", flush=True)

        out = OraclePipeline(ou)(**kw, **gens(), callback=cb).frames
        snaps["final"] = out.float().contiguous()          # step 25 with frame 0 re-injected (pipe:699-700)
        snaps["weights_checksum"] = P.weights_checksum(ou)
        save_file(snaps, os.path.join(P.GOLDEN_DIR, "oracle_latents_config2_25steps.safetensors"))
        print(f"# latents: {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
