#!/usr/bin/env python
"""BASELINE configs[1] (16 f x 512^2, CFG, fp16) timed from a C host: records the pipeline's per-sample preparation and its
denoising step as launch plans (handle.record_prepare_plan / record_step_plan), writes them with the weights to a scratch
directory, runs tests/c_host/denoise_host (plain C: libi2v_hip.so + the HIP runtime, no Python in that process) for 25 DDIM steps
and compares its latents and step time with the Python pipeline's hipGraph replay of the same step in this process.
    python tools/c_denoise_bench.py [--frames 16 --size 512 --steps 25 --ip]      (builder-run; profiles/r6_c_host_denoise.txt)"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--ip", action="store_true")
    args = ap.parse_args()
    import bench
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd import handle as H
    dev = torch.device("cuda:0")
    hu = bench.build_hip_model(dev, seed=1234)
    if args.ip:
        hu._load_ip_adapter_weights(bench.synthetic_ip_state_dict(hu))
    pipe = pkg.I2VAdapterPipeline(unet=hu)
    sch = pipe.scheduler
    sch.set_timesteps(25)
    ts = sch.timesteps
    F, h = args.frames, args.size // 8
    s = bench.sample_inputs(0, F, h, args.ip)
    ie = torch.cat([torch.zeros_like(s["ie"]), s["ie"]]).half().to(dev) if args.ip else None
    ctx = torch.cat([s["ne"], s["pe"]]).half().to(dev)
    lat0, cond = s["lat"].to(dev), s["cond"].to(dev)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        t0 = time.time()
        manifest = H.export_denoiser(pipe, d, num_frames=F, latent_height=h, latent_width=h, batch=1, ctx_len=77,
                                     clip_dim=1024 if args.ip else None, num_inference_steps=25, guidance_scale=7.5)
        H.write_denoise_inputs(os.path.join(d, "inputs.bin"), hu.config, manifest, lat0, cond, ctx, ie)
        # the Python pipeline's own route on the same sample: per-sample preparation, the step captured by torch, replayed
        with torch.no_grad():
            st = dict(latents=lat0.clone(), cond=cond, copies=2, num_frames=F, guidance=7.5, t_table=ts.float().to(dev),
                      coef=sch.step_coefficients(ts).to(dev), step_idx=torch.zeros(1, dtype=torch.int32, device=dev), ctx_text=ctx,
                      ctx_ip=hu._project_image_embeds({"image_embeds": ie}) if ie is not None else None)
            st["ctx_proj"] = hu.project_context(st["ctx_text"], st["ctx_ip"])
            st["temb_table"] = hu.project_time_table(st["t_table"])
            pipe._step(st)
            st["latents"].copy_(lat0)
            st["step_idx"].zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                pipe._step(st)
            st["latents"].copy_(lat0)
            st["step_idx"].zero_()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g.replay()
            torch.cuda.synchronize()
            py_ms = (time.perf_counter() - t1) / args.steps * 1e3
            ref = st["latents"].clone().cpu()
        prep_blob = open(os.path.join(d, "prepare.plan"), "rb").read()
        step_blob = open(os.path.join(d, "step.plan"), "rb").read()
        wbytes = os.path.getsize(os.path.join(d, "weights.bin"))
        print(f"# export_denoiser: plans {len(prep_blob) / 1e6:.2f} + {len(step_blob) / 1e6:.2f} MB, weights.bin {wbytes / 1e9:.2f} GB "
              f"({manifest['weights']['state_dict_keys']} state-dict keys, {manifest['weights']['packs']} packs, "
              f"{manifest['weights']['per_sample_buffers']} per-sample buffers), manifest.json; {time.time() - t0:.0f} s", flush=True)
        del hu, pipe, g, st
        torch.cuda.empty_cache()
        exe = os.path.join(ROOT, "tests", "c_host", "denoise_host")
        r = subprocess.run([exe] + [os.path.join(d, n) for n in ("prepare.plan", "step.plan", "weights.bin", "inputs.bin", "out.bin")] +
                           [str(args.steps)], capture_output=True, text=True)
        print(r.stdout.strip(), r.stderr.strip()[-500:])
        if r.returncode != 0:
            raise SystemExit(r.returncode)
        got = torch.from_numpy(np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32).copy()).view(ref.shape)
    print(f"python pipeline, hipGraph replay of the same step in this process: {py_ms:.3f} ms per step ({1e3 / py_ms:.2f} steps/s)")
    print(f"latents after {args.steps} steps: C host == Python pipeline bit for bit: {bool(torch.equal(got, ref))}; max|latent| {ref.abs().max().item():.3f}")


if __name__ == "__main__":
    main()
