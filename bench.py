#!/usr/bin/env python
"""Headline benchmark: denoising steps/sec @ 16 f x 512 x 512, SD-1.5 + motion-adapter + I2V-Adapter, fp16.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one iteration of the reference loop pipe:666-697 for one (image, prompt) sample: frame-0 overwrite,
CFG duplicate (B = 2), UNet forward, CFG combine, DDIM update -- replayed as one hipGraph.  Inputs are synthetic
(SURVEY 8d): random-init weights of the SD-1.5 + AnimateDiff + I2V-Adapter architecture (seed 1234, adapter to_out
~ N(0, 0.02^2)), resident in HBM before the timed region.  Each rank runs its own sample (weak scaling, no per-step
collective); rank 0 builds the weights and broadcasts them once over RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     live HIP-event timing of the dominant kernel class over one instrumented forward,
  "cpu_baseline": the CPU oracle (fp32, unfused torch graph = the reference's op graph) timed on the host cores on a
                  bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per step, SURVEY.md Appendix B (2*MAC, B = 2 CFG, K/V of adapter & text counted once per clip)
FLOPS_PER_STEP = {"cfg2": 40.199e12, "cfg1": 4.299e12}
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0

SD15 = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
            layers_per_block=2, cross_attention_dim=768, num_attention_heads=8, norm_num_groups=32,
            motion_num_attention_heads=8, motion_max_seq_length=32)


def build_weights_cpu(seed=1234):
    """fp16 state dict of the full architecture with torch default inits under a fixed seed, built with the oracle
    classes on the CPU (the HIP model has the same keys / shapes); adapter to_out re-drawn N(0, 0.02^2)."""
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel as OracleUNet
    from tests.parity import randomize_adapter_out_
    torch.manual_seed(seed)
    o = OracleUNet(**SD15)
    randomize_adapter_out_(o)
    return o


def build_hip_model(dev, state_dict=None):
    import i2v_adapter_unofficial_amd as pkg
    with torch.device("meta"):
        m = pkg.UNetMotionCrossFrameAttnModel(**SD15)
    m = m.to_empty(device=dev).half()
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m.eval()


def cpu_baseline(oracle_unet):
    """Oracle timed on the host cores on a bounded sample: ONE CFG UNet forward (B = 2) at BASELINE config 1's shape
    (8 f x 256^2 => latents (2, 8, 4, 32, 32), 4.299 TFLOP), scaled to the 16 f x 512^2 step by the FLOP ratio."""
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn(2, 77, 768, generator=g)
    # thread count: the host may expose far more hardware threads than the job may use (256 threads measured 5x
    # SLOWER than 16 on the GPU box); calibrate on a 1-frame forward and keep the fastest
    probe = torch.randn(2, 1, 4, 32, 32, generator=g)
    best_t, cores = None, 1
    with torch.no_grad():
        for th in sorted({min(th, os.cpu_count() or 1) for th in (8, 16, 32)}):
            torch.set_num_threads(th)
            t0 = time.time()
            oracle_unet(probe, torch.tensor(500), True, ctx)
            dt = time.time() - t0
            if best_t is None or dt < best_t:
                best_t, cores = dt, th
        torch.set_num_threads(cores)
        x = torch.randn(2, 8, 4, 32, 32, generator=g)
        t0 = time.time()
        oracle_unet(x, torch.tensor(500), True, ctx)
        dt = time.time() - t0
    scaled = dt * FLOPS_PER_STEP["cfg2"] / FLOPS_PER_STEP["cfg1"]
    return {"value": 1.0 / scaled, "unit": "denoising steps/sec", "cores": cores, "kind": "port",
            "sample": (f"one fp32 CFG UNet forward of the CPU oracle at 8f x 256x256 (4.299 TFLOP) took {dt:.2f} s on "
                       f"{cores} threads (fastest of 8/16/32 on a {os.cpu_count()}-thread host); scaled by "
                       "40.199/4.299 to the 16f x 512x512 step"),
            "oracle_tflops": FLOPS_PER_STEP["cfg1"] / dt / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd import kernels as K
    from i2v_adapter_unofficial_amd.profiling import KernelProfile
    from i2v_adapter_unofficial_amd.sharding import broadcast_model_weights

    # ---- weights: rank 0 initialises, one flat RCCL broadcast to the other ranks
    cpu_base = None
    t_start = time.time()
    if rank == 0:
        oracle = build_weights_cpu()
        sd = {k: v.half() for k, v in oracle.state_dict().items()}
        model = build_hip_model(dev, sd)
        del sd
    else:
        oracle = None
        model = build_hip_model(dev)
    if world > 1:
        nbytes = broadcast_model_weights(model, src=0)
        if rank == 0:
            print(f"# broadcast {nbytes / 1e9:.2f} GB of weights over RCCL", file=sys.stderr)

    t_built = time.time()
    # ---- synthetic sample of this rank (SURVEY 8d seeds, offset by rank: independent samples)
    F, h_lat = args.frames, args.size // 8
    g = torch.Generator().manual_seed(1000 * rank + 1)
    cond = torch.randn(1, 4, h_lat, h_lat, generator=g)
    pe = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(1000 * rank + 2))
    ne = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(1000 * rank + 3))
    lat = torch.randn(1, F, 4, h_lat, h_lat, generator=torch.Generator().manual_seed(1000 * rank + 5))
    pipe = pkg.I2VAdapterPipeline(unet=model)
    sch = pipe.scheduler
    sch.set_timesteps(25)
    timesteps = sch.timesteps
    n_tab = len(timesteps)
    st = dict(latents=lat.to(dev), cond=cond.to(dev), copies=2, num_frames=F, guidance=7.5,
              t_table=timesteps.float().to(dev), coef=sch.step_coefficients(timesteps).to(dev),
              step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
              ctx_text=torch.cat([ne, pe]).to(dev, torch.float16).contiguous(), ctx_ip=None)
    lat0 = st["latents"].clone()

    def reset():
        st["latents"].copy_(lat0)
        st["step_idx"].zero_()

    with torch.no_grad():
        pipe._step(st)                      # eager warm-up: packs the kernel-layout weights, sizes the allocator
        reset()
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                pipe._step(st)
            reset()

        def run_step():
            if graph is not None:
                graph.replay()
            else:
                pipe._step(st)

        for i in range(args.warmup):
            run_step()
        reset()
        # ---- timed region: exactly K steps between barrier + synchronize on both sides
        done = 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        while done < args.steps:
            if done % n_tab == 0 and done:
                reset()                     # wrap the 25-entry timestep table: next sample (async copies on the stream)
            run_step()
            done += 1
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        finite = bool(torch.isfinite(st["latents"]).all().item())

        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        # ---- live per-kernel-class roofline (rank 0): one instrumented eager step behind a GPU backlog so that
        #      the event pairs bracket back-to-back kernels, not host launch gaps
        roof, classes = None, None
        if rank == 0:
            reset()
            torch.cuda._sleep(int(2.0e8))   # ~0.1 s device-side spin (not one of our kernels): the host runs ahead
            with KernelProfile() as prof:
                pipe._step(st)
            classes = prof.summary()
            dom = max(classes, key=lambda c: classes[c]["ms"])
            d = classes[dom]
            if d["flops"] > 0:
                roof = {"bound": "mfma", "kernel": dom, "achieved": d["tflops"], "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": d["tflops"] / MFMA_PEAK_TFLOPS, "traffic": None,
                        "launches": d["calls"], "avg_launch_us": d["ms"] * 1e3 / d["calls"],
                        "flops_per_launch": d["flops"] / d["calls"]}
            else:
                roof = {"bound": "hbm", "kernel": dom, "achieved": d["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": d["gbps"] / HBM_PEAK_GBPS, "traffic": None, "launches": d["calls"],
                        "avg_launch_us": d["ms"] * 1e3 / d["calls"], "bytes_per_launch": d["bytes"] / d["calls"]}
            # HBM-side bytes per launch of that class: PMC counters cannot be read from inside the process, so this
            # is the committed result of the separate rocprofv3 --pmc passes over this same command
            # (tools/pmc_traffic.sh -> profiles/r1_traffic.json), valid for the default workload only
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")
            if os.path.exists(tpath) and args.frames == 16 and args.size == 512:
                with open(tpath) as f:
                    tcls = json.load(f).get("classes", {})
                if dom in tcls:
                    roof["traffic"] = tcls[dom]["bytes_per_launch"]
                    roof["traffic_source"] = "profiles/r1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"

    used_graph = graph is not None
    graph = None
    t_gpu_done = time.time()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(oracle)
    if rank == 0:
        print(f"# timings: build {t_built - t_start:.1f}s, gpu {t_gpu_done - t_built:.1f}s, cpu baseline "
              f"{time.time() - t_gpu_done:.1f}s", file=sys.stderr)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        step_flops = FLOPS_PER_STEP["cfg2"] if (F == 16 and args.size == 512) else None
        out = {
            "metric": "denoising steps/sec @ 16fx512x512 SD1.5+I2V-Adapter", "value": value,
            "unit": "denoising steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"SD-v1.5 + motion-adapter-v1-5-2 + I2V-Adapter topology, {F}f x {args.size}x{args.size}, "
                                   "CFG 7.5 (B=2), DDIM 25-step table, fp16, IP off, 1 sample per GPU (BASELINE configs[1])",
                       "samples_per_gpu": 1, "graph": used_graph, "finite": finite,
                       "unet_forwards_per_cfg_half_per_sec": 2 * value,
                       "step_tflops": None if step_flops is None else step_flops / 1e12,
                       "achieved_tflops_per_gpu": None if step_flops is None else step_flops / (ms * 1e-3) / 1e12},
            "roofline": roof,
            "cpu_baseline": cpu_base,
            "kernel_classes": {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                               for k, v in (classes or {}).items()},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
