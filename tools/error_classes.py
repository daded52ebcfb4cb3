#!/usr/bin/env python
"""What fp16 STORAGE alone costs on this network, measured on the CPU oracle -- no GPU involved.

The per-layer budget (tools/error_budget.py -> profiles/r4_error_budget.jsonl) shows no outlier layer: every module adds
3.3 - 4.4e-4 of its output's RMS, of which 2.8e-4 is the nearest-even rounding of the module's fp16 OUTPUT (2^-11 / sqrt 3).
This script separates the classes by running the fp32 oracle on the bench's CFG-shaped config-2 forward with fp16 rounding
applied ONLY at chosen places (everything else exact fp32):

  outer    the residual stream between modules: outputs of conv_in, every ResnetBlock2D, spatial transformer, motion
           module, down / up sampler rounded to fp16 (what ANY implementation that stores the stream in fp16 pays, however
           exact its kernels are)
  blocks   additionally the token stream inside the transformers (output of every transformer block = after its last
           residual add, and proj_in's output)
  everyop  every aten op's result rounded (oracle/fp16_emulation.py: the reference's own fp16 op graph)
  (r6) the precise-stream forms -- the stream is kept as an fp16 hi + lo pair, so the IDENTITY path of every residual add is
  exact and only what a module's branch reads as an MFMA / norm operand is the rounded hi part:
  hilo     inputs of every branch rounded (ResnetBlock2D.norm1 and .conv_shortcut, the entry GroupNorm of the spatial
           transformers and motion modules, the samplers' convs, conv_norm_out); the residual adds see the exact stream
  hilo_sc  the same, but conv_shortcut reads the exact stream too (the 1x1 shortcut GEMM over hi AND lo)
  hilo_blocks  hilo + the token stream inside the transformers rounded (what the fused 64^2 kernels keep in fp16)
  hilo_gn  hilo without the rounding where a GroupNorm kernel could read hi + lo (ResnetBlock2D.norm1, conv_norm_out, the entry norms of
           the transformers / motion modules below the 64^2 level -- at 64^2 they are folded into proj_in, whose MFMA operand is the raw
           hi); conv_shortcut and the samplers' convs still read the rounded stream
  hilo_gn_sc  hilo_gn + conv_shortcut reading the exact stream

  python tools/error_classes.py [--frames 16 --size 512] [--modes outer,blocks,everyop] [--out profiles/r4_error_classes.jsonl]
Test infrastructure only (imports oracle/)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--modes", default="outer,blocks,everyop")
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4_error_classes.jsonl"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)

    import bench
    from oracle import blocks as ob, i2v_adapter as oi
    from oracle.fp16_emulation import emulate_reference_fp16
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel as OracleUNet

    # the bench's weights: same law and seed as init_random_weights_ draws on the GPU are not reproducible on the CPU
    # generator, so this study draws its own (same law: torch default Linear / Conv init, adapter to_out ~ N(0, 0.02^2),
    # fp16-representable) -- the question is the error CLASS sizes, not a bit-comparison with the GPU run
    torch.manual_seed(1234)
    ou = OracleUNet(**bench.SD15)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in ou.named_parameters():
            if ".i2v_adapter.to_out." in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            p.copy_(p.half().float())
    ou.eval()
    ob.Attention._sdpa = lambda self, q, k, v: F.scaled_dot_product_attention(q, k, v, scale=self.scale)

    h_lat = args.size // 8
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, args.frames, 4, h_lat, h_lat, generator=g).half().float()
    x = torch.cat([lat, lat])
    ctx = torch.randn(2, 77, 768, generator=g).half().float()
    t = torch.tensor([481, 481])

    def r16(o):
        if torch.is_tensor(o):
            return o.half().float()
        if isinstance(o, tuple):
            return tuple(r16(e) for e in o)
        if hasattr(o, "sample"):
            o.sample = o.sample.half().float()
        return o

    outer_cls = (ob.ResnetBlock2D, oi.I2VAdapterTransformer2DModel, ob.TransformerTemporalModel, ob.Downsample2D,
                 ob.Upsample2D)
    block_cls = (ob.BasicTransformerBlock,)

    def pre16(mod, a):
        return tuple(r16(e) for e in a)

    def run(mode):
        handles = []
        if mode.startswith("hilo"):
            gn_exact = mode.startswith("hilo_gn")
            c0 = bench.SD15["block_out_channels"][0]
            for n, m in ou.named_modules():
                if isinstance(m, ob.ResnetBlock2D):
                    if not gn_exact:
                        handles.append(m.norm1.register_forward_pre_hook(pre16))
                    if m.conv_shortcut is not None and mode not in ("hilo_sc", "hilo_gn_sc"):
                        handles.append(m.conv_shortcut.register_forward_pre_hook(pre16))
                elif isinstance(m, (oi.I2VAdapterTransformer2DModel, ob.TransformerTemporalModel)):
                    if not gn_exact or m.norm.num_channels == c0:          # (64^2 level: folded into proj_in, the operand is the raw hi)
                        handles.append(m.norm.register_forward_pre_hook(pre16))
                elif isinstance(m, (ob.Downsample2D, ob.Upsample2D)):
                    handles.append(m.conv.register_forward_pre_hook(pre16))
                elif n == "conv_norm_out":
                    if not gn_exact:
                        handles.append(m.register_forward_pre_hook(pre16))
                elif mode == "hilo_blocks" and (isinstance(m, block_cls) or n.endswith(".proj_in")):
                    handles.append(m.register_forward_hook(lambda mod, a, o: r16(o)))
        if mode in ("outer", "blocks"):
            for n, m in ou.named_modules():
                if isinstance(m, outer_cls) or n == "conv_in":
                    handles.append(m.register_forward_hook(lambda mod, a, o: r16(o)))
                elif mode == "blocks" and (isinstance(m, block_cls) or n.endswith(".proj_in")):
                    handles.append(m.register_forward_hook(lambda mod, a, o: r16(o)))
        t0 = time.time()
        with torch.no_grad():
            if mode == "everyop":
                with emulate_reference_fp16():
                    y = ou(x, t, True, ctx).sample
                ob.Attention._sdpa = lambda self, q, k, v: F.scaled_dot_product_attention(q, k, v, scale=self.scale)
            else:
                y = ou(x, t, True, ctx).sample
        for hnd in handles:
            hnd.remove()
        return y, time.time() - t0

    ref, dt = run("fp32")
    print(f"# fp32 reference forward: {dt:.0f} s, max|ref| {ref.abs().max():.3f}, rms {ref.pow(2).mean().sqrt():.3f}",
          file=sys.stderr, flush=True)
    rows = []
    for mode in [m for m in args.modes.split(",") if m]:
        y, dt = run(mode)
        d = y - ref
        row = dict(mode=mode, frames=args.frames, size=args.size, max_ref=ref.abs().max().item(),
                   rms_ref=ref.pow(2).mean().sqrt().item(), max_abs_err=d.abs().max().item(),
                   rms_err=d.pow(2).mean().sqrt().item(), seconds=dt)
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
