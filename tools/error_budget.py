#!/usr/bin/env python
"""Per-layer error budget of the fp16 HIP path against the fp32 CPU oracle.

One CFG-shaped UNet forward of BASELINE configs[1] (16 f x 512^2, B = 2, SD-1.5 width) is run through the HIP model and
through the oracle on the same weights and inputs.  After every resnet / spatial transformer / motion module / sampler
two numbers are recorded:
  cum : max|HIP - oracle| of the module's OUTPUT inside the two whole-network runs (error accumulated so far), and
  own : max|HIP(oracle's input, rounded to fp16) - oracle| for that module alone (the error the module adds by itself).
Output: JSON lines (default profiles/r4_error_budget.jsonl) + a table of the widest contributors on stderr.

  python tools/error_budget.py [--frames 16] [--size 512] [--out profiles/r4_error_budget.jsonl] [--no-own]
The oracle forward takes ~50 s on 32 host threads; `own` adds one HIP module call per tap.
Test infrastructure only: imports oracle/ as the checker."""
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
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4_error_budget.jsonl"))
    ap.add_argument("--no-own", action="store_true")
    ap.add_argument("--cfg-shared", type=int, default=1)
    ap.add_argument("--threads", type=int, default=32)
    args = ap.parse_args()

    import bench
    import i2v_adapter_unofficial_amd as pkg
    from i2v_adapter_unofficial_amd import blocks as hb, i2v_adapter as hi
    from oracle import blocks as ob, i2v_adapter as oi
    from oracle.unet_motion_cross_frame_attn import UNetMotionCrossFrameAttnModel as OracleUNet

    dev = torch.device("cuda:0")
    h_lat = args.size // 8
    model = bench.build_hip_model(dev, seed=1234)
    with torch.device("meta"):
        ou = OracleUNet(**bench.SD15)
    ou = ou.to_empty(device="cpu").float()
    ou.load_state_dict({k: v.detach().float().cpu() for k, v in model.state_dict().items()})
    ou.eval()
    ob.Attention._sdpa = lambda self, q, k, v: F.scaled_dot_product_attention(q, k, v, scale=self.scale)
    torch.set_num_threads(min(args.threads, os.cpu_count() or 1))

    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, args.frames, 4, h_lat, h_lat, generator=g).half().float()
    x = torch.cat([lat, lat])
    ctx = torch.randn(2, 77, 768, generator=g).half().float()
    t = torch.tensor([481, 481])

    # ---- HIP run with taps on the internal token-major outputs
    hip_names = {id(m): n for n, m in model.named_modules()}
    taps = {}
    tap_classes = (hb.ResnetBlock2D, hi.I2VAdapterTransformer2DModel, hb.TransformerTemporalModel, hb.Downsample2D,
                   hb.Upsample2D)
    originals = {}

    def wrap(cls):
        orig = cls._fwd
        originals[cls] = orig

        def tapped(self, *a, **kw):
            y = orig(self, *a, **kw)
            taps[hip_names[id(self)]] = y.detach().clone()
            return y
        cls._fwd = tapped

    for c in tap_classes:
        wrap(c)
    with torch.no_grad():
        got = model(x.to(dev), t.to(dev), True, ctx.to(dev),
                    cross_attention_kwargs={"cfg_shared_prefix": bool(args.cfg_shared)}).sample.float().cpu()
    torch.cuda.synchronize()
    for c in tap_classes:
        c._fwd = originals[c]

    # ---- oracle run; every tapped module compares as it goes
    hip_mods = dict(model.named_modules())
    rows = []
    o_classes = (ob.ResnetBlock2D, oi.I2VAdapterTransformer2DModel, ob.TransformerTemporalModel, ob.Downsample2D,
                 ob.Upsample2D)

    def to_dev(v):
        if torch.is_tensor(v):
            return v.to(dev).half() if v.dtype.is_floating_point else v.to(dev)
        if isinstance(v, (tuple, list)):
            return type(v)(to_dev(e) for e in v)
        return v

    def hook_for(name):
        def hook(mod, a, kw, out):
            ref = out[0] if isinstance(out, tuple) else (out.sample if hasattr(out, "sample") else out)
            mref = ref.abs().max().item()
            row = dict(name=name, type=type(mod).__name__, shape=list(ref.shape), max_ref=mref,
                       rms_ref=ref.pow(2).mean().sqrt().item())
            tap = taps.pop(name, None)
            if tap is not None:
                hip_out = tap.permute(0, 3, 1, 2).float().cpu()           # [N, H, W, C] tokens -> NCHW
                n = min(hip_out.shape[0], ref.shape[0])                   # cfg-shared prefix: the HIP run holds one half
                d = (hip_out[:n] - ref[:n])
                row.update(cum_err=d.abs().max().item(), cum_rms=d.pow(2).mean().sqrt().item())
            if not args.no_own:
                with torch.no_grad():
                    hm = hip_mods[name]
                    ho = hm(*to_dev(a), **{k: to_dev(v) for k, v in kw.items() if k != "return_dict"})
                    ho = ho[0] if isinstance(ho, tuple) else (ho.sample if hasattr(ho, "sample") else ho)
                    d = ho.float().cpu() - ref
                    row.update(own_err=d.abs().max().item(), own_rms=d.pow(2).mean().sqrt().item())
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
        return hook

    handles = [m.register_forward_hook(hook_for(n), with_kwargs=True) for n, m in ou.named_modules()
               if isinstance(m, o_classes)]
    t0 = time.time()
    with torch.no_grad():
        ref = ou(x, t, True, ctx).sample
    dt = time.time() - t0
    for hnd in handles:
        hnd.remove()
    final = dict(name="noise_pred", type="UNet output", shape=list(ref.shape), max_ref=ref.abs().max().item(),
                 rms_ref=ref.pow(2).mean().sqrt().item(), cum_err=(got - ref).abs().max().item(),
                 cum_rms=(got - ref).pow(2).mean().sqrt().item(), oracle_seconds=dt)
    rows.append(final)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    print(f"# wrote {args.out}: {len(rows)} taps, oracle forward {dt:.1f} s", file=sys.stderr)
    key = "own_err" if not args.no_own else "cum_err"
    worst = sorted((r for r in rows if key in r), key=lambda r: -r[key] / max(r["max_ref"], 1e-30))[:12]
    for r in worst:
        print(f"# {r['name']:58s} {r['type']:30s} max|ref| {r['max_ref']:8.3f}  {key} {r[key]:.3e} "
              f"({r[key] / r['max_ref']:.2e} of max)  cum {r.get('cum_err', float('nan')):.3e}", file=sys.stderr)
    print(json.dumps(final))


if __name__ == "__main__":
    main()
