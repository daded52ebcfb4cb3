"""Per-shape timing of one denoising step (config 2): which GEMM / conv / attention shapes take the time."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd import kernels as K
from i2v_adapter_unofficial_amd import profiling

dev = torch.device("cuda:0")
model = bench.build_hip_model(dev, seed=1)
with torch.no_grad():
    for p in model.parameters():
        p.normal_(0, 0.02)
F_, h = 16, 64
pipe = pkg.I2VAdapterPipeline(unet=model)
sch = pipe.scheduler; sch.set_timesteps(25); ts = sch.timesteps
st = dict(latents=torch.randn(1, F_, 4, h, h, device=dev), cond=torch.randn(1, 4, h, h, device=dev), copies=2, num_frames=F_,
          guidance=7.5, t_table=ts.float().to(dev), coef=sch.step_coefficients(ts).to(dev),
          step_idx=torch.zeros(1, dtype=torch.int32, device=dev),
          ctx_text=torch.randn(2, 77, 768, device=dev).half(), ctx_ip=None)
recs = []
orig = {n: getattr(K, n) for n in ("gemm", "conv3x3", "attention", "temporal_attention", "groupnorm", "layernorm")}
depth = [0]
def wrap(name, fn):
    def w(*a, **kw):
        if depth[0]:
            return fn(*a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        depth[0] += 1; s.record(); out = fn(*a, **kw); depth[0] -= 1; e.record()
        if name == "gemm":
            key = ("gemm", a[0].shape[0], a[1].shape[0], a[1].shape[1], kw.get("epilogue", 0), kw.get("store", 0)); fl = 2.0 * a[0].shape[0] * a[1].shape[0] * a[1].shape[1]
        elif name == "conv3x3":
            key = ("conv", tuple(a[0].shape), a[1].shape[0], kw.get("stride", 1), kw.get("upsample", False)); fl = 2.0 * out.shape[0] * out.shape[1] * out.shape[2] * a[1].shape[0] * a[1].shape[1]
        elif name == "attention":
            key = ("attn", kw["batch_q"], kw["lq"], kw["lk"], kw["head_dim"], kw.get("kv_group", 1)); fl = 4.0 * kw["batch_q"] * kw["lq"] * kw["lk"] * kw["heads"] * kw["head_dim"]
        else:
            key = (name, tuple(out.shape)); fl = 0.0
        recs.append((key, fl, s, e)); return out
    return w
with torch.no_grad():
    pipe._step(st); torch.cuda.synchronize()
    for n, f in orig.items(): setattr(K, n, wrap(n, f))
    torch.cuda._sleep(int(2e8))
    pipe._step(st); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, s, e in recs:
    d = agg.setdefault(key, [0, 0.0, 0.0]); d[0] += 1; d[1] += s.elapsed_time(e); d[2] += fl
tot = sum(d[1] for d in agg.values())
print(f"total {tot:.2f} ms")
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{d[1]:8.3f} ms {d[0]:4d}x  {d[2] / max(d[1], 1e-9) / 1e9:8.1f} TF/s  {key}")
