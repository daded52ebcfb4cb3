"""One adapter training step (forward + backward + clip + AdamW) of the SD-1.5-width model on one MI355X, timed with HIP
events: python tools/train_probe.py [frames] [size]   (default 16 512; B = 1 clip)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_hip_model
from i2v_adapter_unofficial_amd.training import AdapterOptimizer, UNetAdapterTrainer
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
m = build_hip_model(dev, seed=1234)
tr, opt = UNetAdapterTrainer(m), AdapterOptimizer(m, lr=1e-5)
g = torch.Generator().manual_seed(0)
lat = size // 8
x = torch.randn(1, F, 4, lat, lat, generator=g).half().to(dev)
noise = torch.randn(1, F, 4, lat, lat, generator=g).to(dev)
ctx = torch.randn(1, 77, 768, generator=g).half().to(dev)
t = torch.tensor([481], device=dev)
for it in range(3):
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(); tr.forward(x, t, ctx); e[1].record()
    loss, grads = tr.backward(noise, loss_scale=2.0 ** 12); e[2].record()
    opt.step(grads); e[3].record(); torch.cuda.synchronize()
    print(f"step {it}: forward {e[0].elapsed_time(e[1]):.1f} ms, backward {e[1].elapsed_time(e[2]):.1f} ms, clip + AdamW "
          f"{e[2].elapsed_time(e[3]):.2f} ms, loss {loss.item():.4f}, {sum(v.numel() for v in grads.values()) / 1e6:.1f} M trainable, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
