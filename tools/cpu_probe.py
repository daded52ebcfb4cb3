"""Probe: oracle forward time on the host for several thread counts (sizing of bench.py's cpu_baseline sample)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_weights_cpu
t0 = time.time(); o = build_weights_cpu(); print("build", time.time() - t0, "cpus", os.cpu_count(), flush=True)
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 2, 4, 32, 32, generator=g); ctx = torch.randn(2, 77, 768, generator=g)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.time(); o(x, torch.tensor(500), True, ctx); dt = time.time() - t0
    print(f"threads {th}: 2f x 256^2 forward {dt:.2f}s", flush=True)
