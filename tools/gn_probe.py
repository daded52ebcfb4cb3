"""GroupNorm (+SiLU) timing at the step's small-level shapes (I2V_GN_SLAB / _MAX / _ROWB select the one-launch form)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
out = []
for n, hh, c1, c2 in ((32, 8, 1280, 0), (32, 8, 1280, 1280), (32, 16, 1280, 0), (32, 16, 1280, 1280), (32, 16, 1280, 640), (32, 32, 640, 0), (32, 32, 640, 640)):
    x = torch.randn(n, hh, hh, c1, device=dev).half(); x2 = torch.randn(n, hh, hh, c2, device=dev).half() if c2 else None
    ga = torch.randn(c1 + c2, device=dev).half(); be = torch.randn(c1 + c2, device=dev).half()
    g = torch.cuda.CUDAGraph()
    k.groupnorm(x, ga, be, 32, 1e-5, x2=x2, silu=True); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10): k.groupnorm(x, ga, be, 32, 1e-5, x2=x2, silu=True)
    out.append(f"{hh}x{hh}x{c1 + c2}: {timeit(g.replay, 5, 2) / 10:6.1f}")
print(" | ".join(out))
