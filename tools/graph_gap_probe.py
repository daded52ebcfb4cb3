"""Cost of one more kernel inside a captured graph: N tiny launches (i2v_silu_f16 on 64 elements) replayed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
x = torch.randn(64, device=dev).half()
for n in (100, 1000):
    g = torch.cuda.CUDAGraph(); k.silu(x); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n): k.silu(x)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    print(f"{n} tiny kernels per graph: {s.elapsed_time(e) / 10 / n * 1e3:.2f} us per launch")
