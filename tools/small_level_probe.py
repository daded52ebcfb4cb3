"""GEMM shapes of the 8 x 8 level (M = 2048) under the dispatch selected by I2V_GEMM_BIG (unset: automatic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn):
    g = torch.cuda.CUDAGraph(); fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / 50 * 1e3
out = []
for M, N, K, res in ((2048, 1280, 1280, True), (2048, 2560, 1280, False), (2048, 1280, 5120, True), (2048, 1280, 1280, False),
                     (8192, 1280, 1280, True), (8192, 2560, 1280, False)):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    out.append(f"{M}x{N}x{K}{'+res' if res else ''}: {timeit(lambda: k.gemm(a, w, b, residual=r)):6.1f}")
print(os.environ.get("I2V_GEMM_BIG", "auto"), " | ".join(out))
