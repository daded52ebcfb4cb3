"""Per-round cost of the 8-wave GEMM: N = 2560, K = 320 at 1..16 tile rounds (256 tiles of 256 x 320 per round)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
N = 2560
K = int(os.environ.get("K", "320"))
w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
for rounds in (1, 2, 4, 8, 16):
    M = 8192 * rounds
    a = torch.randn(M, K, device=dev).half()
    # a CUDA graph of 10 launches: no host launch overhead in the figure
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    k.gemm(a, w, b, out=out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            k.gemm(a, w, b, out=out)
    t = timeit(g.replay, iters=5, warm=2) / 10
    print(f"rounds={rounds:2d} M={M:6d}: {t:7.1f} us  per round {t/rounds:5.1f} us  {2.0*M*N*K/t/1e6:5.0f} TF")
