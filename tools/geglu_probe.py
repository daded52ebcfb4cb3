"""Fixed per-tile cost of the 64x64-level FF1 GEMM (131072 x 2560, GEGLU) vs K, against the plain epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N = 131072, 2560
for K in (128, 320, 640, 1280):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev).half()
    tg = timeit(lambda: k.gemm(a, w, b, epilogue=k.I2V_EPI_GEGLU))
    tp = timeit(lambda: k.gemm(a, w, b))
    print(f"K={K:5d}: geglu {tg:7.1f} us ({2.0*M*N*K/tg/1e6:5.0f} TF)   plain {tp:7.1f} us ({2.0*M*N*K/tp/1e6:5.0f} TF)   per tile-round {tg/16:5.1f} / {tp/16:5.1f} us")
