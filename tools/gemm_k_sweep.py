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
M = int(os.environ.get("M", "131072"))
for N in (320, 2560):
    for K in (64, 128, 320, 640, 1280, 2560):
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        b = torch.randn(N, device=dev).half()
        t0 = timeit(lambda: k.gemm(a, w))
        t1 = timeit(lambda: k.gemm(a, w, b, epilogue=k.I2V_EPI_GEGLU))
        print(f"M={M} N={N} K={K}: plain {t0:8.1f} us ({2.0*M*N*K/t0/1e6:6.0f} TF)   geglu {t1:8.1f} us ({2.0*M*N*K/t1/1e6:6.0f} TF)")
