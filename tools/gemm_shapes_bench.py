"""GEMM shapes of one config-2 step (M, N, K) timed under the tile forced by env (I2V_GEMM_BIG=0/128/256)."""
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
shapes = [(131072, 320, 320), (131072, 960, 320), (131072, 2560, 320), (131072, 320, 640), (131072, 320, 1280),
          (32768, 640, 640), (32768, 1280, 640), (32768, 1920, 640), (32768, 5120, 640), (32768, 640, 1280), (32768, 640, 2560),
          (8192, 1280, 1280), (8192, 2560, 1280), (8192, 3840, 1280), (8192, 10240, 1280), (8192, 1280, 2560), (8192, 1280, 5120),
          (2048, 1280, 1280), (2048, 2560, 1280), (2048, 10240, 1280), (2048, 1280, 5120), (320, 131072, 320), (640, 32768, 640), (1280, 8192, 1280)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    t = timeit(lambda: k.gemm(a, w))
    print(f"{M:7d} {N:6d} {K:5d}  {t:8.1f} us  {2.0*M*N*K/t/1e6:6.0f} TF")
