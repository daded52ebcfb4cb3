"""Feed-forward GEMM shapes with many column tiles (GEGLU at the 32x32 / 16x16 / 8x8 levels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
out = []
for M, N, K in ((131072, 2560, 320), (32768, 5120, 640), (8192, 10240, 1280), (2048, 10240, 1280), (8192, 2560, 1280), (8192, 3840, 1280), (32768, 1920, 640)):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    epi = k.I2V_EPI_GEGLU if N >= 2560 and N != 3840 and not (N == 2560 and K == 1280) else k.I2V_EPI_NONE
    out.append(f"{M}x{N}x{K}{' geglu' if epi else ''}: {timeit(lambda: k.gemm(a, w, b, epilogue=epi)):6.1f}")
print(os.path.basename(os.environ.get("I2V_LIB_PATH", "in-tree")), " | ".join(out))
