"""A few step shapes (plain + GEGLU + V^T) timed with the library selected by I2V_LIB_PATH (same-box A/B of builds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out = []
for M, N, K, kind in [(131072, 320, 320, "plain"), (131072, 2560, 320, "geglu"), (32768, 640, 640, "plain"), (32768, 5120, 640, "geglu"),
                      (131072, 320, 1280, "resid"), (131072, 320, 320, "resid"), (32768, 640, 640, "resid"), (8192, 1280, 1280, "resid"),
                      (131072, 320, 320, "vt")]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev).half()
    if kind == "plain": fn = lambda: k.gemm(a, w, b)
    elif kind == "geglu": fn = lambda: k.gemm(a, w, b, epilogue=k.I2V_EPI_GEGLU)
    elif kind == "resid":
        r = torch.randn(M, N, device=dev).half(); fn = lambda: k.gemm(a, w, b, residual=r)
    else: fn = lambda: k.project_vt(a, w, 4096)
    out.append(f"{kind}{M}x{N}x{K}: {timeit(fn):7.1f}us")
print(os.path.basename(os.environ.get("I2V_LIB_PATH", "in-tree")), " | ".join(out))
