"""The 8 x 8 level's plain GEMMs (2048 rows and fewer: too few tiles for the chip) under I2V_GEMM_DEEP = 0 (generic 4-wave
kernel / 128 x 320 tiles, one K tile in flight) / 1 (128 x 128 or 128 x 256 tiles, three or two K tiles in flight): time
(hipGraph replays) and the max error against a torch fp32 reference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / (5 * reps) * 1e3
tag = "DEEP=" + os.environ.get("I2V_GEMM_DEEP", "1")
torch.manual_seed(0)
shapes = [(2048, 1280, 1280, True), (2048, 1280, 1280, False), (2048, 2560, 1280, False), (2048, 1280, 2560, True),
          (2048, 3840, 1280, False), (2048, 640, 640, False), (512, 1280, 1280, False), (128, 1280, 1280, False),
          (2000, 1280, 1280, True), (2048, 1280, 5120, True), (8192, 1280, 1280, True), (8192, 320, 320, False)]
for M, N, K, res in shapes:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = (torch.randn(N, device=dev) * 0.1).half(); r = torch.randn(M, N, device=dev).half() if res else None
    fn = lambda: k.gemm(a, w, b, residual=r)
    out = fn(); ref = a.float() @ w.float().T + b.float() + (r.float() if res else 0)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{tag}] gemm {M}x{N}x{K} {'+res' if res else '    '} {timeit(fn):8.1f} us  rel err {err:.2e} sum {out.float().sum().item():.6e}", flush=True)
