"""The 16 x 16 level's long-K GEMMs / convolutions under I2V_GEMM_SPLIT256 = 0 (128-row tiles, one per CU) / 1 (256-row tiles, K
split in two): time (hipGraph replays) and a checksum + max error against a torch fp32 reference of a row block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.blocks import pack_conv3x3
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / (5 * reps) * 1e3
tag = "SPLIT256=" + os.environ.get("I2V_GEMM_SPLIT256", "0")
torch.manual_seed(0)
for M, N, K in [(8192, 1280, 5120), (8192, 1280, 2560), (8192, 1280, 1280)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = (torch.randn(N, device=dev) * 0.1).half(); r = torch.randn(M, N, device=dev).half()
    fn = lambda: k.gemm(a, w, b, residual=r)
    out = fn(); ref = a[:512].float() @ w.float().T + b.float() + r[:512].float()
    err = (out[:512].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{tag}] gemm {M}x{N}x{K} +res {timeit(fn):8.1f} us  rel err {err:.2e} sum {out.float().sum().item():.6e}", flush=True)
for cin, cout, res in [(1280, 1280, True), (2560, 1280, False), (1920, 1280, False), (640, 1280, False)]:
    x = torch.randn(32, 16, 16, cin, device=dev).half()
    wt = torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5
    w = pack_conv3x3(wt); b = (torch.randn(cout, device=dev) * 0.1).half()
    r = torch.randn(32, 16, 16, cout, device=dev).half() if res else None
    fn = lambda: k.conv3x3(x, w, b, residual=r)
    out = fn()
    ref = torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).float(), wt.half().float(), b.float(), padding=1).permute(0, 2, 3, 1)
    if res: ref = ref + r[:2].float()
    err = (out[:2].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{tag}] conv 8192x{cout}x{9 * cin} {'+res' if res else '    '} {timeit(fn):8.1f} us  rel err {err:.2e} sum {out.float().sum().item():.6e}", flush=True)
