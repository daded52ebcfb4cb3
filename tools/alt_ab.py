"""Correctness (vs a torch fp32 reference on the GPU) and time of the short-K GEMM flavours under the dispatch selected by
I2V_GEMM_ALT (0 = 8-wave kernel, 1 / 2 = alternating-groups kernel): run once per setting in one gpurun call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu
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
tag = "ALT=" + os.environ.get("I2V_GEMM_ALT", "0")
torch.manual_seed(0)
for M, N, K, flavour in [(131072, 2560, 320, "geglu"), (131072, 960, 320, "ln"), (131072, 640, 320, "ln"), (131072, 320, 320, "res"),
                         (131072, 640, 320, "plain"), (131072, 640, 320, "rv"), (32768, 5120, 640, "geglu"), (32768, 1920, 640, "ln"),
                         (32768, 1280, 640, "ln"), (32768, 640, 640, "res"), (131072, 1280, 320, "geglu_noln")]:
    a = (torch.randn(M, K, device=dev) * 1.5 + 0.3).half()
    w = torch.randn(N, K, device=dev) * K ** -0.5
    b = torch.randn(N, device=dev) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)
    rows = slice(0, 4096)
    af = a[rows].float()
    ln = torch.nn.functional.layer_norm(af, (K,), gam, bet, 1e-5)
    if flavour == "geglu":
        wf, ws, cb = fold_layernorm_geglu(w, b, gam, bet)
        fn = lambda: k.gemm(a, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, 1e-5))
        y = ln @ w.half().float().T + b.half().float(); ref = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:])
    elif flavour == "geglu_noln":
        w16 = torch.stack([w[:N // 2], w[N // 2:]], 1).reshape(N, K).half(); b16 = torch.stack([b[:N // 2], b[N // 2:]], 1).reshape(N).half()
        fn = lambda: k.gemm(a, w16, b16, epilogue=k.I2V_EPI_GEGLU)
        y = af @ w.half().float().T + b.half().float(); ref = y[:, :N // 2] * torch.nn.functional.gelu(y[:, N // 2:])
    elif flavour == "ln":
        wf, ws, cb = fold_layernorm(w, b, gam, bet)
        fn = lambda: k.gemm(a, wf, cb, ln=(ws, 1e-5))
        ref = ln @ w.half().float().T + b.half().float()
    elif flavour == "res":
        r = torch.randn(M, N, device=dev).half(); w16, b16 = w.half(), b.half()
        fn = lambda: k.gemm(a, w16, b16, residual=r)
        ref = af @ w16.float().T + b16.float() + r[rows].float()
    elif flavour == "rv":
        rv = torch.randn(M // 4096, N, device=dev).half(); w16, b16 = w.half(), b.half()
        fn = lambda: k.gemm(a, w16, b16, rowvec=rv, rows_per_vec=4096)
        ref = af @ w16.float().T + b16.float() + rv[0].float()
    else:
        w16 = w.half(); fn = lambda: k.gemm(a, w16)
        ref = af @ w16.float().T
    out = fn(); torch.cuda.synchronize()
    err = (out[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    # a second region far from the first, and the last rows
    tail = slice(M - 2048, M)
    t = timeit(fn)
    fin = bool(torch.isfinite(out).all().item())
    print(f"[{tag}] gemm {M}x{N}x{K} {flavour:10s} {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.0f} TF  rel err {err:.2e} finite {fin} sum {out.float().sum().item():.6e}", flush=True)
