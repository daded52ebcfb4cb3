"""The step's GEMM / conv flavours (plain + residual, LayerNorm-folded, GEGLU, V^T store, 3x3 conv) timed under the
dispatch selected by the environment (I2V_GEMM_4W=0/1, I2V_GEMM_BIG=...): run twice in one gpurun call for a same-box A/B.
Each problem is captured 10x into a hipGraph and replayed (no host launch gaps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu, pack_conv3x3
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
rows = []
def gemm_case(M, N, K, flavour):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5)
    b = torch.randn(N, device=dev) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)
    if flavour == "res":
        r = torch.randn(M, N, device=dev).half(); w16, b16 = w.half(), b.half()
        fn = lambda: k.gemm(a, w16, b16, residual=r)
    elif flavour == "ln":
        wf, ws, cb = fold_layernorm(w, b, gam, bet)
        fn = lambda: k.gemm(a, wf, cb, ln=(ws, 1e-5))
    elif flavour == "geglu":
        wf, ws, cb = fold_layernorm_geglu(w, b, gam, bet)
        fn = lambda: k.gemm(a, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, 1e-5))
    elif flavour == "vt":
        wf, ws, cb = fold_layernorm(w, None, gam, bet)
        L = 4096 if M == 131072 else (1024 if M == 32768 else 256)
        out = torch.empty((M // L, N, L), dtype=torch.float16, device=dev)
        fn = lambda: k.project_vt(a, wf, L, out=out, bias=cb, ln=(ws, 1e-5))
    else:
        w16 = w.half(); fn = lambda: k.gemm(a, w16)
    t = timeit(fn)
    rows.append((f"gemm {M}x{N}x{K} {flavour}", t, 2.0 * M * N * K / t / 1e6))
def conv_case(n, hw, cin, cout):
    x = torch.randn(n, hw, hw, cin, device=dev).half()
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5)
    b = torch.randn(cout, device=dev).half(); r = torch.randn(n, hw, hw, cout, device=dev).half()
    t = timeit(lambda: k.conv3x3(x, w, b, residual=r))
    rows.append((f"conv {n * hw * hw}x{cout}x{9 * cin} +res", t, 2.0 * n * hw * hw * cout * 9 * cin / t / 1e6))
only = os.environ.get("ONLY", "")
cases = [(131072, 320, 320, "res"), (131072, 320, 320, "vt"), (131072, 960, 320, "ln"), (131072, 640, 320, "ln"),
         (131072, 2560, 320, "geglu"), (131072, 320, 1280, "res"),
         (32768, 640, 640, "res"), (32768, 640, 640, "vt"), (32768, 1920, 640, "ln"), (32768, 1280, 640, "ln"),
         (32768, 5120, 640, "geglu"), (32768, 640, 2560, "res"),
         (8192, 1280, 1280, "res"), (8192, 1280, 1280, "vt"), (8192, 3840, 1280, "ln"), (8192, 2560, 1280, "ln"),
         (8192, 10240, 1280, "geglu"), (8192, 1280, 5120, "res")]
for c in cases:
    if only and only not in f"{c[0]}x{c[1]}x{c[2]} {c[3]}":
        continue
    gemm_case(*c)
if not only or "conv" in only:
    conv_case(32, 64, 320, 320); conv_case(32, 32, 640, 640); conv_case(32, 16, 1280, 1280)
tag = " ".join(f"{e}={os.environ[e]}" for e in ("I2V_GEMM_4W", "I2V_GEMM_BIG", "I2V_GEMM_PERSIST") if e in os.environ) or "default"
for name, t, tf in rows:
    print(f"[{tag}] {name:36s} {t:8.1f} us {tf:7.0f} TF")
