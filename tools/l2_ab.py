"""The 16 x 16 level's GEMMs (8192 rows: exactly one round of 128 x 320 tiles) under an environment switch: time (hipGraph
replays of a chain of DIFFERENT operands, so that A is not cache-hot) and the max error against torch fp32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fns):
    for f in fns: f()
    torch.cuda.synchronize(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / (5 * len(fns)) * 1e3
tag = os.environ.get("AB_TAG", "")
torch.manual_seed(0)
shapes = [(131072, 320, 320, True, False), (131072, 320, 1280, True, False), (131072, 640, 320, False, True),
          (8192, 1280, 1280, True, False), (8192, 1280, 1280, False, True), (8192, 2560, 1280, False, True),
          (8192, 1280, 2560, True, False), (8192, 1280, 5120, True, False), (32768, 640, 640, True, False)]
for M, N, K, res, ln in shapes:
    n = 12 if M * K < (1 << 27) else 4
    As = [torch.randn(M, K, device=dev).half() for _ in range(n)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = (torch.randn(N, device=dev) * 0.1).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    g_ = torch.randn(K, device=dev).half(); be = torch.randn(K, device=dev).half()
    fold = None
    if ln:
        from i2v_adapter_unofficial_amd.blocks import fold_layernorm
        fold = fold_layernorm(w, b, g_, be)   # (W', wsum, b')
    def mk(a):
        if ln: return lambda: k.gemm(a, fold[0], fold[2], ln=(fold[1], 1e-5))
        return lambda: k.gemm(a, w, b, residual=r)
    try:
        fns = [mk(a) for a in As]; out = fns[0]()
    except Exception as ex:
        print("skip", M, N, K, ln, type(ex).__name__, ex); continue
    x = As[0].float()
    if ln: x = torch.nn.functional.layer_norm(x, (K,), g_.float(), be.float())
    ref = x @ w.float().T + b.float() + (r.float() if res else 0)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{tag}] gemm {M}x{N}x{K} {'+res' if res else '    '} {'+ln' if ln else '   '} {timeit(fns):8.1f} us  rel err {err:.2e}", flush=True)
