"""The 64x64-level N = 320 projections WITH bias + residual (the in-model form: HBM-bound), for env-override sweeps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, K in [(131072, 320, 320), (131072, 320, 640), (131072, 640, 320), (32768, 640, 640), (131072, 320, 1280)]:
    # several operand sets so that successive launches do not find their inputs in the 256 MB Infinity Cache
    sets = [(torch.randn(M, K, device=dev).half(), torch.randn(M, N, device=dev).half(), torch.empty(M, N, device=dev).half())
            for _ in range(4)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    i = [0]
    def run():
        a, r, o = sets[i[0] % 4]; i[0] += 1
        k.gemm(a, w, b, residual=r, out=o)
    t = timeit(run)
    byt = (M * K + 2 * M * N) * 2
    print(f"{M:7d} {N:5d} {K:5d} +bias+residual  {t:8.1f} us  {2.0*M*N*K/t/1e6:6.0f} TF  {byt/t/1e6:6.2f} TB/s")
