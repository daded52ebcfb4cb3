"""Where the HBM-bound 131072 x 320 x 320 projection loses bandwidth: plain / +residual, cache-hot vs rotating operands."""
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
M, N, K = 131072, 320, int(os.environ.get("K", "320"))
w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
for nsets in (1, 6):
    sets = [(torch.randn(M, K, device=dev).half(), torch.randn(M, N, device=dev).half(), torch.empty(M, N, device=dev).half())
            for _ in range(nsets)]
    for resid in (False, True):
        i = [0]
        def run():
            a, r, o = sets[i[0] % nsets]; i[0] += 1
            k.gemm(a, w, b, residual=r if resid else None, out=o)
        t = timeit(run)
        byt = (M * K + (2 if resid else 1) * M * N) * 2
        print(f"sets={nsets} residual={resid}:  {t:7.1f} us  {byt/t/1e6:5.2f} TB/s")
# reference points: a pure copy and an add of the same sizes (torch elementwise kernels)
x = [torch.randn(M, N, device=dev).half() for _ in range(6)]; y = [torch.empty_like(x[0]) for _ in range(6)]
i = [0]
def cp():
    j = i[0] % 6; i[0] += 1
    y[j].copy_(x[j])
t = timeit(cp); print(f"copy {M}x{N}: {t:7.1f} us  {2*M*N*2/t/1e6:5.2f} TB/s")
def add():
    j = i[0] % 6; i[0] += 1
    torch.add(x[j], x[(j + 1) % 6], out=y[j])
t = timeit(add); print(f"add  {M}x{N}: {t:7.1f} us  {3*M*N*2/t/1e6:5.2f} TB/s")
