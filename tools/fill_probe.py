"""One tile round of the 8-wave GEMM with 32 ... 256 of the CUs busy: is the epilogue's store phase limited per CU
(time independent of how many CUs run) or chip-wide (time grows with the number of busy CUs)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg
k = pkg.kernels; dev = torch.device("cuda:0")
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for N, K, epi in ((320, 320, 0), (320, 1280, 0), (2560, 320, k.I2V_EPI_GEGLU)):
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    for tiles in (32, 64, 128, 256, 512, 1024):
        M = 256 * tiles // (N // 320)
        # 20 different A / out buffers so that nothing is cache-hot across the launches of one graph
        aa = [torch.randn(M, K, device=dev).half() for _ in range(10)]
        oo = [torch.empty(M, N // (2 if epi else 1), device=dev, dtype=torch.float16) for _ in range(10)]
        k.gemm(aa[0], w, b, out=oo[0], epilogue=epi); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for a, o in zip(aa, oo):
                k.gemm(a, w, b, out=o, epilogue=epi)
        t = timeit(g.replay) / 10
        print(f"N={N} K={K} epi={epi} tiles={tiles:5d} M={M:6d}: {t:7.1f} us")
