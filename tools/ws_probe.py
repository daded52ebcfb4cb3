"""K = 320 GEMM flavours of the 64 x 64 level, timed one by one (us per call, TFLOP/s, GB/s) under the current environment:
run once with I2V_GEMM_WS=0 (the 8-wave tile kernel) and once with the default (weight-stationary kernel) in ONE gpurun call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu  # noqa: E402

k = pkg.kernels
dev = torch.device("cuda:0")
tag = "WS=" + os.environ.get("I2V_GEMM_WS", "1")
M = int(os.environ.get("WS_PROBE_M", "131072"))
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 320, generator=g).half().to(dev)
res = torch.randn(M, 320, generator=g).half().to(dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for N, kind in ((320, "plain"), (320, "res"), (320, "ln"), (640, "ln_pe"), (960, "ln"), (2560, "geglu_ln"), (1280, "ln")):
    w = (torch.randn(N, 320, generator=g) / 18).half()
    b = torch.randn(N, generator=g).half()
    ga, be = torch.ones(320), torch.zeros(320)
    bytes_ = M * 320 * 2 + M * (N // 2 if "geglu" in kind else N) * 2 + (M * 320 * 2 if kind == "res" else 0)
    if kind == "plain":
        wd, bd = w.to(dev), b.to(dev)
        fn = lambda: k.gemm(x, wd, bd)
    elif kind == "res":
        wd, bd = w.to(dev), b.to(dev)
        fn = lambda: k.gemm(x, wd, bd, residual=res)
    elif kind == "ln":
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w.float(), b.float(), ga, be))
        fn = lambda: k.gemm(x, wf, cb, ln=(ws, 1e-5))
    elif kind == "ln_pe":
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w.float(), b.float(), ga, be))
        pe = torch.randn(32, N, generator=g).half().to(dev)
        fn = lambda: k.gemm(x, wf, cb, ln=(ws, 1e-5), rowvec=pe, rowvec_period=16)
    else:
        wf, ws, cb = (t.to(dev) for t in fold_layernorm_geglu(w.float(), b.float(), ga, be))
        fn = lambda: k.gemm(x, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, 1e-5))
    us = timeit(fn)
    print(f"{tag} {M}x{N}x320 {kind:9s} {us:8.1f} us  {2.0 * M * N * 320 / us / 1e6:7.1f} TFLOP/s  {bytes_ / us / 1e3:7.1f} GB/s", flush=True)
