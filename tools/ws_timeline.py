"""In-kernel timeline of the weight-stationary GEMM (a -DI2V_WS_PROBE build: bash tools/build_variant.sh ws_probe
"-DI2V_WS_PROBE" gemm_ws.hip; run with I2V_LIB_PATH=.ab_libs/ws_probe.so): s_memtime stamps of workgroup 0 / wave 0 at the top
of each 64-row block, after its barrier, and after the MFMA loop (shader-clock cycles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2v_adapter_unofficial_amd as pkg  # noqa: E402
from i2v_adapter_unofficial_amd import kernels as k  # noqa: E402
from i2v_adapter_unofficial_amd.blocks import fold_layernorm, fold_layernorm_geglu  # noqa: E402

dev = torch.device("cuda:0")
dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
orig = k._attach_splitk_workspace


def attach(lib, p, device):
    p.workspace, p.workspace_bytes = dbg.data_ptr(), dbg.numel() * 8
    return dbg


k._attach_splitk_workspace = attach
M = 131072
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 320, generator=g).half().to(dev)
for N, kind in ((320, "plain"), (2560, "plain"), (2560, "geglu_ln"), (640, "ln")):
    w = (torch.randn(N, 320, generator=g) / 18).half()
    b = torch.randn(N, generator=g).half()
    if kind == "plain":
        wd, bd = w.to(dev), b.to(dev)
        fn = lambda: k.gemm(x, wd, bd)
    elif kind == "ln":
        wf, ws, cb = (t.to(dev) for t in fold_layernorm(w.float(), b.float(), torch.ones(320), torch.zeros(320)))
        fn = lambda: k.gemm(x, wf, cb, ln=(ws, 1e-5))
    else:
        wf, ws, cb = (t.to(dev) for t in fold_layernorm_geglu(w.float(), b.float(), torch.ones(320), torch.zeros(320)))
        fn = lambda: k.gemm(x, wf, cb, epilogue=k.I2V_EPI_GEGLU, ln=(ws, 1e-5))
    for _ in range(3):
        fn()
    dbg.zero_()
    fn()
    torch.cuda.synchronize()
    t = dbg.cpu().view(-1, 4)
    n = int((t[:, 0] != 0).sum())
    t = t[:n].double()
    t0 = t[0, 0]
    print(f"# {M}x{N}x320 {kind}: {n} blocks of workgroup 0; cycles: wait+barrier | mfma loop + slices | (unused) | block period")
    for i in range(min(n, 10)):
        per = (t[i + 1, 0] - t[i, 0]) if i + 1 < n else float("nan")
        print(f"  block {i}: {t[i, 1] - t[i, 0]:6.0f} {t[i, 2] - t[i, 1]:6.0f} {t[i, 3] - t[i, 2]:6.0f}   {per:6.0f}   (start {t[i, 0] - t0:.0f})")
    if n > 2:
        d = t[1:n, 0] - t[:n - 1, 0]
        print(f"  mean block period {d.mean():.0f} ticks; mfma loop {(t[:n, 2] - t[:n, 1]).mean():.0f}; epilogue {(t[:n, 3] - t[:n, 2]).mean():.0f}; wait {(t[:n, 1] - t[:n, 0]).mean():.0f}")
